"""humanvid_amd -- MI355X-native CamAnimate denoising path (HIP kernels behind the reference's Python surface)."""
__version__ = "0.1.0"
