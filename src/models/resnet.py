"""src.models.resnet (reference: /root/reference/src/models/resnet.py)."""
from humanvid_amd.unet3d import Downsample3D, InflatedConv3d, InflatedGroupNorm, ResnetBlock3D, Upsample3D  # noqa: F401
