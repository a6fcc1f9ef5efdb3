"""src.models.attention (reference: /root/reference/src/models/attention.py)."""
from humanvid_amd.unet2d import BasicTransformerBlock  # noqa: F401
from humanvid_amd.unet3d import TemporalBasicTransformerBlock  # noqa: F401
