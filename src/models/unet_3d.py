"""src.models.unet_3d (reference: /root/reference/src/models/unet_3d.py)."""
from humanvid_amd.unet3d import InflatedConv3d, InflatedGroupNorm, UNet3DConditionModel, UNet3DConditionOutput  # noqa: F401
