"""src.models.unet_2d_condition (reference: /root/reference/src/models/unet_2d_condition.py): the ReferenceNet."""
from humanvid_amd.unet2d import UNet2DConditionModel  # noqa: F401
