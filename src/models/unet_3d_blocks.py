"""src.models.unet_3d_blocks (reference: /root/reference/src/models/unet_3d_blocks.py)."""
from humanvid_amd.unet3d import (CrossAttnDownBlock3D, CrossAttnUpBlock3D, DownBlock3D, UNetMidBlock3DCrossAttn,  # noqa: F401
                                 UpBlock3D)
