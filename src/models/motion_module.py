"""src.models.motion_module (reference: /root/reference/src/models/motion_module.py)."""
from humanvid_amd.unet3d import (PositionalEncoding, TemporalTransformer3DModel, TemporalTransformerBlock,  # noqa: F401
                                 VanillaTemporalModule, VersatileAttention, get_motion_module, zero_module)
