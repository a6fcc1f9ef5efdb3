"""src.models.transformer_3d (reference: /root/reference/src/models/transformer_3d.py)."""
from humanvid_amd.unet3d import Transformer3DModel  # noqa: F401
