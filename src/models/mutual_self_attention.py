"""src.models.mutual_self_attention (reference: /root/reference/src/models/mutual_self_attention.py)."""
from humanvid_amd.reference_control import ReferenceAttentionControl, torch_dfs  # noqa: F401
