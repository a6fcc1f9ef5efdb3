"""src.cameractrl.pose_adaptor (reference: /root/reference/src/cameractrl/pose_adaptor.py)."""
from humanvid_amd.conditioning import CameraPoseEncoder  # noqa: F401
