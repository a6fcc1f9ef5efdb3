"""src.models.pose_guider (reference: /root/reference/src/models/pose_guider.py)."""
from humanvid_amd.conditioning import PoseGuider  # noqa: F401
