"""src.pipelines.pipeline_pose2vid_long (reference: /root/reference/src/pipelines/pipeline_pose2vid_long.py)."""
from humanvid_amd.pipeline import Pose2VideoPipeline, Pose2VideoPipelineOutput  # noqa: F401
