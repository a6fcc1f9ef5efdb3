"""src.pipelines.pipeline_pose2img (reference: /root/reference/src/pipelines/pipeline_pose2img.py)."""
from humanvid_amd.pipeline import Pose2ImagePipeline, Pose2ImagePipelineOutput  # noqa: F401
