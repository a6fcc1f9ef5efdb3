"""src.pipelines.pipeline_pose2vid (reference: /root/reference/src/pipelines/pipeline_pose2vid.py; the class is
called Pose2VideoPipeline there as well -- the all-frames-in-one-forward variant without camera control)."""
from humanvid_amd.pipeline import Pose2VideoPipelineOutput  # noqa: F401
from humanvid_amd.pipeline import Pose2VideoShortPipeline as Pose2VideoPipeline  # noqa: F401
