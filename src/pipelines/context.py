"""src.pipelines.context (reference: /root/reference/src/pipelines/context.py)."""
from humanvid_amd.scheduler import get_context_scheduler, get_total_steps, ordered_halving, uniform  # noqa: F401
