"""src.pipelines.utils (reference: /root/reference/src/pipelines/utils.py)."""
from humanvid_amd.latent_interp import (get_tensor_interpolation_method, linear, set_tensor_interpolation_method,  # noqa: F401
                                        slerp)
