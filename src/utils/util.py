"""src.utils.util (reference: /root/reference/src/utils/util.py) -- the helpers the inference scripts use.  The training-only
ones (save_checkpoint, delete_additional_ckpt, show_image_grid) are out of scope."""
from humanvid_amd.util import (get_fps, import_filename, make_grid, read_frames, save_image_grid,  # noqa: F401
                               save_videos_from_pil, save_videos_grid, seed_everything)
