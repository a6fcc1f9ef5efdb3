"""src.utils.util (reference: /root/reference/src/utils/util.py)."""
from humanvid_amd.util import (delete_additional_ckpt, get_fps, import_filename, make_grid, read_frames,  # noqa: F401
                               save_checkpoint, save_image_grid, save_videos_from_pil, save_videos_grid, seed_everything,
                               show_image_grid)
