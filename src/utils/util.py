"""src.utils.util (reference: /root/reference/src/utils/util.py) -- the helpers the inference scripts use, plus the three the
reference's training scripts import from the same module (save_checkpoint, delete_additional_ckpt, show_image_grid)."""
from humanvid_amd.util import (delete_additional_ckpt, get_fps, import_filename, make_grid, read_frames,  # noqa: F401
                               save_checkpoint, save_image_grid, save_videos_from_pil, save_videos_grid, seed_everything,
                               show_image_grid)
