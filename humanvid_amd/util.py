"""Script-side helpers with the names scripts/pose2vid.py and scripts/pose2img.py import from `src.utils.util`
(/root/reference/src/utils/util.py:16-197): seeding, frame reading, grid / video writing.

Outside the denoising path (plain host code).  Differences from the reference, all forced by this image: `torchvision`
is not a dependency (the image grid is assembled with torch here, same layout as torchvision.utils.make_grid with its
default padding of 2), and `av` (PyAV) is imported lazily -- .gif output and everything that does not touch a video
container work without it, .mp4 reading / writing raise an ImportError that says what is missing.
"""
from __future__ import annotations

import importlib
import os
import os.path as osp
import random
import shutil
import sys
from pathlib import Path

import numpy as np
import torch
from PIL import Image


def seed_everything(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed % (2**32))
    random.seed(seed)


def import_filename(filename):
    spec = importlib.util.spec_from_file_location("mymodule", filename)
    module = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = module
    spec.loader.exec_module(module)
    return module


def _av():
    try:
        import av  # noqa: PLC0415
    except ImportError as e:  # pragma: no cover - depends on the host
        raise ImportError("reading / writing video containers needs PyAV (`av`), which is not installed here; "
                          ".gif output and image grids work without it") from e
    return av


def make_grid(images: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """[n,c,h,w] -> [c, rows*(h+padding)+padding, cols*(w+padding)+padding] (torchvision.utils.make_grid layout;
    a single image is returned unpadded, as torchvision does)."""
    if images.ndim == 3:
        images = images[None]
    n, c, h, w = images.shape
    if c == 1:
        images = images.expand(n, 3, h, w)
        c = 3
    if n == 1:
        return images[0]
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = images.new_full((c, rows * (h + padding) + padding, cols * (w + padding) + padding), pad_value)
    for i in range(n):
        r, q = divmod(i, cols)
        y, x = r * (h + padding) + padding, q * (w + padding) + padding
        grid[:, y:y + h, x:x + w] = images[i]
    return grid


def save_videos_from_pil(pil_images, path, fps=8, bitrate="5000k", crf=19, preset="slow"):
    """same signature, defaults and encoder options as /root/reference/src/utils/util.py:82-103 (the bitrate string goes to
    the encoder as option "b" untouched: "5000k", "10M", ...)"""
    save_fmt = Path(path).suffix
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    if save_fmt == ".mp4":
        av = _av()
        width, height = pil_images[0].size
        container = av.open(path, "w")
        stream = container.add_stream("libx264", rate=fps)
        stream.width, stream.height, stream.pix_fmt = width, height, "yuv420p"
        stream.options = {"b": str(bitrate), "crf": str(crf), "preset": preset}
        for im in pil_images:
            container.mux(stream.encode(av.VideoFrame.from_image(im)))
        container.mux(stream.encode())
        container.close()
    elif save_fmt == ".gif":
        pil_images[0].save(fp=path, format="GIF", append_images=pil_images[1:], save_all=True,
                           duration=(1 / fps * 1000), loop=0)
    else:
        raise ValueError("Unsupported file type. Use .mp4 or .gif.")


def _grid_rows(width, height, n_rows):
    return 2 if width / height > 1 else n_rows  # landscape clips: two per row


def save_videos_grid(videos: torch.Tensor, path: str, rescale=False, n_rows=4, fps=8):
    """videos [b,c,t,h,w] in [0,1] (or [-1,1] with rescale) -> one grid frame per time step."""
    frames = videos.permute(2, 0, 1, 3, 4)
    n_rows = _grid_rows(videos.shape[-1], videos.shape[-2], n_rows)
    outputs = []
    for x in frames:
        x = make_grid(x, nrow=n_rows).permute(1, 2, 0)
        if rescale:
            x = (x + 1.0) / 2.0
        outputs.append(Image.fromarray((x * 255).numpy().astype(np.uint8)))
    save_videos_from_pil(outputs, path, fps)


def save_image_grid(images: torch.Tensor, path: str, n_rows=6):
    n_rows = _grid_rows(images.shape[-1], images.shape[-2], n_rows)
    x = make_grid(images, nrow=n_rows).permute(1, 2, 0)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    Image.fromarray((x * 255).numpy().astype(np.uint8)).save(path)


def read_frames(video_path):
    av = _av()
    container = av.open(video_path)
    stream = next(s for s in container.streams if s.type == "video")
    frames = [frame.to_image().convert("RGB") for packet in container.demux(stream) for frame in packet.decode()]
    container.close()
    return frames


def get_fps(video_path):
    av = _av()
    container = av.open(video_path)
    stream = next(s for s in container.streams if s.type == "video")
    fps = stream.average_rate
    container.close()
    return fps


# ---- checkpoint book-keeping and a notebook helper -------------------------------------------------------------------------
# Not on the denoising path; restored in round 5 because the module is a DROP-IN for src/utils/util.py and the reference's
# train_stage_1.py:42 / train_stage_2.py:45 import these names from it.  Behaviour follows
# /root/reference/src/utils/util.py:17-45, 66-79, 165-173.
def _numbered(names, sep_index):
    """(number, name) pairs of `<stem>-<n>[.ext]` entries, oldest first"""
    return sorted((int(n.split("-")[sep_index].split(".")[0]), n) for n in names)


def save_checkpoint(model, save_dir, prefix, ckpt_num, total_limit=None, logger=None):
    """torch.save(model.state_dict()) -> <save_dir>/<prefix>-<ckpt_num>.pth, after pruning the oldest `<prefix>*` files so that
    at most `total_limit` remain once this one is written; prefix "motion_module" keeps only the motion-module tensors."""
    if total_limit is not None:
        have = _numbered([f for f in os.listdir(save_dir) if f.startswith(prefix)], 1)
        drop = [name for _, name in have[:max(0, len(have) - total_limit + 1)]]
        if drop and logger is not None:
            logger.info(f"{len(have)} checkpoints already exist, removing {len(drop)} checkpoints")
            logger.info(f"removing checkpoints: {', '.join(drop)}")
        for name in drop:
            os.remove(osp.join(save_dir, name))
    state = model.state_dict()
    if prefix == "motion_module":
        state = type(state)((k, v) for k, v in state.items() if "motion_module" in k)
    torch.save(state, osp.join(save_dir, f"{prefix}-{ckpt_num}.pth"))


def delete_additional_ckpt(base_path, num_keep):
    """keep the `num_keep` newest `checkpoint-<n>` directories under base_path"""
    have = _numbered([d for d in os.listdir(base_path) if d.startswith("checkpoint-")], -1)
    for _, name in have[:max(0, len(have) - num_keep)]:
        shutil.rmtree(osp.join(base_path, name), ignore_errors=True)


def show_image_grid(images: torch.Tensor, n_rows=6):
    """display a [n, c, h, w] batch in [0, 1] as one grid (matplotlib is imported on use)"""
    import matplotlib.pyplot as plt

    grid = (make_grid(images, nrow=n_rows).permute(1, 2, 0).squeeze(-1) * 255).numpy().astype(np.uint8)
    plt.imshow(Image.fromarray(grid))
    plt.axis("off")
    plt.show()

