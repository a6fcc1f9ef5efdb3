"""Camera front-end: TUM pose rows -> relative camera-to-world -> per-pixel Pluecker embedding.

API-compatible with `Camera` / `ray_condition` / `get_relative_pose` of
/root/reference/src/dataset/dance_image_h_v_camera.py:17-130 and `camera_file_to_embedding` of
/root/reference/scripts/pose2vid.py:29-83 (pinned by tests/golden/plucker.npz, which the reference's own code produced
from data/test_set/camera_test_set.zip), but organised around a whole trajectory at once:

 * `CameraTrack`: every row of a pose file in one float64 numpy pass -- unit quaternions to rotations through the
   Euler-Rodrigues form  R = (1 - 2|v|^2) I + 2 v v^T + 2 w [v]x, rigid inverses in closed form ([R^T | -R^T t]),
   poses relative to the first camera by one batched matmul.  `Camera` is a one-row view with the attribute names the
   reference's callers use (fx, fy, cx, cy, c2w_mat, w2c_mat).
 * `ray_condition`: the Pluecker map (o x d, d) from broadcast pixel-centre grids -- the same arithmetic, per output
   element, as the device kernel hv_plucker_unshuffle (csrc/hv_elementwise.h), which fuses it into the
   CameraPoseEncoder's PixelUnshuffle so that the per-step path never materialises the map.

Runs once per clip on the host.  SURVEY.md row a22.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np
import torch

# which side of the transform a dataset's pose files store (dance_image_h_v_camera.py:49-66)
_STORES_C2W = ("pexels", "inference", "ubc", "tiktok", "webvid", "test")
_STORES_W2C = ("bedlam", "blender", "ue_rendered")


def _rotations(q: np.ndarray) -> np.ndarray:
    """unit quaternions [n,4] as (x, y, z, w) -> rotation matrices [n,3,3]."""
    v, w = q[:, :3], q[:, 3]
    n = q.shape[0]
    skew = np.zeros((n, 3, 3))
    skew[:, 0, 1], skew[:, 0, 2] = -v[:, 2], v[:, 1]
    skew[:, 1, 0], skew[:, 1, 2] = v[:, 2], -v[:, 0]
    skew[:, 2, 0], skew[:, 2, 1] = -v[:, 1], v[:, 0]
    vv = (v * v).sum(1)
    return ((1.0 - 2.0 * vv)[:, None, None] * np.eye(3) + 2.0 * v[:, :, None] * v[:, None, :]
            + 2.0 * w[:, None, None] * skew)


def _rigid(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    M = np.zeros((R.shape[0], 4, 4))
    M[:, :3, :3], M[:, :3, 3], M[:, 3, 3] = R, t, 1.0
    return M


class CameraTrack:
    """rows: [n, 10 | 11] = timestamp, tx ty tz, qx qy qz qw, fx, fy [, scene scale]."""

    def __init__(self, rows, pose_file_name: str, image_scale=(1920, 1080)):
        a = np.asarray(rows, dtype=np.float64)
        if a.ndim == 1:
            a = a[None]
        assert a.shape[1] in (10, 11), (
            f"length of entry should be 11 (extrinsic + fx fy + scale) or 10 (+ fx fy), got {a.shape[1]}")
        wd, ht = image_scale
        if wd > ht:  # landscape: the file's fx is trusted, fy follows from the aspect ratio (and vice versa)
            self.fx = a[:, 8].copy()
            self.fy = self.fx * (wd / ht)
        else:
            self.fy = a[:, 9].copy()
            self.fx = self.fy * (ht / wd)
        self.cx = np.full(len(a), 0.5)
        self.cy = np.full(len(a), 0.5)
        self.timestamp = a[:, 0].copy()
        q = a[:, 4:8] / np.linalg.norm(a[:, 4:8], axis=1, keepdims=True)
        R, t = _rotations(q), a[:, 1:4]
        if any(k in pose_file_name for k in _STORES_W2C):
            fwd, inv_is_c2w = _rigid(R, t), True
        elif any(k in pose_file_name for k in _STORES_C2W):
            scale = a[:, 10:11] if a.shape[1] == 11 else 1.0
            fwd, inv_is_c2w = _rigid(R, t * scale), False
        else:
            raise ValueError(f"Unknown camera pose dataset name: {pose_file_name}")
        Rt = np.swapaxes(fwd[:, :3, :3], 1, 2)
        inv = _rigid(Rt, -np.einsum("nij,nj->ni", Rt, fwd[:, :3, 3]))
        self.c2w, self.w2c = (inv, fwd) if inv_is_c2w else (fwd, inv)

    def __len__(self):
        return len(self.fx)

    def select(self, idx) -> "CameraTrack":
        out = object.__new__(CameraTrack)
        idx = np.asarray(idx, dtype=np.int64)
        for k in ("fx", "fy", "cx", "cy", "timestamp", "c2w", "w2c"):
            setattr(out, k, getattr(self, k)[idx])
        return out

    def relative_to_first(self) -> np.ndarray:
        """[n,4,4] float32: pose of every camera in the first camera's frame (the first one is the identity)."""
        rel = np.einsum("ij,njk->nik", self.w2c[0], self.c2w)
        rel[0] = np.eye(4)
        return rel.astype(np.float32)

    def intrinsics_px(self, img_size) -> np.ndarray:
        wd, ht = img_size
        return np.stack([self.fx * wd, self.fy * ht, self.cx * wd, self.cy * ht], axis=1).astype(np.float32)


class Camera(object):
    """One pose row (the reference's per-frame object)."""

    def __init__(self, entry, pose_file_name, image_scale=(1920, 1080)):
        assert len(entry) == 10 or len(entry) == 11, (
            f"length of entry should be 11 (extrinsic + fx fy + scale) or 10 (+ fx fy), got {len(entry)}")
        tr = CameraTrack([list(entry)], pose_file_name, image_scale)
        self.fx, self.fy, self.cx, self.cy = float(tr.fx[0]), float(tr.fy[0]), 0.5, 0.5
        self.timestamp = float(tr.timestamp[0])
        self.c2w_mat, self.w2c_mat = tr.c2w[0], tr.w2c[0]


def _as_track(cams) -> CameraTrack:
    if isinstance(cams, CameraTrack):
        return cams
    out = object.__new__(CameraTrack)
    for k, attr in (("fx", "fx"), ("fy", "fy"), ("cx", "cx"), ("cy", "cy"), ("timestamp", "timestamp")):
        setattr(out, k, np.asarray([getattr(c, attr) for c in cams], dtype=np.float64))
    out.c2w = np.stack([c.c2w_mat for c in cams])
    out.w2c = np.stack([c.w2c_mat for c in cams])
    return out


def get_relative_pose(cam_params: Sequence[Camera]):
    return _as_track(cam_params).relative_to_first()


def ray_condition(K, c2w, H, W, device, flip_flag=None):
    """K [B,V,4] (fx,fy,cx,cy in pixels), c2w [B,V,4,4] -> Pluecker map [B,V,H,W,6] = (o x d, d), d the unit ray
    through each pixel centre in the world (= first camera) frame, o the camera centre."""
    if flip_flag is not None and int(torch.sum(flip_flag).item()) > 0:
        raise NotImplementedError("flip_flag is a training-time augmentation, unused at inference")
    K, c2w = K.to(device), c2w.to(device)
    dt = c2w.dtype
    fx, fy, cx, cy = (K[..., i].to(dt)[..., None, None] for i in range(4))          # [B,V,1,1]
    u = (torch.arange(W, device=device, dtype=dt) + 0.5)[None, None, None, :]       # pixel centres
    v = (torch.arange(H, device=device, dtype=dt) + 0.5)[None, None, :, None]
    x = ((u - cx) / fx).expand(*K.shape[:2], H, W)
    y = ((v - cy) / fy).expand(*K.shape[:2], H, W)
    cam_dir = torch.stack([x, y, torch.ones_like(x)], dim=-1)
    cam_dir = cam_dir / cam_dir.norm(dim=-1, keepdim=True)
    d = torch.einsum("bvij,bvhwj->bvhwi", c2w[..., :3, :3], cam_dir)
    o = c2w[..., :3, 3][:, :, None, None, :]
    moment = torch.stack([o[..., 1] * d[..., 2] - o[..., 2] * d[..., 1],
                          o[..., 2] * d[..., 0] - o[..., 0] * d[..., 2],
                          o[..., 0] * d[..., 1] - o[..., 1] * d[..., 0]], dim=-1)
    return torch.cat([moment, d], dim=-1)


def load_cameras(pose_file, img_size) -> List[Camera]:
    rows = np.loadtxt(pose_file, dtype=np.float64, ndmin=2)
    return [Camera(r, pose_file, img_size) for r in rows]


def static_camera_entry(img_size):
    """identity pose with the default focal length of scripts/pose2vid.py:56-61 on the trusted axis"""
    f = 1.788079
    fx, fy = (1.0, f) if img_size[0] > img_size[1] else (f, 1.0)
    return [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, fx, fy, 1.0]


def cameras_to_params(cam_params, img_size):
    """cam_params[0] is the reference camera, cam_params[1:] the target frames -> (K [F,4] intrinsics in pixels,
    c2w [F,4,4] poses relative to the reference camera): the inputs of `ray_condition`, and of the on-device
    Pluecker front-end (CameraPoseEncoder.forward_nhwc_from_cameras)."""
    tr = _as_track(cam_params)
    return torch.from_numpy(tr.intrinsics_px(img_size)[1:].copy()), torch.from_numpy(tr.relative_to_first()[1:].copy())


def cameras_to_embedding(cam_params, img_size) -> torch.Tensor:
    """cam_params[0] is the reference camera, cam_params[1:] the target frames -> [1, F, 6, H, W]."""
    K, c2w = cameras_to_params(cam_params, img_size)
    pl = ray_condition(K[None], c2w[None], img_size[1], img_size[0], device="cpu")
    return pl[0].permute(0, 3, 1, 2).contiguous()[None]


def camera_file_to_embedding(video_length, pose_path, ref_img_idx, tgt_img_idx, img_size, is_same_video=True):
    """The script-level helper (scripts/pose2vid.py:29-83): locate the camera file next to the pose video, fall back to a
    static camera when there is none, put the reference frame first."""
    camera_file = pose_path
    for a, b in (("/dwpose/", "/camera/"), ("/pose_videos/", "/camera/"), (".mp4", ".txt"), (".png", ".txt")):
        camera_file = camera_file.replace(a, b)
    if os.path.exists(camera_file):
        track = CameraTrack(np.loadtxt(camera_file, dtype=np.float64, ndmin=2), camera_file, img_size)
    else:
        track = CameraTrack([static_camera_entry(img_size)] * video_length, "test", img_size)
    first = ref_img_idx if is_same_video else tgt_img_idx[0]
    return cameras_to_embedding(track.select([first] + list(tgt_img_idx)), img_size)
