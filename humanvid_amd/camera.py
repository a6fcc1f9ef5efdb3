"""Camera front-end: TUM pose rows -> relative camera-to-world -> per-pixel Pluecker embedding.

Drop-in for `Camera` / `ray_condition` / `get_relative_pose` of
/root/reference/src/dataset/dance_image_h_v_camera.py:17-130 and `camera_file_to_embedding` of
/root/reference/scripts/pose2vid.py:29-83.  Runs once per clip on the host (float64 pose algebra in
numpy, float32 ray map in torch) exactly like the reference; the per-step consumer is the
CameraPoseEncoder (conditioning.py).  SURVEY.md row a22.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np
import torch

_C2W_NAMES = ("pexels", "inference", "ubc", "tiktok", "webvid", "test")
_W2C_NAMES = ("bedlam", "blender", "ue_rendered")


class Camera(object):
    def __init__(self, entry, pose_file_name, image_scale=(1920, 1080)):
        assert len(entry) == 10 or len(entry) == 11, (
            f"length of entry should be 11 (extrinsic + fx fy + scale) or 10 (+ fx fy), got {len(entry)}")
        if image_scale[0] > image_scale[1]:
            self.fx = entry[8]
            self.fy = self.fx * (image_scale[0] / image_scale[1])
        else:
            self.fy = entry[9]
            self.fx = self.fy * (image_scale[1] / image_scale[0])
        self.cx = 0.5
        self.cy = 0.5
        self.timestamp = entry[0]
        tx, ty, tz = entry[1:4]
        qx, qy, qz, qw = entry[4:8]
        scale = entry[10] if len(entry) == 11 else 1.0
        norm = np.linalg.norm([qx, qy, qz, qw])
        # (the reference appends a warning line to ./data/broken_kps_videos.txt when |norm-1| > 1e-3;
        #  a library must not write into the caller's tree, so the side effect is dropped)
        qx, qy, qz, qw = [x / norm for x in [qx, qy, qz, qw]]
        rotation = self.quaternion_to_rotation_matrix(qx, qy, qz, qw)
        translation = np.array([tx, ty, tz])
        if any(k in pose_file_name for k in _W2C_NAMES):
            self.w2c_mat = np.eye(4)
            self.w2c_mat[:3, :3] = rotation
            self.w2c_mat[:3, 3] = translation
            self.c2w_mat = np.linalg.inv(self.w2c_mat)
        elif any(k in pose_file_name for k in _C2W_NAMES):
            self.c2w_mat = np.eye(4)
            self.c2w_mat[:3, :3] = rotation
            self.c2w_mat[:3, 3] = translation * scale
            self.w2c_mat = np.linalg.inv(self.c2w_mat)
        else:
            raise ValueError(f"Unknown camera pose dataset name: {pose_file_name}")

    @staticmethod
    def quaternion_to_rotation_matrix(qx, qy, qz, qw):
        return np.array([
            [1 - 2 * qy**2 - 2 * qz**2, 2 * qx * qy - 2 * qz * qw, 2 * qx * qz + 2 * qy * qw],
            [2 * qx * qy + 2 * qz * qw, 1 - 2 * qx**2 - 2 * qz**2, 2 * qy * qz - 2 * qx * qw],
            [2 * qx * qz - 2 * qy * qw, 2 * qy * qz + 2 * qx * qw, 1 - 2 * qx**2 - 2 * qy**2],
        ])


def ray_condition(K, c2w, H, W, device, flip_flag=None):
    """K [B,V,4] (fx,fy,cx,cy in pixels), c2w [B,V,4,4] -> Pluecker map [B,V,H,W,6] = (o x d, d)."""
    if flip_flag is not None and int(torch.sum(flip_flag).item()) > 0:
        raise NotImplementedError("flip_flag is a training-time augmentation, unused at inference")
    B, V = K.shape[:2]
    j, i = torch.meshgrid(torch.linspace(0, H - 1, H, device=device, dtype=c2w.dtype),
                          torch.linspace(0, W - 1, W, device=device, dtype=c2w.dtype), indexing="ij")
    i = i.reshape([1, 1, H * W]).expand([B, V, H * W]) + 0.5
    j = j.reshape([1, 1, H * W]).expand([B, V, H * W]) + 0.5
    fx, fy, cx, cy = K.chunk(4, dim=-1)
    zs = torch.ones_like(i)
    xs = (i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    zs = zs.expand_as(ys)
    directions = torch.stack((xs, ys, zs), dim=-1)
    directions = directions / directions.norm(dim=-1, keepdim=True)
    rays_d = directions @ c2w[..., :3, :3].transpose(-1, -2)
    rays_o = c2w[..., :3, 3]
    rays_o = rays_o[:, :, None].expand_as(rays_d)
    rays_dxo = torch.cross(rays_o, rays_d, dim=-1)
    plucker = torch.cat([rays_dxo, rays_d], dim=-1)
    return plucker.reshape(B, c2w.shape[1], H, W, 6)


def get_relative_pose(cam_params: Sequence[Camera]):
    abs_w2cs = [c.w2c_mat for c in cam_params]
    abs_c2ws = [c.c2w_mat for c in cam_params]
    target = np.eye(4)
    abs2rel = target @ abs_w2cs[0]
    ret = [target] + [abs2rel @ c2w for c2w in abs_c2ws[1:]]
    return np.array(ret, dtype=np.float32)


def load_cameras(pose_file, img_size) -> List[Camera]:
    with open(pose_file, "r") as f:
        rows = [[float(x) for x in ln.strip().split(" ")] for ln in f.readlines()]
    return [Camera(r, pose_file, img_size) for r in rows]


def static_camera_entry(img_size):
    """scripts/pose2vid.py:56-61."""
    if img_size[0] > img_size[1]:
        return [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.788079, 1.0]
    return [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.788079, 1.0, 1.0]


def cameras_to_params(cam_params: Sequence[Camera], img_size):
    """cam_params[0] is the reference camera, cam_params[1:] the target frames -> (K [F,4] intrinsics in pixels,
    c2w [F,4,4] poses relative to the reference camera): the inputs of `ray_condition`, and of the on-device
    Pluecker front-end (CameraPoseEncoder.forward_nhwc_from_cameras)."""
    K = np.asarray([[c.fx * img_size[0], c.fy * img_size[1], c.cx * img_size[0], c.cy * img_size[1]]
                    for c in cam_params[1:]], dtype=np.float32)
    return torch.as_tensor(K), torch.as_tensor(get_relative_pose(cam_params)[1:])


def cameras_to_embedding(cam_params: Sequence[Camera], img_size) -> torch.Tensor:
    """cam_params[0] is the reference camera, cam_params[1:] the target frames -> [1, F, 6, H, W]."""
    K, c2w = cameras_to_params(cam_params, img_size)
    c2w = c2w[None]
    pl = ray_condition(K[None], c2w, img_size[1], img_size[0], device="cpu")
    return pl[0].permute(0, 3, 1, 2).contiguous().unsqueeze_(0)


def camera_file_to_embedding(video_length, pose_path, ref_img_idx, tgt_img_idx, img_size, is_same_video=True):
    camera_file = (pose_path.replace("/dwpose/", "/camera/").replace("/pose_videos/", "/camera/")
                   .replace(".mp4", ".txt").replace(".png", ".txt"))
    if not os.path.exists(camera_file):
        cams = [Camera(static_camera_entry(img_size), "test", img_size)] * video_length
    else:
        cams = load_cameras(camera_file, img_size)
    if is_same_video:
        cams = [cams[ref_img_idx]] + [cams[idx] for idx in tgt_img_idx]
    else:
        cams = [cams[tgt_img_idx[0]]] + [cams[idx] for idx in tgt_img_idx]
    return cameras_to_embedding(cams, img_size)
