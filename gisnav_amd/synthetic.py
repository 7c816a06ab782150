"""Seeded synthetic frame<->tile pairs and the camera model.

The reference feeds PoseNode one ``OrthoStereoImage`` per call: a packed SIFT keypoint
cloud for the camera frame (``KEYPOINT_DTYPE``, ros/gisnav/gisnav/core/_shared.py:26-35),
a mono8 orthoimage tile and a mono8 DEM (pose_node.py:207-223).  No dataset or SIFT
extractor is available offline, so the benchmark and the parity tests use this generator:
SIFT-like integer-valued descriptors, a smooth u8 DEM, a known ground-truth pose, and a
query side made by projecting tile points through that pose.

Pure numpy; used by tests, bench.py and smoke() for BOTH the HIP path and the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# docker/gscam/camera_calibration.yaml:1-12 (fx=fy=205.4696, cx=320) with cy moved to 240
# for the 640x480 frame BASELINE.json names.
IMG_W, IMG_H = 640, 480
K_MATRIX = np.array([[205.4696, 0.0, 320.0], [0.0, 205.4696, 240.0], [0.0, 0.0, 1.0]], dtype=np.float64)


@dataclass
class Pair:
    """One camera-frame <-> map-tile pair, in the arrays PoseNode unpacks (pose_node.py:207-252)."""
    kp_q: np.ndarray      # (N,2) f32  query keypoint x,y
    desc_q: np.ndarray    # (N,128) f32 integer-valued SIFT descriptors
    size_q: np.ndarray    # (N,) f32   cv2.KeyPoint.size (diameter px)
    angle_q: np.ndarray   # (N,) f32   cv2.KeyPoint.angle (degrees)
    kp_r: np.ndarray      # (M,2)
    desc_r: np.ndarray    # (M,128)
    size_r: np.ndarray
    angle_r: np.ndarray
    dem: np.ndarray       # (H,W) u8
    ref: np.ndarray       # (H,W) u8 (tile raster; unused by the matcher once SIFT is cached)
    R_gt: np.ndarray      # (3,3) f64  world(tile px) -> camera
    t_gt: np.ndarray      # (3,1) f64
    gt_q2r: np.ndarray    # (N,) int64 index of the true tile keypoint for each query kp, -1 = distractor


def _sift_like_descriptors(rng: np.random.Generator, n: int) -> np.ndarray:
    g = rng.gamma(0.5, 1.0, size=(n, 128))
    g *= 512.0 / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-9)
    return np.clip(np.rint(g), 0, 255).astype(np.float32)


def _smooth_dem(rng: np.random.Generator, h: int, w: int) -> np.ndarray:
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    z = np.zeros((h, w))
    for _ in range(4):
        fx, fy = rng.uniform(0.5, 2.5, 2) * 2 * np.pi / np.array([w, h])
        z += rng.uniform(0.5, 1.0) * np.sin(fx * xx + rng.uniform(0, 6.28)) * np.cos(fy * yy + rng.uniform(0, 6.28))
    z = (z - z.min()) / max(z.max() - z.min(), 1e-9) * 40.0
    return np.rint(z).astype(np.uint8)


def _rot(yaw: float, pitch: float, roll: float) -> np.ndarray:
    cz, sz = np.cos(yaw), np.sin(yaw)
    cy, sy = np.cos(pitch), np.sin(pitch)
    cx, sx = np.cos(roll), np.sin(roll)
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    return rz @ ry @ rx


def make_pair(pair_index: int, n_q: int = 1024, n_r: int = 1024, h: int = IMG_H, w: int = IMG_W,
              flat_dem: bool = False, match_fraction: float = 0.7) -> Pair:
    """seed = 1234 + pair_index."""
    rng = np.random.default_rng(1234 + pair_index)
    kp_r = np.column_stack([rng.uniform(0, w, n_r), rng.uniform(0, h, n_r)]).astype(np.float32)
    size_r = rng.uniform(2, 32, n_r).astype(np.float32)
    angle_r = rng.uniform(0, 360, n_r).astype(np.float32)
    desc_r = _sift_like_descriptors(rng, n_r)
    dem = np.zeros((h, w), np.uint8) if flat_dem else _smooth_dem(rng, h, w)
    ref = rng.integers(0, 256, size=(h, w), dtype=np.uint8)

    height = rng.uniform(150.0, 400.0)
    centre = np.array([w / 2 + rng.uniform(-60, 60), h / 2 + rng.uniform(-60, 60)])
    yaw = np.deg2rad(rng.uniform(-10, 10))
    tilt = np.deg2rad(rng.uniform(-5, 5, 2))
    R = _rot(yaw, tilt[0], tilt[1])
    cam_pos = np.array([centre[0], centre[1], -height])  # camera above the tile (world z points "down")
    t = (-R @ cam_pos).reshape(3, 1)

    # (uniform(0, h) rounded to float32 can land on h itself -- one pair in a few thousand: the lookup clamps, the coordinate stays)
    z = dem[np.minimum(np.floor(kp_r[:, 1]).astype(int), h - 1), np.minimum(np.floor(kp_r[:, 0]).astype(int), w - 1)].astype(np.float64)
    world = np.column_stack([kp_r.astype(np.float64), z])
    cam = world @ R.T + t.T
    uv = (cam[:, :2] / cam[:, 2:3]) * np.array([K_MATRIX[0, 0], K_MATRIX[1, 1]]) + np.array([K_MATRIX[0, 2], K_MATRIX[1, 2]])
    chosen = rng.permutation(n_r)[: int(match_fraction * n_r)]
    inframe = (cam[chosen, 2] > 1) & (uv[chosen, 0] >= 0) & (uv[chosen, 0] < w) & (uv[chosen, 1] >= 0) & (uv[chosen, 1] < h)
    chosen = chosen[inframe][: n_q]
    k = len(chosen)
    kp_q = np.empty((n_q, 2), np.float32)
    desc_q = np.empty((n_q, 128), np.float32)
    size_q = np.empty(n_q, np.float32)
    angle_q = np.empty(n_q, np.float32)
    gt = np.full(n_q, -1, np.int64)
    scale_ratio = K_MATRIX[0, 0] / height
    kp_q[:k] = (uv[chosen] + rng.normal(0, 0.5, (k, 2))).astype(np.float32)
    desc_q[:k] = np.clip(np.rint(desc_r[chosen] + rng.normal(0, 4.0, (k, 128))), 0, 255).astype(np.float32)
    size_q[:k] = np.clip(size_r[chosen] * scale_ratio, 1.0, 64.0)
    angle_q[:k] = np.mod(angle_r[chosen] - np.rad2deg(yaw) + rng.normal(0, 2.0, k), 360.0)
    gt[:k] = chosen
    nd = n_q - k
    kp_q[k:] = np.column_stack([rng.uniform(0, w, nd), rng.uniform(0, h, nd)])
    desc_q[k:] = _sift_like_descriptors(rng, nd)
    size_q[k:] = rng.uniform(2, 32, nd)
    angle_q[k:] = rng.uniform(0, 360, nd)
    perm = rng.permutation(n_q)
    return Pair(kp_q[perm], desc_q[perm], size_q[perm], angle_q[perm], kp_r, desc_r, size_r, angle_r,
                dem, ref, R, t, gt[perm])


def make_batch(first_index: int, count: int, **kw):
    return [make_pair(first_index + i, **kw) for i in range(count)]


def make_pair_256(pair_index: int, n_q: int = 1024, n_r: int = 1024, h: int = 1080, w: int = 1920, match_fraction: float = 0.7):
    """A frame<->tile pair for a 256-d local feature (BASELINE.json configs[4]: 1920x1080 frames, SuperPoint-style descriptors):
    the geometry of `make_pair` scaled to (w, h), descriptors = unit-norm 256-vectors, matched ones perturbed.  Returns a Pair whose
    desc_* are (N, 256) float32; size / angle are unused by the 256-d matcher (set to 1 / 0)."""
    p = make_pair(pair_index, n_q=n_q, n_r=n_r, h=h, w=w, match_fraction=match_fraction)
    rng = np.random.default_rng(99_000 + pair_index)
    dr = rng.normal(size=(n_r, 256))
    dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    dq = rng.normal(size=(n_q, 256))
    m = p.gt_q2r >= 0
    dq[m] = dr[p.gt_q2r[m]] + 0.25 * rng.normal(size=(int(m.sum()), 256)) / 16.0
    dq /= np.linalg.norm(dq, axis=1, keepdims=True)
    p.desc_q, p.desc_r = dq.astype(np.float32), dr.astype(np.float32)
    p.size_q[:], p.size_r[:], p.angle_q[:], p.angle_r[:] = 1.0, 1.0, 0.0, 0.0
    return p
