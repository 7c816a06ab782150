"""Seam B2: `compute_pose(camera_info, mkp_qry, mkp_ref, elevation)`.

Mirror of ros/gisnav/gisnav/core/_shared.py:89-125 (also used by TwistNode with a zero DEM,
core/twist_node.py:289): numpy in, `(R (3,3) f64, t (3,1) f64)` out.  The DEM lookup is the
host-side marshalling of the reference's `_compute_3d_points`; RANSAC, EPnP, the iterative refinement
and Rodrigues run in `gn_pnp_ransac` on the GPU.  Returns None where the reference would fail
(cv2 returning no model).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from .engine import PoseEngine, RANSAC_ITERATIONS

_default_engine: Optional[PoseEngine] = None


def _engine(max_pts: int) -> PoseEngine:
    global _default_engine
    if _default_engine is None or _default_engine.kmax < max_pts:
        _default_engine = PoseEngine(0, max_batch=1, max_kpts=max(max_pts, 1024))
    return _default_engine


def init(device: int = 0, max_points: int = 4096) -> PoseEngine:
    """Create the module-level solver context ahead of time (otherwise the first compute_pose call pays the hipMalloc inside
    the ROS callback)."""
    global _default_engine
    if _default_engine is None or _default_engine.kmax < max_points or (_default_engine.device.index or 0) != device:
        _default_engine = PoseEngine(device, max_batch=1, max_kpts=max(max_points, 1024))
    return _default_engine


def compute_pose(camera_info, mkp_qry: np.ndarray, mkp_ref: np.ndarray, elevation: Optional[np.ndarray],
                 engine: Optional[PoseEngine] = None) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    n = len(mkp_qry)
    if n < 4:                     # cv2.solvePnPRansac asserts npoints >= 4 (the reference would raise); the shim reports "no pose"
        return None
    eng = engine or _engine(n)
    # object points = reference keypoint + the DEM cell under it (z = 0 without a raster: TwistNode), marshalled straight into the f32 [n][3]
    # array gn_pnp_ransac reads -- the host-side twin of gn_gather_points' lift (reference: _shared.py:95-102)
    obj = np.zeros((n, 3), np.float32)
    obj[:, :2] = mkp_ref
    if elevation is not None:
        cell = np.floor(np.asarray(mkp_ref)).astype(np.intp)
        obj[:, 2] = np.asarray(elevation)[cell[:, 1], cell[:, 0]]
    img = np.ascontiguousarray(mkp_qry, dtype=np.float32)
    k_matrix = np.asarray(camera_info.k, dtype=np.float64).reshape((3, 3))
    R, t, _, ok = eng.pnp_ransac_host(obj, img, k_matrix, RANSAC_ITERATIONS, min_pts=4)   # n == 4: OpenCV's P3P branch
    if not ok:
        return None
    return R, t
