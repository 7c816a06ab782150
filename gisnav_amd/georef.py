"""Host-side mirror of PoseNode's post-pose georeferencing (pose_node.py:333-381, _transformations.py:298-393):
thin ctypes calls into the library's host-side C code (`gn_proj_to_affine`, `gn_pose_to_earth`)."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import _lib


def proj_to_affine(proj_str: str) -> np.ndarray:
    out = np.zeros(12, np.float64)
    rc = _lib.load().gn_proj_to_affine(proj_str.encode(), out.ctypes.data_as(_lib.c_f64p))
    if rc != 0:
        raise ValueError(f"not a '+proj=affine' string with the twelve coefficients ({rc})")
    return out.reshape(3, 4)


def wgs84_to_ecef(lon: float, lat: float, alt: float) -> Tuple[float, float, float]:
    out = np.zeros(3, np.float64)
    _lib.load().gn_wgs84_to_ecef(float(lon), float(lat), float(alt), out.ctypes.data_as(_lib.c_f64p))
    return float(out[0]), float(out[1]), float(out[2])


def pose_to_earth(r: np.ndarray, t: np.ndarray, crs_proj_str: str, ref_shape: Tuple[int, int]) -> Optional[dict]:
    """(r, t) of compute_pose + `msg.crs.data` -> dict(position ECEF [3], orientation (x, y, z, w), lonlatalt) or None."""
    aff = np.ascontiguousarray(proj_to_affine(crs_proj_str).reshape(12))
    R9 = np.ascontiguousarray(np.asarray(r, np.float64).reshape(9)); t3 = np.ascontiguousarray(np.asarray(t, np.float64).reshape(3))
    pos, q, lla = np.zeros(3), np.zeros(4), np.zeros(3)
    p = lambda a: a.ctypes.data_as(_lib.c_f64p)  # noqa: E731
    rc = _lib.load().gn_pose_to_earth(p(R9), p(t3), p(aff), int(ref_shape[0]), int(ref_shape[1]), p(pos), p(q), p(lla))
    if rc == 1:
        return None
    if rc != 0:
        raise _lib.GnError(f"gn_pose_to_earth failed ({rc})")
    return dict(position=pos, orientation=q, lonlatalt=lla)
