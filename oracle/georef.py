"""ORACLE (test infrastructure only).  CPU restatement of PoseNode's post-pose georeferencing (SURVEY.md §8 row a13 /
§8(f) row 4): ros/gisnav/gisnav/core/pose_node.py:333-381 and ros/gisnav/gisnav/_transformations.py:298-393.

Third-party pieces restated from their published algorithms (pyproj, tf_transformations / transforms3d are absent here
-> PARITY UNPINNED): `pyproj` latlong -> geocent on the WGS 84 datum (closed-form geodetic -> ECEF) and
`transforms3d.quaternions.mat2quat` (eigenvector of the largest eigenvalue of the symmetric 4x4 K matrix, w >= 0),
returned in tf_transformations' (x, y, z, w) order.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563


def affine_to_proj(M: np.ndarray) -> str:
    """_transformations.py:274-296 (kept verbatim in behaviour: Python repr of the floats)."""
    return (f"+proj=affine +xoff={M[0, 3]} +yoff={M[1, 3]} +zoff={M[2, 3]} "
            f"+s11={M[0, 0]} +s12={M[0, 1]} +s13={M[0, 2]} +s21={M[1, 0]} +s22={M[1, 1]} +s23={M[1, 2]} "
            f"+s31={M[2, 0]} +s32={M[2, 1]} +s33={M[2, 2]} +no_defs +type=crs +datum=WGS84")


def proj_to_affine(proj_str: str) -> np.ndarray:
    """_transformations.py:298-323."""
    tokens = proj_str.replace("=", " ").split()
    g = lambda k: float(tokens[tokens.index(k) + 1])  # noqa: E731
    return np.array([[g("+s11"), g("+s12"), g("+s13"), g("+xoff")], [g("+s21"), g("+s22"), g("+s23"), g("+yoff")],
                     [g("+s31"), g("+s32"), g("+s33"), g("+zoff")]])


def wgs84_to_ecef(lon: float, lat: float, alt: float) -> Tuple[float, float, float]:
    """_transformations.py:326-345 (pyproj latlong -> geocent, WGS 84)."""
    lam, phi = np.radians(lon), np.radians(lat)
    e2 = WGS84_F * (2.0 - WGS84_F)
    n = WGS84_A / np.sqrt(1.0 - e2 * np.sin(phi) ** 2)
    return (float((n + alt) * np.cos(phi) * np.cos(lam)), float((n + alt) * np.cos(phi) * np.sin(lam)), float((n * (1.0 - e2) + alt) * np.sin(phi)))


def enu_to_ecef_matrix(lon: float, lat: float) -> np.ndarray:
    """_transformations.py:368-393."""
    lon, lat = np.radians(lon), np.radians(lat)
    slat, clat, slon, clon = np.sin(lat), np.cos(lat), np.sin(lon), np.cos(lon)
    return np.array([[-slon, -slat * clon, clat * clon], [clon, -slat * slon, clat * slon], [0, clat, slat]])


def quaternion_from_matrix(m: np.ndarray) -> np.ndarray:
    """tf_transformations.quaternion_from_matrix = transforms3d mat2quat, reordered to (x, y, z, w)."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(m, np.float64)[:3, :3].flat
    K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0], [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                  [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0], [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]              # (w, x, y, z)
    if q[0] < 0:
        q = -q
    return np.array([q[1], q[2], q[3], q[0]])


def pose_to_earth(r: np.ndarray, t: np.ndarray, crs_proj_str: str, ref_shape: Tuple[int, int]) -> Optional[dict]:
    """pose_node.py:333-381: earth-frame position (ECEF) and orientation (x, y, z, w) or None."""
    r_inv = np.asarray(r, np.float64).T
    pos = -r_inv @ np.asarray(t, np.float64).reshape(3, 1)
    x, y = pos[0:2].squeeze().tolist()
    x, y = int(x), int(y)
    if not (0 <= x <= ref_shape[0] and 0 <= y <= ref_shape[1]):
        return None
    affine = proj_to_affine(crs_proj_str)
    t_wgs84 = affine @ np.append(pos, 1)
    ecef = wgs84_to_ecef(*t_wgs84.tolist())
    R = affine[:3, :3]
    R = R / np.linalg.norm(R, axis=0)
    r_ecef = np.eye(4)
    r_ecef[:3, :3] = enu_to_ecef_matrix(t_wgs84[0], t_wgs84[1]) @ (R @ r_inv)
    return dict(position=np.array(ecef), orientation=quaternion_from_matrix(r_ecef), lonlatalt=t_wgs84)
