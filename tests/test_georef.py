"""Post-pose georeferencing (pose_node.py:333-381): oracle known answers and the library's host-side C code against the
oracle.  Host code only -- runs without a GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import georef as og  # noqa: E402
from oracle import pnp_ransac as pr  # noqa: E402


def _crs(lon0=24.94, lat0=60.17, rot_deg=12.0, mpp=1.0):
    """A plausible OrthoStereoImage CRS: rotated / scaled pixel grid -> (lon, lat, alt) near Helsinki, z flipped."""
    a = np.radians(rot_deg)
    dlat = mpp / 111_320.0; dlon = dlat / np.cos(np.radians(lat0))
    M = np.array([[np.cos(a) * dlon, np.sin(a) * dlon, 0.0, lon0], [np.sin(a) * dlat, -np.cos(a) * dlat, 0.0, lat0], [0.0, 0.0, -mpp, 12.5]])
    return og.affine_to_proj(M), M


def test_oracle_known_answers():
    assert np.allclose(og.wgs84_to_ecef(0, 0, 0), (og.WGS84_A, 0, 0), atol=1e-6)
    assert np.allclose(og.wgs84_to_ecef(90, 0, 100), (0, og.WGS84_A + 100, 0), atol=1e-6)
    b = og.WGS84_A * (1 - og.WGS84_F)
    assert np.allclose(og.wgs84_to_ecef(0, 90, 0), (0, 0, b), atol=1e-6)
    assert np.allclose(og.quaternion_from_matrix(np.eye(4)), [0, 0, 0, 1])
    rz = pr.rodrigues_vec2mat(np.array([0, 0, np.pi / 2]))
    assert np.allclose(og.quaternion_from_matrix(rz), [0, 0, np.sqrt(0.5), np.sqrt(0.5)])
    s, M = _crs()
    assert np.array_equal(og.proj_to_affine(s), M)                # repr round trip is exact
    e = og.enu_to_ecef_matrix(0, 0)
    assert np.allclose(e @ np.array([0, 0, 1.0]), [1, 0, 0])      # "up" at (0, 0) is ECEF +x


def test_library_georeferencing_matches_oracle():
    from gisnav_amd import georef as gg
    rng = np.random.default_rng(0)
    s, M = _crs()
    assert np.array_equal(gg.proj_to_affine(s), M)
    for lon, lat, alt in [(24.94, 60.17, 120.0), (-122.3, 37.5, -3.0), (179.9, -89.0, 9000.0), (0.0, 0.0, 0.0)]:
        assert np.allclose(gg.wgs84_to_ecef(lon, lat, alt), og.wgs84_to_ecef(lon, lat, alt), rtol=0, atol=1e-8)
    n_some = 0
    for i in range(200):
        rv = rng.normal(0, 0.4, 3); rv[2] += rng.uniform(-3, 3)
        r = pr.rodrigues_vec2mat(rv)
        cam = np.array([rng.uniform(-40, 520), rng.uniform(-40, 680), -rng.uniform(80, 400)])     # camera centre in raster coordinates
        t = -r @ cam.reshape(3, 1)
        o = og.pose_to_earth(r, t, s, (480, 640))
        g = gg.pose_to_earth(r, t, s, (480, 640))
        assert (o is None) == (g is None), i
        if o is None:
            continue
        n_some += 1
        assert np.allclose(g["position"], o["position"], rtol=0, atol=1e-7)          # metres
        assert np.allclose(g["lonlatalt"], o["lonlatalt"], rtol=0, atol=1e-12)
        assert min(np.abs(g["orientation"] - o["orientation"]).max(), np.abs(g["orientation"] + o["orientation"]).max()) < 1e-10
        assert abs(np.linalg.norm(g["orientation"]) - 1) < 1e-12 and g["orientation"][3] >= 0
    assert 50 < n_some < 200                                         # both branches (pose and None) exercised


def test_library_rejects_malformed_proj_strings():
    import pytest
    from gisnav_amd import georef as gg
    with pytest.raises(ValueError):
        gg.proj_to_affine("+proj=affine +xoff=1 +yoff=2")
