"""StereoNode reference-raster preparation: oracle KATs on the CPU, the HIP kernel against the
oracle through the C ABI on the GPU (bit-exact u8 pixels)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stereo_warp as sw  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _tile(seed, h=700, w=900):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (96 + 80 * np.sin(xx / 37.0) * np.cos(yy / 23.0)).astype(np.int64)
    bgr = np.clip(base[..., None] + rng.integers(-40, 41, (h, w, 3)), 0, 255).astype(np.uint8)
    dem = np.clip(20 + 15 * np.sin(xx / 90.0 + yy / 70.0) + rng.integers(0, 3, (h, w)), 0, 255).astype(np.uint8)
    return bgr, dem


# ------------------------------------------------------------------ oracle known answers (CPU)
def test_oracle_gray_coefficients_and_rounding():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 20, 30], [1, 1, 1]]], np.uint8)   # B, G, R
    assert sw.bgr2gray_u8(px)[0].tolist() == [29, 150, 76, 255, 22, 1]


def test_oracle_identity_and_quarter_turns_are_exact_permutations():
    bgr, dem = _tile(0, 64, 96)
    st = np.dstack((sw.bgr2gray_u8(bgr), dem))
    eye = np.array([[1, 0, 0], [0, 1, 0]], np.float64)
    assert np.array_equal(sw.warp_affine_u8(st, eye, (96, 64)), st)
    sq = st[:, :64]                                            # square, even size: centre (32, 32)
    m = sw.get_rotation_matrix_2d((32, 32), 90.0, 1.0)
    r = sw.warp_affine_u8(sq, m, (64, 64))
    # positive angle = counter-clockwise: dst(x, y) = src(64 - y, x); row/col 0 of the source fall off the frame
    exp = np.zeros_like(sq)
    for y in range(1, 64):
        exp[y, :, :] = sq[:, 64 - y, :]
    assert np.array_equal(r[1:], exp[1:])
    assert (r[0] == 0).all()                                   # constant border


def test_oracle_half_pixel_shift_is_the_fixed_point_average():
    src = np.array([[10, 20, 31, 255]], np.uint8)
    m = np.array([[1, 0, 0.5], [0, 1, 0]], np.float64)       # dst(x) = src(x - 0.5)
    out = sw.warp_affine_u8(src, m, (4, 1))
    assert out[0].tolist() == [5, 15, 26, 143]                  # (0+10)/2, (10+20)/2, (20+31+1)>>1, (31+255)/2


def test_oracle_rotate_and_crop_matrix_maps_back_to_the_original_frame():
    bgr, dem = _tile(1)
    st = np.dstack((sw.bgr2gray_u8(bgr), dem))
    out, minv = sw.rotate_and_crop_center(st, 35.0, (480, 640))
    assert out.shape == (480, 640, 2)
    full = sw.warp_affine_u8(st, sw.get_rotation_matrix_2d((450, 350), 35.0, 1.0), (900, 700))
    assert np.array_equal(out, full[110:590, 130:770])          # crop of the full rotation
    p = minv @ np.array([320.0, 240.0, 1.0])                    # crop centre -> image centre
    assert np.allclose(p[:2], [450.0, 350.0], atol=1e-9)


def test_golden_stereo_fixture_matches_the_oracle():
    g = np.load(os.path.join(GOLD, "stereo_rot_seed3.npz"))
    ref, dem, minv = sw.stereo_reference(g["bgr"], g["dem"], float(g["angle"]), (int(g["crop"][0]), int(g["crop"][1])))
    assert np.array_equal(ref, g["ref"]) and np.array_equal(dem, g["dem_out"])
    assert np.allclose(minv, g["minv"], rtol=0, atol=1e-12)


# ------------------------------------------------------------------ HIP kernel vs oracle (GPU)
@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from gisnav_amd.engine import PoseEngine
    return PoseEngine(0, max_batch=1, max_kpts=128, precision="f32")


@pytest.mark.gpu
@pytest.mark.parametrize("angle", [0.0, 5.0, 35.0, 90.0, 177.5, 270.0, 355.0, -12.25])
def test_stereo_reference_bit_exact(eng, angle):
    from gisnav_amd.stereo import rotate_and_crop_center, stereo_reference
    bgr, dem = _tile(3)
    ref, dm, minv = stereo_reference(eng, bgr, dem, angle, (480, 640))
    oref, odem, ominv = sw.stereo_reference(bgr, dem, angle, (480, 640))
    assert np.array_equal(ref.cpu().numpy(), oref)                    # every u8 pixel identical
    assert np.array_equal(dm.cpu().numpy(), odem)
    assert np.allclose(minv, ominv, rtol=0, atol=1e-9)
    st = np.dstack((sw.bgr2gray_u8(bgr), dem))
    out, m2 = rotate_and_crop_center(eng, st, angle, (480, 640))       # the 2-channel entry point
    assert np.array_equal(out.cpu().numpy(), np.dstack((oref, odem)))
    assert np.allclose(m2, ominv, rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_stereo_crop_as_large_as_the_tile_shows_the_zero_border(eng):
    from gisnav_amd.stereo import stereo_reference
    bgr, dem = _tile(4, 480, 640)
    ref, dm, _ = stereo_reference(eng, bgr, dem, 30.0, (480, 640))
    oref, odem, _ = sw.stereo_reference(bgr, dem, 30.0, (480, 640))
    assert np.array_equal(ref.cpu().numpy(), oref) and np.array_equal(dm.cpu().numpy(), odem)
    assert (oref[0, :10] == 0).all()                                    # corners rotate out of the source


@pytest.mark.gpu
def test_stereo_golden_fixture_through_c_abi(eng):
    from gisnav_amd.stereo import stereo_reference
    g = np.load(os.path.join(GOLD, "stereo_rot_seed3.npz"))
    ref, dm, minv = stereo_reference(eng, g["bgr"], g["dem"], float(g["angle"]), (int(g["crop"][0]), int(g["crop"][1])))
    assert np.array_equal(ref.cpu().numpy(), g["ref"]) and np.array_equal(dm.cpu().numpy(), g["dem_out"])
    assert np.allclose(minv, g["minv"], rtol=0, atol=1e-9)
