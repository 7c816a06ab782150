"""Developer tool: StereoNode's reference raster (BGR -> gray + DEM, rotate, centre crop) against its oracle over random tile sizes (odd ones too),
crop sizes (smaller than / equal to the tile) and angles: every u8 pixel identical.  A crop LARGER than the tile must be refused (GnError): the
reference's numpy slice `rotated[dy:dy + h, dx:dx + w]` with a negative start wraps around and returns a differently shaped (often empty) array there,
which nothing downstream can use; GISNav's tiles are always padded beyond the crop.   python tests/sweeps/fuzz_stereo.py [trials]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import stereo_warp as sw  # noqa: E402   (checker, as in tests/)
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.stereo import stereo_reference  # noqa: E402
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(77)
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f32")
bad = 0
for t in range(trials):
    h, w = int(rng.integers(17, 900)), int(rng.integers(17, 1100))
    ch, cw = (int(rng.integers(8, h + 40)), int(rng.integers(8, w + 40))) if t % 4 else (480, 640)
    angle = float(rng.choice([rng.uniform(-400, 400), rng.choice([0.0, 90.0, 180.0, 270.0, 360.0, -90.0, 45.0])]))
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    dem = rng.integers(0, 256, (h, w), dtype=np.uint8)
    try:
        oref, odem, ominv = sw.stereo_reference(bgr, dem, angle, (ch, cw))
    except Exception as e:  # noqa: BLE001
        oref = None; oerr = e
    try:
        ref, dm, minv = stereo_reference(eng, bgr, dem, angle, (ch, cw))
        got = (ref.cpu().numpy(), dm.cpu().numpy())
    except Exception as e:  # noqa: BLE001
        got = None; gerr = e
    if ch > h or cw > w:
        same = got is None and "gn_stereo_reference" in str(gerr)
        if not same:
            print(f"trial {t}: tile {h}x{w} crop {ch}x{cw}: a crop larger than the tile was NOT refused MISMATCH", flush=True)
    elif oref is None or got is None:
        same = oref is None and got is None
        print(f"trial {t}: tile {h}x{w} crop {ch}x{cw} angle {angle:.3f}: oracle {'raises ' + type(oerr).__name__ if oref is None else 'ok'}, here {'raises ' + type(gerr).__name__ + ': ' + str(gerr)[:80] if got is None else 'ok'} {'(both refuse)' if same else 'MISMATCH'}", flush=True)
    else:
        same = got[0].shape == oref.shape and np.array_equal(got[0], oref) and np.array_equal(got[1], odem) and np.allclose(minv, ominv, rtol=0, atol=1e-9)
        if not same:
            print(f"trial {t}: tile {h}x{w} crop {ch}x{cw} angle {angle:.3f}: {int((got[0] != oref).sum()) if got[0].shape == oref.shape else 'shape'} gray / {int((got[1] != odem).sum()) if got[1].shape == odem.shape else 'shape'} dem pixels differ MISMATCH", flush=True)
    bad += not same
print(f"{trials} trials, mismatching: {bad}")
sys.exit(1 if bad else 0)
