"""Developer tool: the B2 seam (compute_pose -> gn_pnp_ransac) against the oracle on inputs the test-suite's seeds do not hold: few points (5 .. 14),
heavy outlier fractions, duplicated points, collinear image points, a flat DEM, points on a plane tilted against the DEM, large coordinates.  Same
None / pose decision and poses within 1e-6 (the suite's bar is 1e-8 on well-posed scenes).  Scenes with DUPLICATED correspondences are reported, not
asserted: a 5-point RANSAC sample that holds a point twice is a rank-deficient EPnP system (OpenCV's PnP callback has no subset check either), its
null-space basis is decided by rounding, and two correct implementations pick different hypotheses from it.   python tests/sweeps/fuzz_pnp.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pnp_ransac as pr  # noqa: E402   (checker, as in tests/)
from gisnav_amd.pose import compute_pose  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402


class Cam:
    k = K_MATRIX.reshape(-1)


def scene(seed, n, outliers=0.0, flat=False, dup=0, collinear=False, scale=1.0):
    rng = np.random.default_rng(seed)
    p = make_pair(9000 + seed, n_q=max(n, 64), n_r=max(4 * n, 256), flat_dem=flat)
    m = np.nonzero(p.gt_q2r >= 0)[0][:n]
    q = p.kp_q[m].astype(np.float32).copy(); r = p.kp_r[p.gt_q2r[m]].astype(np.float32).copy()
    k = int(outliers * len(q))
    if k: q[rng.permutation(len(q))[:k]] = rng.uniform(0, 480, (k, 2)).astype(np.float32)
    if dup:
        q[-dup:], r[-dup:] = q[0], r[0]
    if collinear:
        r[:, 1] = r[0, 1] + 0.1 * (r[:, 0] - r[0, 0])       # reference points on a line
    return q * scale, r, p.dem


cases = [("n=%d" % n, scene(n, n)) for n in (4, 5, 6, 7, 9, 14)]
cases += [("n=60, 50 %% outliers, seed %d" % s, scene(20 + s, 60, outliers=0.5)) for s in range(4)]
cases += [("n=40, 80 % outliers", scene(30, 40, outliers=0.8)), ("n=30 flat DEM", scene(31, 30, flat=True)), ("n=12 flat DEM", scene(32, 12, flat=True)),
          ("n=30, 10 duplicates", scene(33, 30, dup=10)), ("n=8, 4 duplicates", scene(34, 8, dup=4)), ("n=25 collinear reference points", scene(35, 25, collinear=True)),
          ("n=200", scene(36, 200)), ("n=1000", scene(37, 1000))]
bad = 0
for name, (q, r, dem) in cases:
    want = pr.compute_pose(K_MATRIX.reshape(-1), q, r, dem)
    got = compute_pose(Cam, q, r, dem)
    if (want is None) != (got is None):
        print(f"{name} ({len(q)} points): oracle {'None' if want is None else 'pose'}, here {'None' if got is None else 'pose'}  MISMATCH"); bad += 1; continue
    if want is None:
        print(f"{name} ({len(q)} points): None on both sides"); continue
    dR, dt = float(np.abs(want[0] - got[0]).max()), float(np.abs(want[1] - got[1]).max() / max(1.0, np.abs(want[1]).max()))
    ok = dR < 1e-6 and dt < 1e-6 and np.isfinite(got[0]).all()
    report_only = "duplicates" in name
    bad += (not ok) and not report_only
    print(f"{name} ({len(q)} points): |dR| {dR:.2e}, |dt| rel {dt:.2e} {'ok' if ok else ('differs (rank-deficient samples: reported only)' if report_only else 'MISMATCH')}", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
