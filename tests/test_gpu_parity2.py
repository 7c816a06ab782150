"""Round-2 parity tests (-m gpu), closing the blind spots VERDICT r1 named:

  * the matcher ABOVE 1024 keypoints per side (the live reference's cv2.SIFT_create() is unbounded, pose_node.py:122):
    N = 1536 and 2048 against the oracle, indices bit-exact in f32 mode;
  * LOW-MARGIN weights (the blocks rewrite the residual stream several times over, near-ties are common): correspondence
    index mismatches vs the oracle are COUNTED per precision mode -- f32 must be 0, the reduced modes are reported and
    bounded (SURVEY.md section 7 "Hard parts": a tolerance-mode metric, not a bit-exactness claim);
  * planar PnP on 224 seeds: homography start INCLUDING findHomography's 10-step LM polish (on the GPU since r02m; without it the
    agreement was 4e-8) -- inlier count identical, ||dR||, ||dt||/||t|| <= 1e-8, max reported;
  * the f16x2 domain guard: weights scaled so that activations overflow fp16 -> the call is re-run in the exact-split f32x3 mode,
    counted, never silently inf.

Results of the reporting tests are also written to gpurun_out/parity_r02.json.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import oracle_match
from gisnav_amd.synthetic import K_MATRIX, make_pair
from gisnav_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOW_MARGIN = dict(ffn_out_std=4.8e-3, final_scale=4.0, matchability_bias=0.0, matchability_std=0.05)
MID_MARGIN = dict(ffn_out_std=1.2e-3, final_scale=12.0, matchability_bias=2.0, matchability_std=0.05)
MODES = ["f32", "bf16_attn", "f32x3_bf16_attn", "f16x2_bf16_attn", "f16x2_f16_attn"]


def _report(key, value):
    path = os.path.join(ROOT, "gpurun_out", "parity_r02.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[key] = value
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.mark.parametrize("n", [1536, 2048])
def test_matcher_above_1024_keypoints_against_oracle(n, state_dict_np, state_dict_t, dev):
    """npad 1536 / 2048: tile-variant selection of the GEMMs, attention key loops and match-head strips beyond the bench size."""
    from gisnav_amd.engine import PoseEngine
    pairs = [make_pair(300 + n, n_q=n - 37, n_r=n), make_pair(301 + n, n_q=n, n_r=n - 129)]
    ref = [oracle_match(state_dict_t, p) for p in pairs]
    for prec in ("f32", "f16x2_bf16_attn"):
        eng = PoseEngine(0, max_batch=2, max_kpts=n, precision=prec, state_dict=state_dict_np)
        inp = eng.stage_inputs(pairs)
        idx, score, n_match = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        out = eng.estimate(inp, K_MATRIX)
        torch.cuda.synchronize()
        for b, (mq, mr, sc, oidx) in enumerate(ref):
            k = int(n_match[b])
            assert k == len(oidx) > 400, (prec, k, len(oidx))
            assert np.array_equal(idx[b, :k].cpu().numpy(), oidx.numpy()), prec
            assert np.abs(score[b, :k].cpu().numpy() - sc.numpy()[:, 0]).max() < (1e-5 if prec == "f32" else 5e-3)
            assert int(out["ok"][b]) == 1 and np.linalg.norm(out["R"][b].cpu().numpy() - pairs[b].R_gt) < 5e-3
        del eng


def _mismatches(idx_gpu, k_gpu, oidx):
    a = {(int(q), int(r)) for q, r in idx_gpu[:k_gpu]}
    b = {(int(q), int(r)) for q, r in oidx}
    return len(a ^ b), len(b)


@pytest.mark.parametrize("name,kw,th", [("low_margin", LOW_MARGIN, 0.0), ("mid_margin", MID_MARGIN, 0.01)])
def test_low_margin_weights_index_mismatch_counts_per_precision(name, kw, th, dev):
    """Near-tie regime (low_margin: every block rewrites the residual stream several times over, scores ~1e-5, threshold 0 = pure
    mutual nearest neighbour; mid_margin: scores straddle the 0.01 threshold).  f32 mode must reproduce the oracle's correspondences exactly; for the reduced-precision modes the
    symmetric difference of the match sets is counted, reported, and bounded (5 % of the oracle's matches)."""
    from gisnav_amd.engine import PoseEngine
    sd = synthetic_state_dict(0, **kw)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    pairs = [make_pair(400 + i, n_q=512 - 31 * i, n_r=512 - 17 * i) for i in range(4)]
    ref = [oracle_match(tsd, p, filter_threshold=th) for p in pairs]
    assert sum(len(r[3]) for r in ref) > 200, "the low-margin set must still produce matches at this threshold"
    table = {}
    for prec in MODES:
        eng = PoseEngine(0, max_batch=4, max_kpts=512, precision=prec, state_dict=sd, filter_threshold=th)
        inp = eng.stage_inputs(pairs)
        idx, score, n_match = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        torch.cuda.synchronize()
        mism = total = 0
        for b, (_, _, _, oidx) in enumerate(ref):
            d, t = _mismatches(idx[b].cpu().numpy(), int(n_match[b]), oidx.numpy())
            mism += d; total += t
        table[prec] = {"index_mismatches": mism, "oracle_matches": total}
        del eng
    print(name, table)
    _report("index_mismatches_" + name, table)
    assert table["f32"]["index_mismatches"] == 0, table
    # fp16 attention operands (the reference's CUDA arithmetic) keep 3 more significand bits than bf16: never more mismatches than bf16 + 1
    assert table["f16x2_f16_attn"]["index_mismatches"] <= table["f16x2_bf16_attn"]["index_mismatches"] + 1, table
    for prec in MODES[1:]:
        assert table[prec]["index_mismatches"] <= 0.05 * table[prec]["oracle_matches"], table


def test_planar_pnp_homography_start_on_224_seeds(state_dict_np, dev):
    """Flat DEM (TwistNode's zero raster, or a flat tile): solvePnP(ITERATIVE) starts from findHomography(method 0) = normalised
    DLT + the 10-iteration LMSolver polish (both in k_pnp_refine; oracle/pnp_ransac.py:_homography_refine), then the 20-iteration
    pose LM.  Inlier COUNT identical on every seed, pose within 1e-8 (measured max 7.6e-10; it was 4e-8 while the kernel
    skipped the polish and relied on the pose LM reaching the same optimum from a slightly different start)."""
    from gisnav_amd.engine import PoseEngine
    from oracle import pnp_ransac as pr
    eng = PoseEngine(0, max_batch=32, max_kpts=256, precision="f32", state_dict=state_dict_np)
    worst = {"dR": 0.0, "dt": 0.0, "inlier_count_diffs": 0}
    for base in range(0, 224, 32):
        objs, imgs, refs = [], [], []
        for s in range(32):
            seed = 1000 + base + s
            p = make_pair(seed, n_q=256, n_r=256, flat_dem=True)
            q = np.nonzero(p.gt_q2r >= 0)[0]
            mq, mr = p.kp_q[q].copy(), p.kp_r[p.gt_q2r[q]]
            rs = np.random.default_rng(seed)
            no = int(len(q) * rs.uniform(0.0, 0.35))
            mq[:no] = np.column_stack([rs.uniform(0, 640, no), rs.uniform(0, 480, no)]).astype(np.float32)
            mq[no:] += rs.normal(0, rs.uniform(0.0, 1.5), (len(q) - no, 2)).astype(np.float32)      # extra pixel noise: marginal inliers
            obj = np.hstack((mr, np.zeros((len(mr), 1), np.float32))).astype(np.float32)
            o = np.zeros((256, 3), np.float32); o[: len(obj)] = obj
            m = np.zeros((256, 2), np.float32); m[: len(mq)] = mq
            objs.append(o); imgs.append(m)
            refs.append((len(obj), pr.solve_pnp_ransac(obj, mq, K_MATRIX, 10)))
        n_pts = torch.tensor([r[0] for r in refs], dtype=torch.int32, device=dev)
        R, t, ninl, ok = eng.pnp_ransac(torch.from_numpy(np.stack(objs)).to(dev), torch.from_numpy(np.stack(imgs)).to(dev), n_pts, K_MATRIX)
        torch.cuda.synchronize()
        for s, (_, (oko, r, tt, inl)) in enumerate(refs):
            assert bool(oko) == bool(int(ok[s]))
            if not oko:
                continue
            worst["inlier_count_diffs"] += int(int(ninl[s]) != len(inl))
            worst["dR"] = max(worst["dR"], float(np.linalg.norm(R[s].cpu().numpy() - pr.rodrigues_vec2mat(r))))
            worst["dt"] = max(worst["dt"], float(np.linalg.norm(t[s].cpu().numpy() - tt) / np.linalg.norm(tt)))
    print("planar PnP, 224 seeds:", worst)
    _report("planar_pnp_224_seeds", worst)
    assert worst["inlier_count_diffs"] == 0 and worst["dR"] < 1e-8 and worst["dt"] < 1e-8, worst


def test_nonplanar_pnp_dlt_start_on_96_seeds(state_dict_np, dev):
    """DEM with relief: solvePnP(ITERATIVE) starts from the 12 x 12 DLT, whose solution the kernel takes from inverse iteration on
    Cholesky factors (the smallest eigenvector only) where the oracle diagonalises L^T L: outliers, pixel noise and marginal
    inliers as in the planar test -- inlier COUNT identical, pose within 1e-8 (measured 6.4e-10) of the oracle on every seed."""
    from gisnav_amd.engine import PoseEngine
    from oracle import pnp_ransac as pr
    eng = PoseEngine(0, max_batch=32, max_kpts=256, precision="f32", state_dict=state_dict_np)
    worst = {"dR": 0.0, "dt": 0.0, "inlier_count_diffs": 0}
    for base in range(0, 96, 32):
        objs, imgs, refs = [], [], []
        for s in range(32):
            seed = 3000 + base + s
            p = make_pair(seed, n_q=256, n_r=256, flat_dem=False)
            q = np.nonzero(p.gt_q2r >= 0)[0]
            mq, mr = p.kp_q[q].copy(), p.kp_r[p.gt_q2r[q]]
            rs = np.random.default_rng(seed)
            no = int(len(q) * rs.uniform(0.0, 0.35))
            mq[:no] = np.column_stack([rs.uniform(0, 640, no), rs.uniform(0, 480, no)]).astype(np.float32)
            mq[no:] += rs.normal(0, rs.uniform(0.0, 1.5), (len(q) - no, 2)).astype(np.float32)
            x, y = np.floor(mr).astype(int).T
            obj = np.hstack((mr, p.dem[y, x].reshape(-1, 1))).astype(np.float32)
            o = np.zeros((256, 3), np.float32); o[: len(obj)] = obj
            m = np.zeros((256, 2), np.float32); m[: len(mq)] = mq
            objs.append(o); imgs.append(m)
            refs.append((len(obj), pr.solve_pnp_ransac(obj, mq, K_MATRIX, 10)))
        n_pts = torch.tensor([r[0] for r in refs], dtype=torch.int32, device=dev)
        R, t, ninl, ok = eng.pnp_ransac(torch.from_numpy(np.stack(objs)).to(dev), torch.from_numpy(np.stack(imgs)).to(dev), n_pts, K_MATRIX)
        torch.cuda.synchronize()
        for s, (_, (oko, r, tt, inl)) in enumerate(refs):
            assert bool(oko) == bool(int(ok[s]))
            if not oko:
                continue
            worst["inlier_count_diffs"] += int(int(ninl[s]) != len(inl))
            worst["dR"] = max(worst["dR"], float(np.linalg.norm(R[s].cpu().numpy() - pr.rodrigues_vec2mat(r))))
            worst["dt"] = max(worst["dt"], float(np.linalg.norm(t[s].cpu().numpy() - tt) / np.linalg.norm(tt)))
    print("non-planar PnP, 96 seeds:", worst)
    _report("nonplanar_pnp_96_seeds", worst)
    assert worst["inlier_count_diffs"] == 0 and worst["dR"] < 1e-8 and worst["dt"] < 1e-8, worst


def test_f16x2_domain_guard_flags_overflow_and_falls_back(state_dict_np, state_dict_t, dev):
    """One out_proj scaled so that the message leaves fp16's range (|msg| >> 65504; LayerNorm brings the stream back, so the
    f32 oracle is unimpressed).  Guard 'flag': the call reports ZERO matches (never inf / NaN garbage) and the status says so.
    Guard 'sync': the call is re-run with exact three-term bf16 splits and returns the oracle's correspondences.  Unscaled
    weights never trip."""
    from gisnav_amd.engine import PoseEngine
    sd = dict(state_dict_np)
    for leaf in ("weight", "bias"):
        k = f"transformers.3.self_attn.out_proj.{leaf}"
        sd[k] = state_dict_np[k] * np.float32(3.0e5)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    pairs = [make_pair(500 + i, n_q=256 - 9 * i, n_r=250) for i in range(2)]
    ref = [oracle_match(tsd, p) for p in pairs]
    assert all(len(r[3]) > 50 for r in ref)
    eng = PoseEngine(0, max_batch=2, max_kpts=256, precision="f16x2_bf16_attn", state_dict=sd, guard="flag")
    inp = eng.stage_inputs(pairs)
    # Since round 4 out_proj is COMPOSED into ffn.0's weights at load time (the message never exists as fp16 rows, and the composed matrix gets its own
    # power-of-two scale): this weight set stays inside the domain and the default path returns the oracle's correspondences without any fallback.
    idx, score, n_match = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    torch.cuda.synchronize()
    assert eng.guard_status() == (False, 0)
    for b, (_, _, sc, oidx) in enumerate(ref):
        d, t = _mismatches(idx[b].cpu().numpy(), int(n_match[b]), oidx.numpy())
        assert d <= 0.02 * t, (d, t)
    eng.lib.gn_debug_set_variant(eng.ctx, 28, 0)      # the kernel computes the message itself (rounds 2-3): it leaves fp16's range
    idx, score, n_match = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    assert eng.guard_status() == (True, 1)
    assert (n_match.cpu().numpy() == 0).all()
    out = eng.estimate(inp, K_MATRIX)
    assert (out["ok"].cpu().numpy() == 0).all() and (out["n_match"].cpu().numpy() == 0).all()
    del eng
    for knobs in (((28, 0),), ((13, 0),), ((10, 1),)):       # message computed in the block-tail kernel, separate out_proj launch, the three-launch tail
        eng = PoseEngine(0, max_batch=2, max_kpts=256, precision="f16x2_bf16_attn", state_dict=sd, guard="sync")
        for which, value in knobs:
            eng.lib.gn_debug_set_variant(eng.ctx, which, value)
        idx, score, n_match = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        torch.cuda.synchronize()
        assert eng.guard_status()[1] == 1, knobs
        for b, (_, _, sc, oidx) in enumerate(ref):
            k = int(n_match[b])
            d, t = _mismatches(idx[b].cpu().numpy(), k, oidx.numpy())
            assert d <= 0.02 * t, (knobs, d, t)
        del eng
    eng = PoseEngine(0, max_batch=2, max_kpts=256, precision="f16x2_bf16_attn", state_dict=state_dict_np, guard="sync")
    idx, score, n_match = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    assert eng.guard_status() == (False, 0) and (n_match.cpu().numpy() > 50).all()


@pytest.mark.gpu
@pytest.mark.parametrize("flat", [False, True], ids=["relief", "flat"])
def test_pnp_with_more_inliers_than_the_refinement_keeps_in_lds(state_dict_np, dev, flat):
    """k_pnp_refine compacts the inliers into LDS (2048 correspondences); a pair with more of them goes through the same passes
    on the context's global scratch (PnpArgs.pts_ws).  ~2900 correspondences, 8 % gross outliers, against the oracle: inlier count
    identical, pose within 1e-8."""
    from gisnav_amd.engine import PoseEngine
    from oracle import pnp_ransac as pr
    K = 4096
    eng = PoseEngine(0, max_batch=2, max_kpts=K, precision="f32", state_dict=state_dict_np)
    objs, imgs, refs = [], [], []
    for s in range(2):
        seed = 7000 + s
        p = make_pair(seed, n_q=K, n_r=K, flat_dem=flat)
        q = np.nonzero(p.gt_q2r >= 0)[0]
        mq, mr = p.kp_q[q].copy(), p.kp_r[p.gt_q2r[q]]
        rs = np.random.default_rng(seed)
        no = int(len(q) * 0.08)
        sel = rs.choice(len(q), no, replace=False)                       # outliers scattered through the list, not a prefix
        mq[sel] = np.column_stack([rs.uniform(0, 640, no), rs.uniform(0, 480, no)]).astype(np.float32)
        mq += rs.normal(0, 0.5, mq.shape).astype(np.float32)
        x, y = np.floor(mr).astype(int).T
        obj = np.hstack((mr, p.dem[y, x].reshape(-1, 1))).astype(np.float32)
        o = np.zeros((K, 3), np.float32); o[: len(obj)] = obj
        m = np.zeros((K, 2), np.float32); m[: len(mq)] = mq
        objs.append(o); imgs.append(m)
        refs.append((len(obj), pr.solve_pnp_ransac(obj, mq, K_MATRIX, 10)))
    n_pts = torch.tensor([r[0] for r in refs], dtype=torch.int32, device=dev)
    R, t, ninl, ok = eng.pnp_ransac(torch.from_numpy(np.stack(objs)).to(dev), torch.from_numpy(np.stack(imgs)).to(dev), n_pts, K_MATRIX)
    torch.cuda.synchronize()
    for s, (_, (oko, r, tt, inl)) in enumerate(refs):
        assert oko and int(ok[s]) == 1
        assert len(inl) > 2048, len(inl)                                 # the case under test
        assert int(ninl[s]) == len(inl)
        dR = float(np.linalg.norm(R[s].cpu().numpy() - pr.rodrigues_vec2mat(r)))
        dt = float(np.linalg.norm(t[s].cpu().numpy() - tt) / np.linalg.norm(tt))
        assert dR < 1e-8 and dt < 1e-8, (s, dR, dt)
