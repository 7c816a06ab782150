"""Round-3 parity tests (-m gpu), closing what VERDICT r2 named:

  * the FULL-SIZE oracle check is no longer a spot check: all 32 bench pairs in f32 mode, and 8 pairs of each FAST mode
    (f16x2_bf16_attn, f16x2_f16_attn) compared DIRECTLY with the oracle (not with the f32 GPU result) -- indices identical,
    scores 5e-3, pose 1e-8;
  * the fp16-attention precision mode (GN_PREC_F16X2_F16_ATTN = the reference's CUDA arithmetic): kernel against fp64, the
    optimistic-reference fallback at fp16's range, whole matcher against fixtures, low-margin mismatch counts beside bf16's;
  * configs[4] at ITS size: the SuperPoint extractor at 1920x1080 (batch 2, k = 1024) against the oracle and against a
    transformers-generated fixture, f32 and split-fp16 arithmetic: the keypoint sets are EQUAL except for keypoints whose
    decision margin (distance of the score from the top-k cut-off) is below the stated score tolerance, and those are counted;
  * per-group guard words of sub-batch streams: a trip in one group zeroes that group's pairs only;
  * PoseEngine.grow() replays sticky context state;
  * RCCL: a world-1 nccl process group on cuda:0 carries the broadcast / all_gather / all_reduce the N > 1 bench uses, and
    `bench.py --gpus 2` on a 1-GPU box exits non-zero instead of silently running one rank.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import oracle_match
from gisnav_amd.synthetic import K_MATRIX, make_pair
from gisnav_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FAST_MODES = ["f16x2_bf16_attn", "f16x2_f16_attn"]


def _report(key, value):
    path = os.path.join(ROOT, "gpurun_out", "parity_r03.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[key] = value
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ------------------------------------------------------------------ full size, directly against the oracle
@pytest.fixture(scope="module")
def bench_pairs():
    return [make_pair(i) for i in range(32)]


@pytest.fixture(scope="module")
def oracle_32(bench_pairs, state_dict_t):
    """The oracle on ALL 32 bench pairs (1024 keypoints per side): matches + pose.  ~10-20 s of host time."""
    from oracle import pnp_ransac as pr
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    out = []
    for p in bench_pairs:
        mq, mr, sc, oidx = oracle_match(state_dict_t, p)
        R, t = pr.compute_pose(K_MATRIX.reshape(-1), mq.numpy(), mr.numpy(), p.dem)
        out.append((oidx.numpy(), sc.numpy()[:, 0], R, t))
    return out


def _run_mode(prec, pairs, sd):
    from gisnav_amd.engine import PoseEngine
    eng = PoseEngine(0, max_batch=len(pairs), max_kpts=1024, precision=prec, state_dict=sd)
    inp = eng.stage_inputs(pairs)
    idx, score, nm = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    out = eng.estimate(inp, K_MATRIX)
    torch.cuda.synchronize()
    res = (idx.cpu().numpy(), score.cpu().numpy(), nm.cpu().numpy(), {k: v.cpu().numpy() for k, v in out.items()})
    del eng
    return res


def test_full_size_all_32_pairs_f32_against_oracle(bench_pairs, oracle_32, state_dict_np):
    idx, score, nm, out = _run_mode("f32", bench_pairs, state_dict_np)
    worst = [0.0, 0.0, 0.0]
    for b, (oidx, osc, Ro, to) in enumerate(oracle_32):
        k = int(nm[b])
        assert k == len(oidx) and np.array_equal(idx[b, :k], oidx), b            # correspondence indices: bit-exact
        worst[0] = max(worst[0], float(np.abs(score[b, :k] - osc).max()))
        worst[1] = max(worst[1], float(np.linalg.norm(out["R"][b] - Ro)))
        worst[2] = max(worst[2], float(np.linalg.norm(out["t"][b] - to) / np.linalg.norm(to)))
        assert out["ok"][b] == 1
    _report("full_size_f32_vs_oracle_32_pairs", {"max_score_err": worst[0], "max_dR": worst[1], "max_dt_rel": worst[2]})
    assert worst[0] < 1e-5 and worst[1] < 1e-8 and worst[2] < 1e-8, worst


@pytest.mark.parametrize("prec", FAST_MODES)
def test_full_size_fast_modes_directly_against_oracle(prec, bench_pairs, oracle_32, state_dict_np):
    """The bench's fast precision modes at the bench's size, against the ORACLE (not against the f32 GPU result): 8 of the 32 pairs
    asserted (every fourth), all 32 counted in the parity report."""
    idx, score, nm, out = _run_mode(prec, bench_pairs, state_dict_np)
    mism = 0
    worst = [0.0, 0.0, 0.0]
    for b, (oidx, osc, Ro, to) in enumerate(oracle_32):
        k = int(nm[b])
        same = k == len(oidx) and np.array_equal(idx[b, :k], oidx)
        mism += 0 if same else len({tuple(r) for r in idx[b, :k].tolist()} ^ {tuple(r) for r in oidx.tolist()})
        if b % 4 == 0:
            assert same, (prec, b)
            assert np.abs(score[b, :k] - osc).max() < 5e-3
            assert np.linalg.norm(out["R"][b] - Ro) < 1e-8 and np.linalg.norm(out["t"][b] - to) / np.linalg.norm(to) < 1e-8
        if same:
            worst[0] = max(worst[0], float(np.abs(score[b, :k] - osc).max()))
            worst[1] = max(worst[1], float(np.linalg.norm(out["R"][b] - Ro)))
            worst[2] = max(worst[2], float(np.linalg.norm(out["t"][b] - to) / np.linalg.norm(to)))
    _report("full_size_" + prec + "_vs_oracle_32_pairs", {"index_mismatches": mism, "max_score_err": worst[0], "max_dR": worst[1], "max_dt_rel": worst[2]})
    assert mism == 0, (prec, mism)


# ------------------------------------------------------------------ fp16 attention mode
@pytest.fixture(scope="module")
def eng_f16(state_dict_np):
    from gisnav_amd.engine import PoseEngine
    return PoseEngine(0, max_batch=4, max_kpts=256, precision="f16x2_f16_attn", state_dict=state_dict_np)


@pytest.fixture(scope="module")
def eng_bf16(state_dict_np):
    from gisnav_amd.engine import PoseEngine
    return PoseEngine(0, max_batch=4, max_kpts=256, precision="f16x2_bf16_attn", state_dict=state_dict_np)


def _attn_ref(q, k, v, nkv, cross, cast):
    BS, n, _ = q.shape
    outs = []
    for bs in range(BS):
        kvs = bs ^ 1 if cross else bs
        m = int(nkv[kvs])
        qq = cast(q[bs] * 0.125).double().cpu().reshape(n, 4, 64).transpose(0, 1)
        kk = cast(k[kvs, :m]).double().cpu().reshape(m, 4, 64).transpose(0, 1)
        vv = cast(v[kvs, :m]).double().cpu().reshape(m, 4, 64).transpose(0, 1)
        outs.append((torch.softmax(qq @ kk.transpose(1, 2), -1) @ vv).transpose(0, 1).reshape(n, 256).numpy())
    return outs


@pytest.mark.parametrize("cross", [False, True])
def test_fp16_attention_against_fp64_and_tighter_than_bf16(eng_f16, eng_bf16, cross):
    n, BS = 256, 4
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(3)
    q, k, v = (torch.randn(BS, n, 256, generator=g).to(dev) for _ in range(3))
    k[1, 7] *= 6.0
    nkv = torch.tensor([256, 219, 5, 130], dtype=torch.int32, device=dev)
    o16 = eng_f16.debug_attention(q, k, v, nkv, cross, 0.125).cpu().numpy()
    ob = eng_bf16.debug_attention(q, k, v, nkv, cross, 0.125).cpu().numpy()
    ref = _attn_ref(q, k, v, nkv, cross, lambda t: t)
    e16 = max(_rel(o16[bs], ref[bs]) for bs in range(BS))
    eb = max(_rel(ob[bs], ref[bs]) for bs in range(BS))
    _report(f"attention_rel_err_vs_fp64_cross{int(cross)}", {"fp16": e16, "bf16": eb})
    assert e16 < 3e-3 and eb < 2e-2 and e16 < 0.5 * eb, (e16, eb)      # three more significand bits: ~8x closer (measured)
    ref16 = _attn_ref(q, k, v, nkv, cross, lambda t: t.half())
    assert max(_rel(o16[bs], ref16[bs]) for bs in range(BS)) < 1.5e-3    # against fp64 on the fp16-rounded operands: only P's rounding is left


@pytest.mark.parametrize("boost", [1.0, 6.0, 40.0])
def test_fp16_attention_optimistic_reference_falls_back_at_fp16_range(eng_f16, boost):
    """fp16 probabilities overflow at 65504.  The mode's default kernel keeps the exact running maximum in every key tile (lazy rescale:
    p <= 2^8); the optimistic variant (knob 1 = 60: reference searched in the first two tiles only) must fall back when boosted late keys
    push probabilities past fp16's range -> inf denominators -> the workgroup repeats exactly.  Both equal fp64 on fp16-rounded
    operands at every boost (boost 40: |q.k| ~ 40 x 8 x 0.125 is inside fp16's range for the operands, far outside for exp)."""
    n, BS = 256, 2
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(11)
    q, k, v = (torch.randn(BS, n, 256, generator=g).to(dev) for _ in range(3))
    k[:, 160:] *= boost
    nkv = torch.tensor([256, 231], dtype=torch.int32, device=dev)
    exact = eng_f16.debug_attention(q, k, v, nkv, False, 0.125).cpu().numpy()
    eng_f16.lib.gn_debug_set_variant(eng_f16.ctx, 1, 60)
    try:
        out = eng_f16.debug_attention(q, k, v, nkv, False, 0.125).cpu().numpy()
    finally:
        eng_f16.lib.gn_debug_set_variant(eng_f16.ctx, 1, 4)
    assert np.isfinite(out).all() and np.isfinite(exact).all()
    ref = _attn_ref(q, k, v, nkv, False, lambda t: t.half())
    for bs in range(BS):
        assert _rel(out[bs], ref[bs]) < 3e-3 and _rel(exact[bs], ref[bs]) < 3e-3, (bs, boost)
        assert _rel(out[bs], exact[bs]) < 2e-3, (bs, boost)


@pytest.mark.parametrize("name", ["lightglue_seed0_q96_r80", "lightglue_seed0_q200_r256"])
def test_fp16_attention_mode_on_golden_fixtures(eng_f16, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    dev = eng_f16.device
    f = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)[None]  # noqa: E731
    kq = np.column_stack([z["kp_q"], z["size_q"], z["angle_q"]]).astype(np.float32); kr = np.column_stack([z["kp_r"], z["size_r"], z["angle_r"]]).astype(np.float32)
    inp = dict(desc_q=f(z["desc_q"]), kpt_q=f(kq), n_q=torch.tensor([len(kq)], dtype=torch.int32, device=dev),
               desc_r=f(z["desc_r"]), kpt_r=f(kr), n_r=torch.tensor([len(kr)], dtype=torch.int32, device=dev), dem=f(z["dem"], torch.uint8), kpt_format=1)
    idx, score, nm = eng_f16.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    k = int(nm[0])
    assert k == len(z["idx"]) and np.array_equal(idx[0, :k].cpu().numpy(), z["idx"])
    assert np.abs(score[0, :k].cpu().numpy() - z["scores"][:, 0]).max() < 5e-3
    out = eng_f16.estimate(inp, z["K"])
    R, t = out["R"][0].cpu().numpy(), out["t"][0].cpu().numpy()
    assert int(out["ok"][0]) == 1 and np.linalg.norm(R - z["R"]) < 1e-8 and np.linalg.norm(t - z["t"]) / np.linalg.norm(z["t"]) < 1e-8


def test_fp16_attention_mode_guard_trips_on_out_of_range_q_k_v(state_dict_np):
    """q / k / v share the guarded fp16 domain.  The q rows of every Wqkv are scaled by 1e6 and the k rows by 1e-6: the scores -- and so the
    f32 oracle -- are unchanged and every hm16 activation stays where it was, but q no longer fits fp16.  bf16 attention (f32's exponent
    range) does not care; the fp16-attention mode raises the guard: 'flag' reports ZERO matches (never NaN), 'sync' re-runs the call with
    bf16 attention operands and returns the oracle's correspondences."""
    from gisnav_amd.engine import PoseEngine
    sd = dict(state_dict_np)
    for i in range(9):
        for leaf, shape in (("weight", (4, 64, 3, 256)), ("bias", (4, 64, 3))):
            key = f"transformers.{i}.self_attn.Wqkv.{leaf}"
            w = sd[key].copy().reshape(shape)
            w[:, :, 0] *= np.float32(1.0e6); w[:, :, 1] *= np.float32(1.0e-6)
            sd[key] = w.reshape(sd[key].shape)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    pairs = [make_pair(700 + i, n_q=256, n_r=250) for i in range(2)]
    ref = [oracle_match(tsd, p) for p in pairs]
    assert all(len(r[3]) > 50 for r in ref)

    def run(prec, guard):
        eng = PoseEngine(0, max_batch=2, max_kpts=256, precision=prec, state_dict=sd, guard=guard)
        inp = eng.stage_inputs(pairs)
        idx, score, nm = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        torch.cuda.synchronize()
        return idx.cpu().numpy(), nm.cpu().numpy(), eng.guard_status()

    def close_to_oracle(idx, nm):
        for b, (_, _, _, oidx) in enumerate(ref):
            a_ = {tuple(r) for r in idx[b, : nm[b]].tolist()}; b_ = {tuple(r) for r in oidx.numpy().tolist()}
            assert len(a_ ^ b_) <= 0.02 * len(b_), (b, len(a_ ^ b_), len(b_))

    idx, nm, st = run("f16x2_bf16_attn", "flag")
    assert st == (False, 0)
    close_to_oracle(idx, nm)
    idx, nm, st = run("f16x2_f16_attn", "flag")
    assert st[0] and (nm == 0).all()
    idx, nm, st = run("f16x2_f16_attn", "sync")
    assert st[1] == 1
    close_to_oracle(idx, nm)


# ------------------------------------------------------------------ guard word per sub-batch group, grow() replays state
def test_guard_words_are_per_substream_group(state_dict_np):
    from gisnav_amd.engine import PoseEngine
    eng = PoseEngine(0, max_batch=4, max_kpts=256, precision="f16x2_bf16_attn", state_dict=state_dict_np)
    inp = eng.stage_inputs([make_pair(80 + i, n_q=256 - 7 * i, n_r=256 - 3 * i) for i in range(4)])
    ref = {k: v.cpu().numpy().copy() for k, v in eng.estimate(inp, K_MATRIX).items()}
    assert (ref["n_match"] > 30).all() and not eng.guard_status()[0]
    eng.set_substreams(2)
    try:
        eng.lib.gn_debug_set_variant(eng.ctx, 25, 2)                  # group 1 (pairs 2, 3) starts with its guard word raised
        for _ in range(3):                                             # back to back: group 0 of call n + 1 must not clear group 1's word of call n
            out = eng.estimate(inp, K_MATRIX, out=eng.alloc_outputs(4))
        eng.flush(); torch.cuda.synchronize()
        o = {k: v.cpu().numpy() for k, v in out.items()}
        assert np.array_equal(o["n_match"][:2], ref["n_match"][:2]) and np.array_equal(o["R"][:2], ref["R"][:2]) and o["ok"][:2].all()
        assert (o["n_match"][2:] == 0).all() and (o["ok"][2:] == 0).all()
        assert eng.guard_status()[0]
        eng.lib.gn_debug_set_variant(eng.ctx, 25, 1)                  # now group 0 only
        out = eng.estimate(inp, K_MATRIX, out=eng.alloc_outputs(4))
        eng.flush(); torch.cuda.synchronize()
        o = {k: v.cpu().numpy() for k, v in out.items()}
        assert (o["n_match"][:2] == 0).all() and np.array_equal(o["n_match"][2:], ref["n_match"][2:]) and np.array_equal(o["t"][2:], ref["t"][2:])
    finally:
        eng.lib.gn_debug_set_variant(eng.ctx, 25, 0)
    out = eng.estimate(inp, K_MATRIX, out=eng.alloc_outputs(4))
    eng.flush(); torch.cuda.synchronize()
    assert all(np.array_equal(ref[k], out[k].cpu().numpy()) for k in ref) and not eng.guard_status()[0]


def test_grow_replays_sticky_context_state(state_dict_np):
    """LightGlueMatcher.__call__ with hw1 / hw2 on a cloud larger than the context: grow() first, image sizes after -- the call that
    triggers the grow is normalised by the GIVEN sizes; a SuperPoint sharing a grown engine keeps its weights and arithmetic."""
    from gisnav_amd.matcher import LightGlueMatcher
    from oracle import lightglue_sift as lg
    p = make_pair(33, n_q=300, n_r=280)
    tq = torch.from_numpy
    m = LightGlueMatcher("sift", params={"n_layers": 9, "filter_threshold": 0.5, "depth_confidence": -1, "width_confidence": -1},
                         state_dict=state_dict_np, max_kpts=128).to("cuda:0").eval()
    assert m._engine.kmax == 128
    laf = lambda kp, s, a: lg.laf_from_center_scale_ori(tq(kp)[None], tq(s)[None, :, None, None], tq(a)[None, :, None])  # noqa: E731
    dq, dr = lg.rootsift(tq(p.desc_q)), lg.rootsift(tq(p.desc_r))
    hw = (480, 640)
    dists, idx = m(dq.cuda(), dr.cuda(), laf(p.kp_q, p.size_q, p.angle_q).cuda(), laf(p.kp_r, p.size_r, p.angle_r).cuda(), hw1=hw, hw2=hw)
    assert m._engine.kmax >= 300
    dists2, idx2 = m(dq.cuda(), dr.cuda(), laf(p.kp_q, p.size_q, p.angle_q).cuda(), laf(p.kp_r, p.size_r, p.angle_r).cuda(), hw1=hw, hw2=hw)
    assert torch.equal(idx, idx2) and torch.equal(dists, dists2)     # the growing call already used hw (it used the keypoint extent before)
    from gisnav_amd.engine import PoseEngine
    from gisnav_amd.superpoint import SuperPoint
    from oracle import superpoint as osp
    eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f16x2_bf16_attn", feature="superpoint")
    sp = SuperPoint(engine=eng, max_keypoints=64, state_dict=osp.synthetic_state_dict(0), arithmetic="fp16")
    eng.set_substreams(2)
    img = np.random.default_rng(0).uniform(0, 1, (1, 64, 96)).astype(np.float32)
    a = [t.cpu().numpy().copy() if hasattr(t, "cpu") else t for t in sp.detect_and_describe_device(img)]
    eng.grow(512)
    b = [t.cpu().numpy() if hasattr(t, "cpu") else t for t in sp.detect_and_describe_device(img)]      # weights and arithmetic survived
    assert all(np.array_equal(x, y) for x, y in zip(a, b))        # (settings live in the context: gn_resize keeps them, nothing is replayed)


# ------------------------------------------------------------------ configs[4] at its size: SuperPoint at 1920x1080
def _sp_image(seed, h, w):
    sys.path.insert(0, GOLD)
    from make_superpoint_extractor_golden import superpoint_test_image
    return superpoint_test_image(seed, h, w)


def _explain_set_difference(kp_gpu, sc_gpu, kp_ref, sc_ref, k, tol):
    """Keypoints in exactly one of the two sets are admissible only if their decision margin is below the score tolerance: the score
    is within `tol` of the top-k cut-off (the k-th score) -- the only decision a 1e-5 score difference can flip once the NMS maps
    agree.  Returns (symmetric difference, unexplained)."""
    g = {(int(x), int(y)): float(s) for (x, y), s in zip(kp_gpu, sc_gpu)}
    r = {(int(x), int(y)): float(s) for (x, y), s in zip(kp_ref, sc_ref)}
    cut = min(min(r.values()), min(g.values())) if len(r) >= k else 0.005
    diff = set(g) ^ set(r)
    bad = [c for c in diff if abs((g.get(c) if c in g else r[c]) - cut) > tol]
    return len(diff), bad


@pytest.mark.parametrize("prec", ["f32", "f16x2_bf16_attn"])
def test_superpoint_extractor_1080p_against_oracle_and_transformers_fixture(prec):
    from gisnav_amd.engine import PoseEngine
    from gisnav_amd.superpoint import SuperPoint
    from oracle import superpoint as osp
    z = np.load(os.path.join(GOLD, "superpoint_extractor_seed7_1920x1080.npz"))
    h, w, k = int(z["h"]), int(z["w"]), int(z["k"])
    u8a, u8b = _sp_image(int(z["seed"]), h, w), _sp_image(8, h, w)
    assert hashlib.sha256(u8a.tobytes()).hexdigest() == str(z["image_sha256"]), "this platform regenerates different pixels: regenerate the fixture"
    sd = osp.synthetic_state_dict(0)
    eng = PoseEngine(0, max_batch=1, max_kpts=128, precision=prec, feature="superpoint")
    sp = SuperPoint(engine=eng, max_keypoints=k, state_dict=sd)
    batch = np.stack([u8a, u8b])                                                  # batch 2, as uint8 frames (scaled by 1/255 on the device)
    kpt, score, desc, n = sp.detect_and_describe_device(batch)
    torch.cuda.synchronize()
    kpt, score, desc = kpt.cpu().numpy(), score.cpu().numpy(), desc.cpu().numpy()
    rep = {}
    for b, u8 in enumerate((u8a, u8b)):
        img = torch.from_numpy(u8.astype(np.float32) * np.float32(1.0 / 255.0))
        okp, osc, od = osp.detect_and_describe(sd, img, k)
        m = int(n[b])
        assert m == len(okp) == k
        ndiff, bad = _explain_set_difference(kpt[b, :m, :2], score[b, :m], okp.numpy(), osc.numpy(), k, 2e-5)
        assert not bad, (prec, b, bad[:5])
        assert ndiff <= 8, (prec, b, ndiff)                                       # a handful of cut-off near-ties at most (measured: see the report)
        got = {(int(x), int(y)): i for i, (x, y) in enumerate(kpt[b, :m, :2])}
        ref = {(int(x), int(y)): i for i, (x, y) in enumerate(okp.numpy())}
        common = sorted(set(got) & set(ref))
        gi = np.array([got[c] for c in common]); ri = np.array([ref[c] for c in common])
        s_err = float(np.abs(score[b][gi] - osc.numpy()[ri]).max()); d_err = float(np.abs(desc[b][gi] - od.numpy()[ri]).max())
        assert s_err < 1e-5 and d_err < 1e-4, (s_err, d_err)
        assert (np.diff(score[b, :m]) <= 0).all()
        rep[f"image{b}"] = {"keypoints": m, "set_difference": ndiff, "max_score_err": s_err, "max_desc_err": d_err}
    # image 0 against the transformers-generated fixture (third-party code, not the repo's oracle)
    ndiff, bad = _explain_set_difference(kpt[0, :k, :2], score[0, :k], z["keypoints"], z["scores"], k, 2e-5)
    assert not bad and ndiff <= 8, (prec, ndiff, bad[:5])
    got = {(int(x), int(y)): i for i, (x, y) in enumerate(kpt[0, :k, :2])}
    every4 = [(i, got.get((int(x), int(y)))) for i, (x, y) in enumerate(z["keypoints"]) if i % 4 == 0]
    pairs = [(i // 4, j) for i, j in every4 if j is not None]
    assert len(pairs) >= 250
    d_err = float(np.abs(desc[0][[j for _, j in pairs]] - z["desc_every4"][[i for i, _ in pairs]]).max())
    assert d_err < 1e-4, d_err
    rep["vs_transformers_fixture"] = {"set_difference": ndiff, "max_desc_err_every4": d_err}
    _report("superpoint_1080p_" + prec, rep)


# ------------------------------------------------------------------ RCCL / N > 1 launch path
def _env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_rccl_selfcheck_world1_on_cuda0():
    """The collectives the N > 1 bench uses (broadcast of the flattened weights, barrier, all_reduce MAX / SUM, all_gather of result
    records) on a world-1 `nccl` (= RCCL) process group on cuda:0, in a fresh process."""
    r = subprocess.run([sys.executable, "-m", "gisnav_amd.dist", "--selfcheck"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["ok"] and line["world"] == 1 and line["broadcast_mb"] > 40


def test_bench_refuses_to_run_fewer_ranks_than_gpus():
    """`python bench.py --gpus 2` without torchrun spawns 2 nccl ranks itself; on a box with ONE GPU it must exit non-zero with a clear
    message (and print no JSON line) instead of silently measuring one rank."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a 1-GPU box (on a multi-GPU box the same command measures 2 ranks)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "visible" in r.stderr and "--gpus 2" in r.stderr, r.stderr[-2000:]
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


# ------------------------------------------------------------------ input side: raw wire records, pinned double-buffered staging
def _wire_batch(seed0, B, n):
    from gisnav_amd import wire
    pairs = [make_pair(seed0 + i, n_q=n - 3 * i, n_r=n - 5 * i) for i in range(B)]
    msgs = [(wire.pack_keypoints(p.kp_q, p.size_q, p.angle_q, p.desc_q), wire.pack_keypoints(p.kp_r, p.size_r, p.angle_r, p.desc_r), p.dem) for p in pairs]
    return pairs, msgs


@pytest.mark.parametrize("prec", ["f32", "f16x2_bf16_attn"])
def test_wire_records_on_the_device_equal_host_unpacked_inputs(prec, state_dict_np):
    """GN_KPT_RECORD: the 532-byte KEYPOINT_DTYPE records of OrthoStereoImage.query_sift uploaded as they are give bit-identical
    matches and poses to the host-unpacked (x, y, size, angle) + descriptor arrays (pose_node.py:207-213), also with sub-batch streams."""
    from gisnav_amd import _lib
    from gisnav_amd.engine import PoseEngine
    eng = PoseEngine(0, max_batch=4, max_kpts=256, precision=prec, state_dict=state_dict_np)
    pairs, msgs = _wire_batch(900, 4, 256)
    inp = eng.stage_inputs(pairs)
    ref = {k: v.cpu().numpy().copy() for k, v in eng.estimate(inp, K_MATRIX).items()}
    ridx, rsc, rn = (v.cpu().numpy().copy() for v in eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"]))
    assert ref["ok"].all() and (ref["n_match"] > 30).all()
    dev = eng.device
    rec = lambda side: torch.zeros((4, 256, 133), dtype=torch.float32, device=dev)  # noqa: E731
    rq, rr = rec(0), rec(1)
    for b, (q, r, _) in enumerate(msgs):
        rq[b, : len(q) // 532] = torch.from_numpy(np.frombuffer(q, dtype=np.float32).reshape(-1, 133).copy()).to(dev)
        rr[b, : len(r) // 532] = torch.from_numpy(np.frombuffer(r, dtype=np.float32).reshape(-1, 133).copy()).to(dev)
    rinp = dict(desc_q=None, desc_r=None, kpt_q=rq, kpt_r=rr, n_q=inp["n_q"], n_r=inp["n_r"], dem=inp["dem"], kpt_format=_lib.GN_KPT_RECORD)
    idx, sc, nm = eng.match(None, rq, inp["n_q"], None, rr, inp["n_r"], kpt_format=_lib.GN_KPT_RECORD)
    torch.cuda.synchronize()
    assert np.array_equal(nm.cpu().numpy(), rn)
    for b in range(4):
        assert np.array_equal(idx[b, : rn[b]].cpu().numpy(), ridx[b, : rn[b]]) and np.array_equal(sc[b, : rn[b]].cpu().numpy(), rsc[b, : rn[b]])
    for nsub in (1, 2):
        eng.set_substreams(nsub)
        out = eng.estimate(rinp, K_MATRIX)
        eng.flush(); torch.cuda.synchronize()
        assert all(np.array_equal(ref[k], out[k].cpu().numpy()) for k in ref), nsub
    eng.set_substreams(1)
    with pytest.raises(RuntimeError):        # 256-d feature contexts have no wire-record format
        e2 = PoseEngine(0, max_batch=1, max_kpts=128, precision="f32", feature="superpoint", state_dict=synthetic_state_dict(0, feature="superpoint"))
        e2.match(None, rq[:1, :128], inp["n_q"][:1], None, rr[:1, :128], inp["n_r"][:1], kpt_format=_lib.GN_KPT_RECORD)


def test_record_stager_streams_new_batches_with_identical_results(state_dict_np):
    """RecordStager: 7 DIFFERENT batches through 2 device slots (so slots are re-used while earlier work may still be running), staging of
    batch i + 1 issued before the estimate of batch i is enqueued: every batch's outputs equal the plain serial path's."""
    from concurrent.futures import ThreadPoolExecutor
    from gisnav_amd.engine import PoseEngine, RecordStager
    eng = PoseEngine(0, max_batch=3, max_kpts=256, precision="f16x2_bf16_attn", state_dict=state_dict_np)
    batches = [_wire_batch(1000 + 10 * s, 3, 256 - 8 * s) for s in range(7)]
    serial = []
    for pairs, _ in batches:
        o = eng.estimate(eng.stage_inputs(pairs), K_MATRIX)
        torch.cuda.synchronize()
        serial.append({k: v.cpu().numpy().copy() for k, v in o.items()})
    st = RecordStager(eng, max_batch=3, max_kpts=256, dem_hw=(480, 640), depth=2)
    outs = [eng.alloc_outputs(3) for _ in batches]
    with ThreadPoolExecutor(max_workers=1) as pool:
        cur = st.stage(batches[0][1])
        for i in range(len(batches)):
            fut = pool.submit(st.stage, batches[i + 1][1]) if i + 1 < len(batches) else None
            st.wait(cur)
            eng.estimate(cur, K_MATRIX, out=outs[i])
            st.release(cur)
            cur = fut.result() if fut else None
    eng.flush(); torch.cuda.synchronize()
    for s, o in zip(serial, outs):
        assert s["ok"].all()
        for k in s:
            assert np.array_equal(s[k], o[k].cpu().numpy()), k


def test_record_stager_refuses_malformed_batches_before_touching_a_slot(state_dict_np):
    """A refused batch (torn record, too many keypoints, wrong DEM size, too many pairs) raises GnError naming the pair and leaves the slot
    sequence untouched: the next good batch still gives the serial path's results."""
    from gisnav_amd import _lib
    from gisnav_amd.engine import PoseEngine, RecordStager
    eng = PoseEngine(0, max_batch=2, max_kpts=128, precision="f16x2_bf16_attn", state_dict=state_dict_np)
    pairs, msgs = _wire_batch(2000, 2, 120)
    want = {k: v.cpu().numpy().copy() for k, v in eng.estimate(eng.stage_inputs(pairs), K_MATRIX).items()}
    with RecordStager(eng, max_batch=2, max_kpts=128, dem_hw=(480, 640), depth=2, copy_threads=2) as st:
        q, r, dem = msgs[0]
        for bad, word in [([(q[:-7], r, dem)], "532-byte"), ([(q, r, dem[:100])], "DEM raster"), ([msgs[0]] * 3, "pairs staged"),
                          ([(q + q, r, dem)], "exceed"), ([], "pairs staged")]:
            with pytest.raises(_lib.GnError, match=word):
                st.stage(bad)
        assert st._next == 0
        cur = st.stage(msgs)
        st.wait(cur)
        got = eng.estimate(cur, K_MATRIX)
        st.release(cur)
        torch.cuda.synchronize()
        for k in want:
            assert np.array_equal(want[k], got[k].cpu().numpy()), k
    assert st._pool is None


def test_pose_node_shim_single_transfer_path_and_dem_cache(state_dict_np):
    """PoseNode.estimate stages one message with one host-to-device copy and reads its result with one device-to-host copy.  The DEM raster is
    uploaded with EVERY message by default (the reference never caches it, and upstream re-stamps dem_msg per message, stereo_node.py:272: a
    cache keyed on the DEM's stamp would never hit and would serve a stale raster to a caller that reuses a stamp -- ADVICE r3): a new raster
    under an old DEM stamp is used at once.  `cache_dem = True` keeps the device copy per REFERENCE stamp, the key the reference trusts for
    the tile's features (pose_node.py:225-241).  Results equal the engine's plain path bit for bit."""
    from gisnav_amd import wire
    from gisnav_amd.engine import PoseEngine
    from gisnav_amd.pose_node import PoseNode
    p = make_pair(86, n_q=300, n_r=280)
    node = PoseNode(state_dict_np, lambda ref: (p.kp_r, p.desc_r, p.size_r, p.angle_r), max_kpts=512, precision="f16x2_f16_attn")
    cam = wire.CameraInfo(k=K_MATRIX.reshape(-1), height=480, width=640)
    q = wire.pack_keypoints(p.kp_q, p.size_q, p.angle_q, p.desc_q)
    mk = lambda dem, sec, rsec=7: wire.OrthoStereoImage(query_sift=q, reference=wire.ImageMsg(p.ref, wire.Stamp(rsec, 0)), dem=wire.ImageMsg(dem, wire.Stamp(sec, 0)))  # noqa: E731
    assert node.cache_dem is False
    r1 = node.estimate(cam, mk(p.dem, 7))
    eng = PoseEngine(0, max_batch=1, max_kpts=512, precision="f16x2_f16_attn", state_dict=state_dict_np)
    plain = eng.estimate(eng.stage_inputs([p]), K_MATRIX)
    assert r1 is not None and np.array_equal(r1[0], plain["R"][0].cpu().numpy()) and np.array_equal(r1[1], plain["t"][0].cpu().numpy())
    assert node.last_num_matches == int(plain["n_match"][0])
    flat = np.zeros_like(p.dem)
    r_flat = node.estimate(cam, mk(flat, 7))                   # a NEW raster under the SAME DEM stamp: used, not a stale copy
    assert r_flat is not None and not np.array_equal(r_flat[1], r1[1])
    r_back = node.estimate(cam, mk(p.dem, 9))                  # and back, whatever the DEM stamp says
    assert np.array_equal(r_back[0], r1[0]) and np.array_equal(r_back[1], r1[1])
    node.cache_dem = True                                      # opt-in: device copy per reference stamp
    r_c1 = node.estimate(cam, mk(p.dem, 10))
    r_c2 = node.estimate(cam, mk(flat, 11))                    # same reference stamp: the cached raster (documented behaviour of the opt-in)
    assert np.array_equal(r_c1[1], r1[1]) and np.array_equal(r_c2[1], r1[1])
    r_c3 = node.estimate(cam, mk(flat, 11, rsec=8))            # new reference stamp: the new raster
    assert np.array_equal(r_c3[1], r_flat[1])
    node.cache_dem = False
    few = wire.pack_keypoints(p.kp_q[:9], p.size_q[:9], p.angle_q[:9], p.desc_q[:9])
    assert node.estimate(cam, wire.OrthoStereoImage(query_sift=few, reference=wire.ImageMsg(p.ref, wire.Stamp(7, 0)), dem=wire.ImageMsg(p.dem, wire.Stamp(8, 0)))) is None
    assert node.last_num_matches < node.MIN_MATCHES


def test_pinned_uploader_matches_plain_uploads_and_seams_accept_its_views(state_dict_np):
    """gisnav_amd.upload.PinnedUploader: same values and shapes as torch.tensor(a).to(device), dtype conversion on the way,
    buffers that grow, empty arrays; the views it returns drive seams B1 + B2 to the same matches and pose as plain uploads."""
    from gisnav_amd.matcher import LightGlueMatcher
    from gisnav_amd.pose import compute_pose
    from gisnav_amd.upload import PinnedUploader
    dev = torch.device("cuda", 0)
    up = PinnedUploader(dev)
    rng = np.random.default_rng(4)
    for shape in [(5, 128), (300, 128), (7, 2, 3), (0, 128), (301, 128)]:
        a = rng.standard_normal(shape)                                   # float64 on the host
        t = up("x", a)
        assert t.shape == torch.Size(shape) and t.dtype == torch.float32 and t.device == dev
        assert torch.equal(t.cpu(), torch.tensor(a, dtype=torch.float32))
    ids = up("ids", np.arange(17, dtype=np.int64), dtype=torch.int32)
    assert ids.dtype == torch.int32 and ids.cpu().tolist() == list(range(17))
    p = make_pair(87, n_q=256, n_r=240)
    m = LightGlueMatcher("sift", params={"filter_threshold": 0.5, "depth_confidence": -1, "width_confidence": -1}, state_dict=state_dict_np,
                         max_kpts=256, precision="f32").to(dev).eval()

    def laf(kp, size, ang):
        a = np.deg2rad(ang)
        L = np.zeros((len(kp), 2, 3), np.float32)
        L[:, 0, 0], L[:, 0, 1], L[:, 1, 0], L[:, 1, 1] = size * np.cos(a), size * np.sin(a), -size * np.sin(a), size * np.cos(a)
        L[:, :, 2] = kp
        return L

    d_plain, i_plain = m(torch.tensor(p.desc_q).to(dev), torch.tensor(p.desc_r).to(dev), torch.tensor(laf(p.kp_q, p.size_q, p.angle_q)).to(dev)[None],
                         torch.tensor(laf(p.kp_r, p.size_r, p.angle_r)).to(dev)[None])
    d_pin, i_pin = m(up("desc_q", p.desc_q), up("desc_r", p.desc_r), up("laf_q", laf(p.kp_q, p.size_q, p.angle_q))[None], up("laf_r", laf(p.kp_r, p.size_r, p.angle_r))[None])
    assert len(i_plain) > 30 and torch.equal(i_plain, i_pin) and torch.equal(d_plain, d_pin)

    class Cam:
        k = K_MATRIX.reshape(-1)
    idx = i_pin.cpu().numpy()
    pose = compute_pose(Cam, p.kp_q[idx[:, 0]], p.kp_r[idx[:, 1]], p.dem)
    assert pose is not None and pose[0].shape == (3, 3) and pose[1].shape == (3, 1) and abs(np.linalg.det(pose[0]) - 1) < 1e-9


# ------------------------------------------------------------------ solvePnPRansac's npoints == 4 branch (P3P) through the B2 seam
def test_compute_pose_with_exactly_four_points_takes_the_p3p_branch():
    """`compute_pose(camera_info, mkp_qry, mkp_ref, elevation)` with four matches (core/_shared.py:109-116 -> cv2's npoints == 4 branch): Gao's P3P
    on the GPU against the oracle's restatement on 150 noisy scenes with DEM relief -- always the same branch of the four-fold ambiguity (pose
    within 1e-3), and within 1e-6 on >= 95 % of the scenes: where the quartic has a near-double root both sides sit on a flat piece of the
    polynomial and their roots (Aberth iteration vs numpy's companion matrix, both Newton-polished) differ by ~sqrt(eps); 3 points -> None."""
    from gisnav_amd import pose as gpose
    from gisnav_amd.wire import CameraInfo
    from oracle import pnp_ransac as pr
    rs = np.random.default_rng(5)
    K = K_MATRIX
    cam = CameraInfo(k=K.reshape(-1))
    dem = (20 + 15 * np.sin(np.arange(480)[:, None] / 40.0) * np.cos(np.arange(640)[None, :] / 55.0)).astype(np.uint8)
    worst, done, loose = 0.0, 0, 0
    while done < 150:
        rv = rs.normal(0, 0.25, 3)
        R = pr.rodrigues_vec2mat(rv.reshape(3, 1))
        t = np.array([rs.uniform(-40, 40), rs.uniform(-40, 40), rs.uniform(220, 380)])
        ref = np.column_stack([rs.uniform(60, 580, 4), rs.uniform(60, 420, 4)]).astype(np.float32)
        x, y = np.floor(ref).astype(int).T
        obj = np.column_stack([ref, dem[y, x]]).astype(np.float64)
        pc = (R @ obj.T).T + t
        if (pc[:, 2] <= 1).any():
            continue
        qry = ((K @ (pc / pc[:, 2:]).T).T[:, :2] + rs.normal(0, 0.3, (4, 2))).astype(np.float32)
        want = pr.compute_pose(K.reshape(-1), qry, ref, dem)
        got = gpose.compute_pose(cam, qry, ref, dem)
        assert (want is None) == (got is None)
        if want is None:
            continue
        done += 1
        e = max(np.linalg.norm(got[0] - want[0]), np.linalg.norm(got[1] - want[1]) / np.linalg.norm(want[1]))
        worst = max(worst, float(e))
        loose += e >= 1e-6
        assert e < 1e-3, (done, e)
        assert abs(np.linalg.det(got[0]) - 1) < 1e-12
    _report("p3p_four_point_branch_150_scenes", {"max_pose_delta_vs_oracle": worst, "scenes_above_1e-6": int(loose)})
    assert loose <= 7, loose
    assert gpose.compute_pose(cam, qry[:3], ref[:3], dem) is None and pr.compute_pose(K.reshape(-1), qry[:3], ref[:3], dem) is None


# ------------------------------------------------------------------ contexts on concurrent host threads
def test_three_contexts_on_three_host_threads_keep_their_own_kernel_families(state_dict_np):
    """The kernel-family selectors behind the launch functions (exact-f32 / split GEMM family, attention variant, the LoFTR GEMM arithmetic) are
    per host thread and set from the calling context at every API entry: an f32 matcher context, an f16x2 matcher context and a split-fp16 LoFTR
    context driven from three threads at once (ctypes releases the GIL inside the calls) each reproduce their own serial results bit for bit."""
    import threading
    from gisnav_amd import loftr_synthetic as olf
    from gisnav_amd.engine import PoseEngine
    from gisnav_amd.loftr import LoFTR
    dev = torch.device("cuda", 0)
    pairs = [make_pair(70 + i, n_q=256, n_r=256) for i in range(2)]
    engs = {p: PoseEngine(0, max_batch=2, max_kpts=256, precision=p, state_dict=state_dict_np) for p in ("f32", "f16x2_f16_attn")}
    inps = {p: e.stage_inputs(pairs) for p, e in engs.items()}
    lf = LoFTR(state_dict=olf.synthetic_state_dict(0), fine=True, graph=False, arithmetic="split_fp16").to(dev).eval()
    i0, i1 = olf.synthetic_pair(3, 96, 128)
    data = {"image0": i0.to(dev), "image1": i1.to(dev)}

    def run_matcher(p):
        o = engs[p].estimate(inps[p], K_MATRIX)
        torch.cuda.current_stream().synchronize()
        return {k: v.cpu().numpy().copy() for k, v in o.items()}

    def run_loftr():
        o = lf(data)
        torch.cuda.current_stream().synchronize()
        return {k: v.cpu().numpy().copy() for k, v in o.items()}

    jobs = {"f32": lambda: run_matcher("f32"), "f16x2_f16_attn": lambda: run_matcher("f16x2_f16_attn"), "loftr": run_loftr}
    serial = {name: fn() for name, fn in jobs.items()}
    assert serial["f32"]["ok"].all() and serial["f16x2_f16_attn"]["ok"].all() and len(serial["loftr"]["confidence"]) > 20
    errors = []
    start = threading.Barrier(len(jobs))

    def worker(name, fn):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                start.wait()
                for it in range(25):
                    got = fn()
                    for k, v in serial[name].items():
                        if not np.array_equal(v, got[k]):
                            errors.append(f"{name} iteration {it}: {k} differs from the serial run")
                            return
        except Exception as e:   # noqa: BLE001
            errors.append(f"{name}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=worker, args=(n, f)) for n, f in jobs.items()]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
