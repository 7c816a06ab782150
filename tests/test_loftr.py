"""LoFTR (BASELINE.json north_star / configs[1]; judge-added row N2): oracle known-answer tests (CPU) and HIP-vs-oracle parity (-m gpu).

PARITY UNPINNED: kornia is not importable here and the reference tree holds no LoFTR vector, so `oracle/loftr.py` -- a restatement of
the published architecture along kornia's module layout -- is pinned only by the analytic tests below.  Stated tolerances of the GPU
tests: backbone / transformer features max-rel 2e-4 (f32 against f32 with different summation orders through 20 convolutions and 8
encoder layers), coarse match INDICES identical, confidences 2e-4, fine keypoints 2e-3 px.
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import loftr as lf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sd():
    return lf.synthetic_state_dict(0)


# ------------------------------------------------------------------ oracle KATs (CPU)
def test_linear_attention_equals_the_explicit_formula():
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, 7, 3, 4, generator=g, dtype=torch.float64) for _ in range(3))
    out = lf.linear_attention(q.float(), k.float(), v.float()).double()
    phi = lambda t: torch.where(t > 0, t + 1, torch.exp(t))  # noqa: E731  elu(x) + 1
    ref = torch.zeros_like(q)
    for n in range(2):
        for h in range(3):
            Q, K, V = phi(q[n, :, h]), phi(k[n, :, h]), v[n, :, h]
            A = Q @ K.T                                       # un-normalised "attention" of the kernel trick
            ref[n, :, h] = (A @ V) / (A.sum(1, keepdim=True) + 1e-6)
    assert (out - ref).abs().max() < 1e-5


def test_position_encoding_legacy_divisor_and_layout():
    pe = lf.position_encoding_sine(256, 6, 9)
    assert pe.shape == (256, 6, 9)
    for i in (0, 1, 5, 63):
        div = math.exp(-2.0 * i)                              # `-log(1e4) / d_model // 2` == -1.0: the published (buggy) formula
        assert abs(float(pe[4 * i, 2, 3]) - math.sin(4 * div)) < 1e-6 and abs(float(pe[4 * i + 1, 2, 3]) - math.cos(4 * div)) < 1e-6
        assert abs(float(pe[4 * i + 2, 2, 3]) - math.sin(3 * div)) < 1e-6 and abs(float(pe[4 * i + 3, 2, 3]) - math.cos(3 * div)) < 1e-6
    fixed = lf.position_encoding_sine(256, 6, 9, temp_bug_fix=True)
    assert abs(float(fixed[4, 0, 0]) - math.sin(math.exp(2 * -math.log(10000.0) / 128))) < 1e-6


def test_coarse_matching_recovers_a_permutation_and_applies_border_and_threshold():
    hc, wc = 9, 12
    L = hc * wc
    g = torch.Generator().manual_seed(1)
    f0 = torch.randn(1, L, 256, generator=g) * 3.0
    perm = torch.randperm(L, generator=g)
    f1 = torch.zeros_like(f0)
    f1[0, perm] = f0[0]                                       # cell i of image 0 == cell perm[i] of image 1
    b, i, j, conf, k0, k1 = lf.coarse_matching(f0, f1, (hc, wc), (hc, wc))
    inside = lambda c: (c // wc >= 2) & (c // wc < hc - 2) & (c % wc >= 2) & (c % wc < wc - 2)  # noqa: E731
    expect = [(a, int(perm[a])) for a in range(L) if inside(torch.tensor(a)) and inside(perm[a])]
    assert [(int(a), int(c)) for a, c in zip(i, j)] == expect and len(expect) > 3
    assert (conf > 0.99).all() and torch.equal(k0[:, 0], (i % wc).float() * 8) and torch.equal(k1[:, 1], (j // wc).float() * 8)
    f1w = f1 * 0.05                                           # flat similarities: nothing clears thr = 0.2
    assert len(lf.coarse_matching(f0 * 0.05, f1w, (hc, wc), (hc, wc))[1]) == 0


def test_cross_layer_updates_feat1_from_the_updated_feat0(sd):
    g = torch.Generator().manual_seed(2)
    a, b = torch.randn(1, 12, 256, generator=g), torch.randn(1, 12, 256, generator=g)
    f0, f1 = lf.local_feature_transformer(sd, "loftr_coarse", ["self", "cross"], a, b)
    p0, p1 = "loftr_coarse.layers.0", "loftr_coarse.layers.1"
    a1, b1 = lf.encoder_layer(sd, p0, a, a), lf.encoder_layer(sd, p0, b, b)
    a2 = lf.encoder_layer(sd, p1, a1, b1)
    b2 = lf.encoder_layer(sd, p1, b1, a2)                     # a2, not a1
    assert torch.equal(f0, a2) and torch.equal(f1, b2)
    assert not torch.allclose(b2, lf.encoder_layer(sd, p1, b1, a1))


def test_fine_matching_expectation_of_a_peaked_heat_map():
    M, C = 3, 128
    f0 = torch.zeros(M, 25, C); f1 = torch.zeros(M, 25, C)
    f0[:, 12, 0] = 100.0
    for m, r in enumerate((0, 12, 24)):
        f1[m, r, 0] = 100.0                                   # window position r correlates: (x, y) = (r % 5, r // 5)
    k0, k1 = lf.fine_matching(f0, f1, torch.zeros(M, 2), torch.full((M, 2), 40.0))
    assert torch.allclose(k1, torch.tensor([[36.0, 36.0], [40.0, 40.0], [44.0, 44.0]]), atol=1e-4) and torch.equal(k0, torch.zeros(M, 2))


def test_whole_model_matches_the_known_shift_of_a_synthetic_pair(sd):
    h, w = 96, 128
    i0, i1 = lf.synthetic_pair(3, h, w)
    taps = {}
    out = lf.loftr_forward(sd, i0, i1, taps=taps)
    assert taps["x3_out"].shape == (2, 256, 12, 16) and taps["x1_out"].shape == (2, 128, 48, 64)
    i, j = out["i_ids"], out["j_ids"]
    assert len(i) > 20
    assert torch.equal(j, (i // 16 - 1) * 16 + (i % 16 - 2))   # image1 = image0 shifted by (16, 8) px = (2, 1) coarse cells
    assert (out["keypoints1"] - out["keypoints1_c"]).abs().max() <= 4.0 and torch.equal(out["keypoints0"], out["keypoints0_c"])
    assert (torch.diff(i) > 0).all()


# ------------------------------------------------------------------ HIP against the oracle
def _nhwc(t, cpad):
    """oracle (N, C, H, W) -> (N, H, W, cpad) with zero padding channels."""
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cpad)
    out[..., :c] = t.permute(0, 2, 3, 1)
    return out.numpy()


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize("arithmetic", ["exact_f32", "split_fp16"])
@pytest.mark.parametrize("h,w,seed", [(128, 160, 1), (136, 200, 2), (480, 640, 1)])
def test_loftr_hip_against_oracle(h, w, seed, arithmetic, sd):
    """Both arithmetics of the convolutions / linear layers: the exact f32 matrix instruction (configs[1]'s "fp32") and the f32-ACCURATE
    split-fp16 scheme of the matcher (two fp16 terms per operand, three products, f32 accumulate) meet the same bars."""
    from gisnav_amd.loftr import LoFTR
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    i0, i1 = lf.synthetic_pair(seed, h, w)
    taps = {}
    ref = lf.loftr_forward(sd, i0, i1, taps=taps)
    m = LoFTR(state_dict=sd, arithmetic=arithmetic).to("cuda:0").eval()
    out = m({"image0": i0[None, None].cuda(), "image1": i1[None, None].cuda()}, with_ids=True)
    hc, wc, L = h // 8, w // 8, (h // 8) * (w // 8)
    Lp = (L + 127) // 128 * 128
    # backbone
    assert _rel(m.debug_read("x1", 2 * (h // 2) * (w // 2) * 128).reshape(2, h // 2, w // 2, 128), _nhwc(taps["x1"], 128)) < 2e-5
    assert _rel(m.debug_read("x2", 2 * (h // 4) * (w // 4) * 224).reshape(2, h // 4, w // 4, 224), _nhwc(taps["x2_out"], 224)) < 2e-4    # (the buffer holds the FPN's x2_out by now)
    assert _rel(m.debug_read("x3_out", 2 * L * 256).reshape(2, hc, wc, 256), _nhwc(taps["x3_out"], 256)) < 1e-4
    assert _rel(m.debug_read("x1_out", 2 * (h // 2) * (w // 2) * 128).reshape(2, h // 2, w // 2, 128), _nhwc(taps["x1_out"], 128)) < 2e-4
    # coarse transformer output
    tok = m.debug_read("tok", 2 * Lp * 256).reshape(2, Lp, 256)[:, :L]
    f0, f1 = taps["loftr_coarse.7"]
    assert _rel(tok[0], f0[0].numpy()) < 2e-4 and _rel(tok[1], f1[0].numpy()) < 2e-4
    # matches
    assert len(ref["i_ids"]) > (50 if h < 400 else 3000)
    assert torch.equal(out["i_ids"].cpu(), ref["i_ids"]) and torch.equal(out["j_ids"].cpu(), ref["j_ids"])       # coarse correspondence indices: identical
    assert (out["confidence"].cpu() - ref["confidence"]).abs().max() < 2e-4
    assert torch.equal(out["keypoints0"].cpu(), ref["keypoints0"])
    assert (out["keypoints1"].cpu() - ref["keypoints1"]).abs().max() < 2e-3
    assert out["batch_indexes"].shape == (len(ref["i_ids"]),)


@pytest.mark.gpu
def test_loftr_returns_every_match_of_a_large_pair(sd):
    """kornia returns ALL mutual matches; so does the mirror's default (max_matches=None -> one slot per coarse cell of image0).  736 x 1280 has
    14 720 coarse cells -- more than the 8 192 the mirror once defaulted to and the fine level then runs on > 10 000 windows: indices identical to the
    oracle's, every match there, split-fp16 arithmetic (the faster one; the exact one is swept by tools/fuzz_loftr_sizes.py up to 1080 x 1920)."""
    from gisnav_amd.loftr import LoFTR
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    i0, i1 = lf.synthetic_pair(4, 736, 1280)
    ref = lf.loftr_forward(sd, i0, i1)
    out = LoFTR(state_dict=sd, arithmetic="split_fp16").to("cuda:0").eval()({"image0": i0[None, None].cuda(), "image1": i1[None, None].cuda()}, with_ids=True)
    assert len(ref["i_ids"]) > 8192
    assert torch.equal(out["i_ids"].cpu(), ref["i_ids"]) and torch.equal(out["j_ids"].cpu(), ref["j_ids"])
    assert (out["confidence"].cpu() - ref["confidence"]).abs().max() < 2e-4
    assert (out["keypoints1"].cpu() - ref["keypoints1"]).abs().max() < 2e-3


@pytest.mark.gpu
def test_loftr_coarse_only_context_and_drop_in_dictionary(sd):
    from gisnav_amd.loftr import LoFTR
    i0, i1 = lf.synthetic_pair(5, 128, 160)
    ref = lf.loftr_forward(sd, i0, i1, fine=False)
    sd_c = {k: v for k, v in sd.items() if not k.startswith(("loftr_fine", "fine_preprocess", "backbone.layer2_outconv", "backbone.layer1_outconv"))}
    m = LoFTR(state_dict=sd_c, fine=False).to("cuda:0").eval()
    out = m({"image0": i0.cuda(), "image1": i1.cuda()})
    assert set(out) == {"keypoints0", "keypoints1", "confidence", "batch_indexes"} and out["keypoints0"].device.type == "cuda"
    assert torch.equal(out["keypoints0"].cpu(), ref["keypoints0"]) and torch.equal(out["keypoints1"].cpu(), ref["keypoints1"])
    out2 = m({"image0": i0.cuda(), "image1": i1.cuda()})                     # bitwise repeatable (second call: graph replay)
    assert all(torch.equal(out[k], out2[k]) for k in out)
    m_plain = LoFTR(state_dict=sd_c, fine=False, graph=False).to("cuda:0").eval()   # plain stream launches instead of the captured hipGraph
    out3 = m_plain({"image0": i0.cuda(), "image1": i1.cuda()})
    assert all(torch.equal(out[k], out3[k]) for k in out)
    j0, j1 = lf.synthetic_pair(6, 128, 160)                                  # the replayed graph on NEW inputs
    ref2 = lf.loftr_forward(sd, j0, j1, fine=False)
    out4 = m({"image0": j0.cuda(), "image1": j1.cuda()})
    assert torch.equal(out4["keypoints0"].cpu(), ref2["keypoints0"]) and torch.equal(out4["keypoints1"].cpu(), ref2["keypoints1"])
    with pytest.raises(RuntimeError):
        LoFTR(state_dict={"backbone.conv1.weight": sd["backbone.conv1.weight"]}).to("cuda:0")({"image0": i0.cuda(), "image1": i1.cuda()})


def test_product_side_synthetic_generator_equals_the_oracles(sd):
    """bench.py times LoFTR with gisnav_amd.loftr_synthetic (the product never imports oracle/): same tensors, same image pairs."""
    from gisnav_amd import loftr_synthetic as ls
    mine = ls.synthetic_state_dict(0)
    assert set(mine) == set(sd) and all(torch.equal(mine[k], sd[k]) for k in sd)
    a, b = lf.synthetic_pair(4, 64, 96)
    c, d = ls.synthetic_pair(4, 64, 96)
    assert torch.equal(a, c) and torch.equal(b, d)


@pytest.mark.gpu
def test_loftr_in_front_of_the_pose_solver(sd):
    """LoFTR matches -> DEM lift -> solvePnPRansac on the device (`loftr_pose`) against the oracle chain (oracle LoFTR -> oracle compute_pose):
    a tile rendered through a known camera would need a renderer; here the shifted synthetic pair is enough to pin the plumbing -- the same
    matches go into the same solver, so R, t agree with the oracle's to 1e-6 (the fine keypoints differ by < 2e-3 px)."""
    from gisnav_amd.engine import PoseEngine
    from gisnav_amd.loftr import LoFTR, loftr_pose
    from gisnav_amd.synthetic import K_MATRIX
    from oracle import pnp_ransac as pr
    h, w = 240, 320
    i0, i1 = lf.synthetic_pair(7, h, w)
    dem = (10 + 8 * np.sin(np.arange(h)[:, None] / 30.0) * np.cos(np.arange(w)[None, :] / 45.0)).astype(np.uint8)
    ref = lf.loftr_forward(sd, i0, i1)
    want = pr.compute_pose(K_MATRIX.reshape(-1), ref["keypoints0"].numpy(), ref["keypoints1"].numpy(), dem)
    m = LoFTR(state_dict=sd).to("cuda:0").eval()
    eng = PoseEngine(0, max_batch=1, max_kpts=2048, precision="f32")
    got = loftr_pose(m, eng, i0.cuda(), i1.cuda(), dem, K_MATRIX)
    assert (want is None) == (got is None)
    if want is not None:
        assert got[2] == len(ref["i_ids"])
        assert np.linalg.norm(got[0] - want[0]) < 1e-4 and np.linalg.norm(got[1] - want[1]) / np.linalg.norm(want[1]) < 1e-4


@pytest.mark.gpu
def test_loftr_no_matches_and_truncation(sd):
    """A featureless pair has no confident mutual maximum -> zero matches, empty (0, 2) tensors like kornia; `max_matches` below the number of
    matches keeps the FIRST ones in ascending cell order of image0 (kornia returns all: the cap is this build's, documented in the header)."""
    from gisnav_amd.loftr import LoFTR
    flat = torch.full((128, 160), 0.5)
    m = LoFTR(state_dict=sd).to("cuda:0").eval()
    out = m({"image0": flat.cuda(), "image1": flat.cuda()})
    ref = lf.loftr_forward(sd, flat, flat)
    assert len(ref["i_ids"]) == out["keypoints0"].shape[0]
    assert out["keypoints1"].shape == (out["keypoints0"].shape[0], 2) and out["confidence"].shape == (out["keypoints0"].shape[0],)
    i0, i1 = lf.synthetic_pair(1, 128, 160)
    full = lf.loftr_forward(sd, i0, i1)
    assert len(full["i_ids"]) > 40
    capped = LoFTR(state_dict=sd, max_matches=40).to("cuda:0").eval()({"image0": i0.cuda(), "image1": i1.cuda()}, with_ids=True)
    assert capped["keypoints0"].shape == (40, 2) and torch.equal(capped["i_ids"].cpu(), full["i_ids"][:40]) and torch.equal(capped["j_ids"].cpu(), full["j_ids"][:40])
    assert (capped["keypoints1"].cpu() - full["keypoints1"][:40]).abs().max() < 2e-3


@pytest.mark.gpu
def test_loftr_split_arithmetic_falls_back_when_activations_leave_fp16_range(sd):
    """An image scaled by 3e5 drives the backbone's activations past fp16's 65504: the split-fp16 forward raises its guard word and the call
    is repeated on the exact kernels -- the outputs are then BIT-identical to an exact-f32 context's (and never NaN)."""
    from gisnav_amd.loftr import LoFTR
    i0, i1 = lf.synthetic_pair(1, 128, 160)
    big = {"image0": (i0 * 3.0e5).cuda(), "image1": (i1 * 3.0e5).cuda()}
    ex = LoFTR(state_dict=sd, arithmetic="exact_f32").to("cuda:0").eval()(big, with_ids=True)
    sp = LoFTR(state_dict=sd, arithmetic="split_fp16").to("cuda:0").eval()
    got = sp(big, with_ids=True)
    assert all(torch.equal(ex[k], got[k]) for k in ex) and bool(torch.isfinite(got["keypoints1"]).all())
    small = sp({"image0": i0.cuda(), "image1": i1.cuda()}, with_ids=True)       # the same context afterwards, in range: split arithmetic again
    ref = lf.loftr_forward(sd, i0, i1)
    assert torch.equal(small["i_ids"].cpu(), ref["i_ids"]) and torch.equal(small["j_ids"].cpu(), ref["j_ids"])


@pytest.mark.gpu
def test_exact_f32_gemm_on_64_row_tiles_equals_the_128_row_kernel_bitwise():
    """k_gemm_f32_m64 (round 5: the exact-f32 GEMM on 64 x 128 tiles for grids that would leave most CUs without a workgroup -- LoFTR's coarse
    cross halves) sums the same products in the same order as k_gemm_f32_v3: identical bits, with and without bias, at the coarse level's shapes
    and at a shape whose 128-row grid is above the threshold (the 128-row kernel runs either way there)."""
    from gisnav_amd.engine import PoseEngine
    eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f32")
    g = torch.Generator(device="cpu").manual_seed(3)
    try:
        for (M, N, K) in ((4864, 256, 256), (4864, 512, 512), (9728, 256, 256), (256, 128, 64), (128 * 48, 1024, 96)):
            A = torch.randn(M, K, generator=g).cuda(); Wt = torch.randn(N, K, generator=g).cuda(); b = torch.randn(N, generator=g).cuda()
            outs = {}
            for thr in (0, 320):
                eng.lib.gn_debug_set_variant(eng.ctx, 41, thr)
                outs[thr] = (eng.debug_gemm(A, Wt, b).cpu().numpy(), eng.debug_gemm(A, Wt, None).cpu().numpy())
            assert np.array_equal(outs[0][0], outs[320][0]) and np.array_equal(outs[0][1], outs[320][1]), (M, N, K)
            ref = (A.double() @ Wt.double().T + b.double()).cpu().numpy()
            assert np.abs(outs[320][0] - ref).max() < 2e-6 * np.abs(ref).max() * np.sqrt(K)
    finally:
        eng.lib.gn_debug_set_variant(eng.ctx, 41, 320)


@pytest.mark.gpu
@pytest.mark.parametrize("arithmetic", ["exact_f32", "split_fp16"])
def test_loftr_round5_forms_equal_the_forms_they_replace_bitwise(arithmetic, sd):
    """Round 5 changed HOW LoFTR's forward is computed, not what: the next channel slice's halo tile prefetched into registers, the rows-per-wave
    choice re-fitted, the stem's weights transposed in LDS, the zero-padding channel steps of the 196-channel layers skipped, and the fine level
    run only on the windows of the matches there are, sides one behind the other (a cross half computes the side it updates).  Developer knob 42
    = 15 selects every old form: all outputs are identical -- at a size whose grids are ragged, with the match list capped below the number of
    matches (windows behind the cap are skipped), and on a featureless pair (no match at all: every fine-level tile leaves at once)."""
    from gisnav_amd.engine import PoseEngine
    from gisnav_amd.loftr import LoFTR
    eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f32")      # a gn_ctx to reach the process-wide developer knob
    i0, i1 = lf.synthetic_pair(2, 136, 200)
    flat = torch.full((136, 200), 0.5)
    cases = [({"image0": i0.cuda(), "image1": i1.cuda()}, {}), ({"image0": i0.cuda(), "image1": i1.cuda()}, {"max_matches": 37}),
             ({"image0": flat.cuda(), "image1": flat.cuda()}, {})]
    try:
        res = {}
        for knob in (15, 0):
            eng.lib.gn_debug_set_variant(eng.ctx, 42, knob)
            res[knob] = []
            for data, kw in cases:
                m = LoFTR(state_dict=sd, arithmetic=arithmetic, **kw).to("cuda:0").eval()      # (a new context per knob: the graph is captured with it in force)
                res[knob].append([m(data, with_ids=True) for _ in range(2)][-1])
                del m
    finally:
        eng.lib.gn_debug_set_variant(eng.ctx, 42, 0)
    assert res[0][0]["keypoints0"].shape[0] > 100 and res[0][1]["keypoints0"].shape[0] == 37 and res[0][2]["keypoints0"].shape[0] == 0
    for old, new in zip(res[15], res[0]):
        assert old.keys() == new.keys()
        for k in old:
            assert old[k].shape == new[k].shape and torch.equal(old[k], new[k]), k
