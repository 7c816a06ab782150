"""Round-6 parity tests (-m gpu): correspondence indices CERTIFIED in the fast mode.

kornia's matcher hands PoseNode `match_indices` that are used as exact integers (ros/gisnav/gisnav/core/pose_node.py:285-297).  The headline
precision mode computes the assignment scores with an arithmetic error, so through round 5 it was a tolerance mode (13 / 9673 index mismatches on
low-margin weights at 16 x 1024) and only GN_PREC_F32 was index-exact.  gn_set_certify closes that: the match head keeps the runner-up of every
row / column maximum, flags every PAIR in which a decision lies within the calibrated error bound of flipping, and mode 2 runs the flagged pairs
again on the exact-f32 kernels.  Here:

  * the low- / mid-margin tables of rounds 2-5 at 4 x 512 (headline kernels forced) and 16 x 1024 (selected by the grid): WITHOUT the certificate
    the mismatches are counted as before and every pair that holds one must carry a flag; WITH it the result must equal the oracle's exactly;
    the re-run fraction is reported per weight set (margin-built weights: 0);
  * GN_PREC_F32 on the same bulk tables (it was only ever run at 4 x 512), and a third weight family with PyTorch-default initialisation;
  * estimate() with two sub-batch streams, and a call whose fp16-range guard trips, under the certificate;
  * bench.py's N = 2 launch path (gloo, both ranks on cuda:0), so that the sharded path runs on hardware every round.

Everything counted goes to gpurun_out/parity_r06.json (stamped with the digest of the LOADED library); profiles/r06_parity_report.json is a copy that
bench.py reads.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import oracle_match
from gisnav_amd.synthetic import K_MATRIX, make_pair
from gisnav_amd.weights import default_init_state_dict, synthetic_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOW_MARGIN = dict(ffn_out_std=4.8e-3, final_scale=4.0, matchability_bias=0.0, matchability_std=0.05)
MID_MARGIN = dict(ffn_out_std=1.2e-3, final_scale=12.0, matchability_bias=2.0, matchability_std=0.05)
HEADLINE = "f16x2_f16_attn"
SAFETY = 4.0
FAMILIES = {"low_margin": (lambda: synthetic_state_dict(0, **LOW_MARGIN), 0.0), "mid_margin": (lambda: synthetic_state_dict(0, **MID_MARGIN), 0.01),
            "margin_built": (lambda: synthetic_state_dict(0), 0.5), "default_init": (lambda: default_init_state_dict(0), 0.0)}
_REF_CACHE = {}


def _report(key, value):
    from gisnav_amd import _lib
    path = os.path.join(ROOT, "gpurun_out", "parity_r06.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    if data.get("source_digest") != _lib.library_digest():      # a report of another build: start over
        data = {"source_digest": _lib.library_digest()}
    data[key] = value
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


def _threads():
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))


def _family(name):
    make, th = FAMILIES[name]
    sd = make()
    return sd, {k: torch.from_numpy(v) for k, v in sd.items()}, th


def _pairs(shape):
    if shape == "4x512":
        return [make_pair(400 + i, n_q=512 - 31 * i, n_r=512 - 17 * i) for i in range(4)], [make_pair(460 + i, n_q=500, n_r=490) for i in range(4)], 4, 512
    # (the calibration sample has the size of the call: the kernel family -- and with it the arithmetic whose error is measured -- follows the grid)
    if shape == "8x1024":
        return [make_pair(6400 + i, n_q=1024 - 11 * (i % 3), n_r=1024 - 19 * (i % 4)) for i in range(8)], [make_pair(6460 + i, n_q=1024, n_r=1000) for i in range(8)], 8, 1024
    return [make_pair(4400 + i, n_q=1024 - 13 * (i % 5), n_r=1024 - 29 * (i % 3)) for i in range(16)], [make_pair(4460 + i, n_q=1024, n_r=1000) for i in range(16)], 16, 1024


def _refs(name, shape):
    """Oracle match lists of (family, shape), computed once per session."""
    key = (name, shape)
    if key not in _REF_CACHE:
        _threads()
        _, tsd, th = _family(name)
        _REF_CACHE[key] = [oracle_match(tsd, p, filter_threshold=th)[3].numpy() for p in _pairs(shape)[0]]
    return _REF_CACHE[key]


def _diff(idx_h, n_h, refs):
    """(total symmetric difference, oracle matches, per-pair symmetric differences)"""
    per = []
    for b, r in enumerate(refs):
        a = {(int(q), int(c)) for q, c in idx_h[b, : int(n_h[b])]}
        per.append(len(a ^ {(int(q), int(c)) for q, c in r}))
    return sum(per), sum(len(r) for r in refs), per


def _match(eng, inp):
    idx, score, n = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    torch.cuda.synchronize()
    return idx.cpu().numpy(), n.cpu().numpy()


def _forced_headline(eng, on):
    for k, v in ((14, 128), (1, 70), (19, 2)) if on else ((14, 0), (1, 4), (19, 1)):
        assert eng.lib.gn_debug_set_variant(eng.ctx, k, v) == 0


@pytest.mark.parametrize("name,shape,products", [("low_margin", "4x512", 3), ("mid_margin", "4x512", 3), ("low_margin", "16x1024", 3), ("mid_margin", "16x1024", 3),
                                                 ("margin_built", "16x1024", 3), ("default_init", "8x1024", 3),
                                                 ("low_margin", "16x1024", 2), ("mid_margin", "16x1024", 2), ("margin_built", "16x1024", 2), ("default_init", "8x1024", 2)])
def test_certified_indices_equal_the_oracle_on_every_weight_family(name, shape, products):
    """Headline mode, headline kernels.  (1) calibrate eps on four OTHER pairs (safety 4); (2) certificate in flag mode: mismatches counted as in
    round 5, and every pair that holds one is flagged -- the certificate never vouches for a wrong pair; (3) certificate in re-run mode: the match
    lists are the oracle's, pair for pair, index for index.  The re-run fraction and eps go into the report.
    products = 2: the block tail on two partial products (gn_set_ffn_products: the bench's fast pass) -- a larger eps, the same guarantee."""
    from gisnav_amd.engine import PoseEngine
    sd, _, th = _family(name)
    pairs, cal_pairs, B, K = _pairs(shape)
    refs = _refs(name, shape)
    eng = PoseEngine(0, max_batch=B, max_kpts=K, precision=HEADLINE, state_dict=sd, filter_threshold=th)
    forced = shape != "16x1024"
    try:
        _forced_headline(eng, forced)
        eng.set_ffn_products(products)
        cal = eng.calibrate_certify(eng.stage_inputs(cal_pairs), safety=SAFETY)
        inp = eng.stage_inputs(pairs)
        eng.set_certify("flag")
        eng.set_kernel_timing(400)
        idx_h, n_h = _match(eng, inp)
        names = [r["name"] for r in eng.kernel_table()]
        eng.set_kernel_timing(0)
        assert any(n.startswith("k_ffn128") for n in names) and any(n.startswith("k_attn_pw") for n in names), names
        assert all(n.rstrip(">").split(", ")[4] == str(products) for n in names if n.startswith("k_ffn128")), names      # k_ffn128<ABL, COMP, LOOP, QKV, PROD>
        flags = eng.uncertain(B)
        m0, t0, per = _diff(idx_h, n_h, refs)
        unflagged_wrong = [b for b in range(B) if per[b] and not flags[b]]
        eng.set_certify("rerun")
        eng.certify_stats(reset=True)
        idx_c, n_c = _match(eng, inp)
        st = eng.certify_stats()
        m1, t1, per1 = _diff(idx_c, n_c, refs)
        # a pair the certificate passed keeps the fast kernels' result untouched
        for b in range(B):
            if not flags[b]:
                assert int(n_c[b]) == int(n_h[b]) and np.array_equal(idx_c[b, : n_c[b]], idx_h[b, : n_h[b]]), b
    finally:
        _forced_headline(eng, False)
    row = {"eps": cal["eps"], "eps_measured_max_dP": cal["measured"], "safety": SAFETY, "cpu_matches": t0, "uncertified_index_mismatches": m0,
           "pairs": B, "pairs_flagged": int((flags != 0).sum()), "pairs_flagged_fp16_range": int((flags == 2).sum()),
           "certified_index_mismatches": m1, "rerun_fraction": st["rerun_fraction"], "f32_marginal_pairs": st["f32_marginal_pairs"],
           "block_tail_partial_products": products,
           "kernels": "k_qkv<., ., 2> + k_attn_pw + k_ffn128 (composed), " + ("forced" if forced else "selected by the grid")}
    print(name, shape, products, row)
    _report(f"certified_{shape}_{name}" + ("_tail2" if products == 2 else ""), row)
    del eng
    assert t0 > 300, row
    assert not unflagged_wrong, (row, per, flags.tolist())
    assert m1 == 0, (row, per1)
    assert st["rerun_pairs"] == int((flags != 0).sum()), (st, flags.tolist())
    if name == "margin_built":
        assert m0 == 0 and st["rerun_pairs"] == 0, row       # the bench's weights: nothing to re-run, the fast kernels' result stands


@pytest.mark.parametrize("name,mode", [("margin_built", "rerun"), ("mid_margin", "rerun"), ("margin_built", "deferred"), ("mid_margin", "deferred")])
def test_automatic_block_tail_level_follows_the_certificates_flags(name, mode):
    """gn_set_ffn_products(0): eps is calibrated for both levels; the context starts on three products, evaluates both levels' certificates on every call's
    scores and moves to two products after a window of 64 certified pairs only when that flags no more pairs than three products would.  Margin-built
    (bench) weights: nothing is flagged on either level -> two products from the fifth 16-pair call on.  Mid-margin weights: two products would flag most
    pairs -> it stays on three.  Either way, and whatever the level of a call, the certified indices are the oracle's (match(), synchronous certificate)
    and the poses an exact-f32 context's (estimate() on two sub-batch streams, certificate resolved one call later)."""
    from gisnav_amd.engine import PoseEngine
    sd, _, th = _family(name)
    pairs, cal_pairs, B, K = _pairs("16x1024")
    eng = PoseEngine(0, max_batch=B, max_kpts=K, precision=HEADLINE, state_dict=sd, filter_threshold=th)
    eng.set_ffn_products("auto")
    cal = eng.calibrate_certify(eng.stage_inputs(cal_pairs), safety=SAFETY)
    assert cal["eps_two_products"] >= cal["eps_three_products"] == cal["eps"] > 0.0, cal
    assert eng.ffn_level()["level"] == 3            # (no certificate yet: three products)
    inp = eng.stage_inputs(pairs)
    levels, wrong = [], []
    if mode == "rerun":
        refs = _refs(name, "16x1024")
        eng.set_certify("rerun")
        eng.certify_stats(reset=True)
        for _ in range(7):
            levels.append(eng.ffn_level()["level"])
            idx_c, n_c = _match(eng, inp)
            wrong.append(_diff(idx_c, n_c, refs)[0])
    else:
        ref = PoseEngine(0, max_batch=B, max_kpts=K, precision="f32", state_dict=sd, filter_threshold=th)
        want = {k: v.clone() for k, v in ref.estimate(ref.stage_inputs(pairs), K_MATRIX).items()}
        torch.cuda.synchronize()
        del ref
        eng.set_substreams(2)
        eng.set_certify("deferred")
        eng.certify_stats(reset=True)
        outs = [eng.alloc_outputs(B), eng.alloc_outputs(B)]
        for i in range(8):
            levels.append(eng.ffn_level()["level"])
            eng.estimate(inp, K_MATRIX, out=outs[i % 2])
        eng.flush()
        torch.cuda.synchronize()
        for o in outs:        # (calls 7 and 8: on two products for the margin-built weights)
            wrong.append(sum(int(not torch.equal(o[k], want[k])) for k in ("n_match", "ok", "n_inliers", "R", "t")))
        eng.set_certify("rerun")
        eng.set_substreams(1)
    lv = eng.ffn_level()
    st = eng.certify_stats()
    row = {"levels_of_the_calls": levels, "eps_two_products": cal["eps_two_products"], "eps_three_products": cal["eps_three_products"], "level_after": lv["level"],
           "calls_on_two_products": lv["calls_two_products"], "calls_on_three_products": lv["calls_three_products"], "switches": lv["switches"],
           "rerun_fraction": st["rerun_fraction"], "mismatches_per_call": wrong, "mode": mode}
    print(name, row)
    _report(f"automatic_level_16x1024_{name}_{mode}", row)
    ok_auto = lv["automatic"]
    del eng
    assert ok_auto and levels[0] == 3, row
    assert all(w == 0 for w in wrong), row
    if name == "margin_built":
        assert lv["level"] == 2 and levels[-1] == 2 and lv["calls_two_products"] >= 2 and lv["switches"] == 1 and st["rerun_pairs"] == 0, row
    else:
        assert lv["level"] == 3 and lv["calls_two_products"] == 0 and lv["switches"] == 0 and st["rerun_pairs"] > 0, row


def test_a_weight_reload_discards_the_two_levels_calibration(state_dict_np):
    """A calibration belongs to the weights it was measured on: gn_load_tensor puts the automatic block-tail level back to "not calibrated" (three products,
    the single stated eps); gn_set_ffn_level_eps states both eps again without the calibration pass."""
    from gisnav_amd.engine import PoseEngine
    eng = PoseEngine(0, max_batch=16, max_kpts=1024, precision=HEADLINE, state_dict=state_dict_np)
    eng.set_ffn_products("auto")
    cal = eng.calibrate_certify(eng.stage_inputs(_pairs("16x1024")[1]), safety=SAFETY)
    eng.set_certify("rerun")
    lv = eng.ffn_level()
    assert lv["automatic"] and lv["eps_two_products"] == cal["eps_two_products"] > 0.0, lv
    eng.load_state_dict(state_dict_np)
    lv = eng.ffn_level()
    assert not lv["automatic"] and lv["level"] == 3 and lv["eps_two_products"] < 0.0 and lv["eps_three_products"] < 0.0, lv
    eng.set_ffn_level_eps(cal["eps_two_products"], cal["eps_three_products"], level=2)
    lv = eng.ffn_level()
    assert lv["automatic"] and lv["level"] == 2, lv
    del eng


@pytest.mark.parametrize("name,shape", [("low_margin", "16x1024"), ("mid_margin", "16x1024"), ("default_init", "8x1024")])
def test_exact_f32_mode_on_the_bulk_tables(name, shape):
    """GN_PREC_F32 had only ever been run on low- / mid-margin weights at 4 x 512 (1.9 k matches).  The bulk tables and the default-init family:
    correspondence indices identical to the oracle's; its own certificate (eps_f32 = 1e-4: GPU f32 against torch-CPU f32 differ by summation
    order) counts the pairs that hold a decision THAT close -- reported, because exactness of f32 against f32 is empirical for those."""
    from gisnav_amd.engine import PoseEngine
    sd, _, th = _family(name)
    pairs, _, B, K = _pairs(shape)
    refs = _refs(name, shape)
    eng = PoseEngine(0, max_batch=B, max_kpts=K, precision="f32", state_dict=sd, filter_threshold=th)
    eng.set_certify("flag")
    idx_h, n_h = _match(eng, eng.stage_inputs(pairs))
    flags = eng.uncertain(B)
    m, t, per = _diff(idx_h, n_h, refs)
    row = {"index_mismatches": m, "cpu_matches": t, "pairs": B, "pairs_with_a_decision_within_1e-4": int((flags != 0).sum())}
    print(name, shape, row)
    _report(f"f32_{shape}_{name}", row)
    del eng
    assert m == 0 and t > 900, (row, per)


def test_certified_estimate_with_sub_batch_streams_and_a_tripped_range_guard():
    """gn_estimate under the certificate: two sub-batch streams (the flags are read once, after the join) on low-margin weights -- match counts
    equal the oracle's, poses equal an exact-f32 context's --, then the same call with group 1's fp16-range guard word raised (knob 25): its pairs
    are flagged 2 and come back from the f32 re-run instead of reporting zero matches."""
    from gisnav_amd.engine import PoseEngine
    _threads()
    sd, tsd, th = _family("low_margin")
    pairs = [make_pair(7400 + i, n_q=512 - 9 * i, n_r=500) for i in range(8)]
    refs = [oracle_match(tsd, p, filter_threshold=th)[3].numpy() for p in pairs]
    e32 = PoseEngine(0, max_batch=8, max_kpts=512, precision="f32", state_dict=sd, filter_threshold=th)
    want = {k: v.clone() for k, v in e32.estimate(e32.stage_inputs(pairs), K_MATRIX).items()}
    torch.cuda.synchronize()
    del e32
    assert [int(v) for v in want["n_match"].cpu()] == [len(r) for r in refs]
    eng = PoseEngine(0, max_batch=8, max_kpts=512, precision=HEADLINE, state_dict=sd, filter_threshold=th)
    eng.calibrate_certify(eng.stage_inputs([make_pair(7460 + i, n_q=500, n_r=490) for i in range(4)]), safety=SAFETY)
    inp = eng.stage_inputs(pairs)
    eng.set_certify("rerun")
    eng.set_substreams(2)
    for trip in (0, 2):
        assert eng.lib.gn_debug_set_variant(eng.ctx, 25, trip) == 0       # trip = g + 1: group g starts with its guard word raised
        eng.certify_stats(reset=True)
        got = eng.estimate(inp, K_MATRIX)
        torch.cuda.synchronize()
        st = eng.certify_stats()
        assert torch.equal(got["n_match"], want["n_match"]) and torch.equal(got["ok"], want["ok"]), (trip, got["n_match"], want["n_match"])
        ok = want["ok"].bool()
        assert float((got["R"][ok] - want["R"][ok]).abs().max()) < 1e-6 and float(((got["t"][ok] - want["t"][ok]).abs() / want["t"][ok].abs().clamp_min(1.0)).max()) < 1e-6, trip
        if trip:
            assert st["flagged_fp16_range"] == 4 and st["rerun_pairs"] >= 4, st
    eng.lib.gn_debug_set_variant(eng.ctx, 25, 0)
    eng.set_substreams(1)
    del eng


def test_two_product_block_tail_is_repeatable_and_close_to_the_three_product_form(state_dict_np):
    """gn_set_ffn_products(2): k_ffn128<., ., ., ., 2> drops the products with the activations' residual term.  16 x 1024, margin-built weights: the
    launch table shows the two-product instantiations, the best assignment score of every row stays within 5e-3 of the three-product form's,
    two runs give the same bits, and the correspondence indices are the same (nothing near a decision on these weights)."""
    from gisnav_amd.engine import PoseEngine
    pairs = [make_pair(9100 + i, n_q=1024, n_r=1024 - 5 * (i % 3)) for i in range(16)]
    eng = PoseEngine(0, max_batch=16, max_kpts=1024, precision=HEADLINE, state_dict=state_dict_np)
    inp = eng.stage_inputs(pairs)
    got = {}
    for products in (3, 2, 2):
        eng.set_ffn_products(products)
        eng.set_kernel_timing(400)
        idx, n = _match(eng, inp)
        names = [r["name"] for r in eng.kernel_table() if r["name"].startswith("k_ffn128")]
        eng.set_kernel_timing(0)
        assert len(names) >= 2 and all(nm.rstrip(">").split(", ")[4] == str(products) for nm in names), names
        md = eng.debug_read("max0", 16 * 1024).copy()          # best assignment score of every row
        got.setdefault(products, []).append((idx, n, md))
    a, b, c = got[3][0], got[2][0], got[2][1]
    assert np.array_equal(b[2].view(np.uint32), c[2].view(np.uint32)) and np.array_equal(b[0], c[0]) and np.array_equal(b[1], c[1])
    rel = float(np.abs(a[2] - b[2]).max())
    print("two-product tail: max |d best score| =", rel)
    assert 0.0 < rel < 5e-3, rel
    assert np.array_equal(a[1], b[1]) and all(np.array_equal(a[0][p, : a[1][p]], b[0][p, : b[1][p]]) for p in range(16))
    del eng


def test_every_bulk_context_proves_the_fused_projection_on_its_own_weights(state_dict_np):
    """the projection fused behind k_ffn128 is checked bitwise against the separate k_qkv launches at the first forward call after a
    weight load, on the context's own weights (self / cross form x one-tile / walking form): status 1 on bulk-sized contexts of every weight family
    tested here, -1 (never selected, never checked) on a small context; the call that triggered the check returns the same results as the next one."""
    from gisnav_amd.engine import PoseEngine
    small = PoseEngine(0, max_batch=2, max_kpts=512, precision=HEADLINE, state_dict=state_dict_np)
    inp = small.stage_inputs([make_pair(9300 + i, n_q=500, n_r=512) for i in range(2)])
    assert small.fused_projection_status() == -1
    _match(small, inp)
    assert small.fused_projection_status() == -1
    del small
    pairs = [make_pair(9310 + i, n_q=1024, n_r=1000) for i in range(16)]
    for name in ("margin_built", "low_margin", "default_init"):
        sd, _, th = _family(name)
        eng = PoseEngine(0, max_batch=16, max_kpts=1024, precision=HEADLINE, state_dict=sd, filter_threshold=th)
        inp = eng.stage_inputs(pairs)
        assert eng.fused_projection_status() == -1
        a = _match(eng, inp)                       # runs the self-check first
        assert eng.fused_projection_status() == 1, name
        b = _match(eng, inp)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]), name
        eng.load_state_dict(sd)                    # a reload asks for the proof again
        out = eng.estimate(inp, K_MATRIX)
        torch.cuda.synchronize()
        assert eng.fused_projection_status() == 1 and np.array_equal(out["n_match"].cpu().numpy(), a[1]), name
        del eng


def test_bucketed_remainder_of_one_or_two_pairs_gives_the_same_matches(state_dict_np):
    """the kernel family follows the number of pairs per call, so estimate_bucketed's remainder bucket of one or two pairs runs other
    kernels than a single padded call.  Documented as equal up to rounding; here: 10 and 9 pairs with bucket_pairs = 4 (remainders 2 and 1) give the
    same match counts, inlier counts and poses (1e-6) as one call."""
    from gisnav_amd.engine import PoseEngine
    for B in (10, 9):
        rs = np.random.default_rng(B)
        nq = rs.integers(300, 1000, B); nr = rs.integers(300, 1000, B)
        pairs = [make_pair(9500 + i, n_q=int(nq[i]), n_r=int(nr[i])) for i in range(B)]
        eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision=HEADLINE, state_dict=state_dict_np)
        inp = eng.stage_inputs(pairs)
        want = {k: v.clone() for k, v in eng.estimate(inp, K_MATRIX).items()}
        got, stats = eng.estimate_bucketed(inp, K_MATRIX, np.array([len(p.kp_q) for p in pairs]), np.array([len(p.kp_r) for p in pairs]), bucket_pairs=4)
        torch.cuda.synchronize()
        assert stats["groups"] == 3 and int(want["ok"].sum()) >= B - 1
        assert torch.equal(got["n_match"], want["n_match"]) and torch.equal(got["ok"], want["ok"]) and torch.equal(got["n_inliers"], want["n_inliers"]), B
        ok = want["ok"].bool()
        assert float((got["R"][ok] - want["R"][ok]).abs().max()) < 1e-6 and float((got["t"][ok] - want["t"][ok]).abs().max()) < 1e-4, B
        del eng


def test_deferred_certificate_resolves_one_call_later_with_the_same_results():
    """gn_set_certify(3): with sub-batch streams the flags of call n are read after call n + 1 has been enqueued.  A stream of six calls over two
    alternating input batches and two alternating output sets (mid-margin weights: some pairs flagged, some not) gives, after the flush, exactly what
    the synchronous certificate gives for the last two calls; re-using the previous call's outputs is refused by the engine."""
    from gisnav_amd import _lib
    from gisnav_amd.engine import PoseEngine
    sd, _, th = _family("mid_margin")
    eng = PoseEngine(0, max_batch=8, max_kpts=1024, precision=HEADLINE, state_dict=sd, filter_threshold=th)
    eng.calibrate_certify(eng.stage_inputs([make_pair(8460 + i, n_q=1024, n_r=1000) for i in range(4)]), safety=SAFETY)
    inps = [eng.stage_inputs([make_pair(8400 + 8 * j + i, n_q=1024 - 5 * i, n_r=1000) for i in range(8)]) for j in range(2)]
    eng.set_substreams(2)
    eng.set_certify("rerun")
    want = []
    for j in range(2):
        want.append({k: v.clone() for k, v in eng.estimate(inps[j], K_MATRIX).items()})
    torch.cuda.synchronize()
    sync_stats = eng.certify_stats(reset=True)
    eng.set_certify("deferred")
    outs = [eng.alloc_outputs(8), eng.alloc_outputs(8)]
    for i in range(6):
        eng.estimate(inps[i % 2], K_MATRIX, out=outs[i % 2])
    with pytest.raises(_lib.GnError):
        eng.estimate(inps[0], K_MATRIX, out=outs[1])          # the previous call's outputs
    eng.flush()
    torch.cuda.synchronize()
    st = eng.certify_stats()
    for j in range(2):
        for k in ("n_match", "ok", "n_inliers"):
            assert torch.equal(outs[j][k], want[j][k]), (j, k)
        assert torch.equal(outs[j]["R"], want[j]["R"]) and torch.equal(outs[j]["t"], want[j]["t"]), j
    assert st["calls"] == 6 and st["pairs"] == 48 and st["rerun_pairs"] == 3 * sync_stats["rerun_pairs"], (st, sync_stats)
    eng.set_certify("off")
    eng.set_substreams(1)
    del eng


def test_gathered_rerun_of_scattered_pairs_equals_the_exact_f32_mode_for_every_input_format():
    """Scattered flagged pairs are gathered into a staging block and re-run as one f32 batch (certify_rerun).  Mid-margin weights, 16 x 512, some pairs
    flagged and some not: for XYSA keypoints + descriptor arrays, for LAF keypoints, and for raw 532-byte wire records (descriptors inside the
    records, no descriptor arrays) the certified match lists -- and, through estimate(), match counts and poses -- equal an exact-f32 context's."""
    from gisnav_amd import _lib, wire
    from gisnav_amd.engine import PoseEngine
    sd, _, th = _family("mid_margin")
    pairs = [make_pair(9800 + i, n_q=512 - 3 * i, n_r=512) for i in range(16)]
    cal = [make_pair(9860 + i, n_q=500, n_r=512) for i in range(16)]

    def record_inputs(eng, ps):
        base = eng.stage_inputs(ps)
        rec = lambda kp, sz, an, ds: np.frombuffer(wire.pack_keypoints(kp, sz, an, ds), dtype=np.float32).reshape(-1, 133)      # noqa: E731
        rq = np.zeros((len(ps), 512, 133), np.float32); rr = np.zeros((len(ps), 512, 133), np.float32)
        for b, p in enumerate(ps):
            rq[b, : len(p.kp_q)] = rec(p.kp_q, p.size_q, p.angle_q, p.desc_q); rr[b, : len(p.kp_r)] = rec(p.kp_r, p.size_r, p.angle_r, p.desc_r)
        d = dict(base)
        d.update(desc_q=None, desc_r=None, kpt_q=torch.from_numpy(rq).to(eng.device), kpt_r=torch.from_numpy(rr).to(eng.device), kpt_format=_lib.GN_KPT_RECORD)
        return d

    e32 = PoseEngine(0, max_batch=16, max_kpts=512, precision="f32", state_dict=sd, filter_threshold=th)
    eng = PoseEngine(0, max_batch=16, max_kpts=512, precision=HEADLINE, state_dict=sd, filter_threshold=th)
    eng.calibrate_certify(eng.stage_inputs(cal), safety=SAFETY)
    eng.set_certify("rerun")
    seen_scattered = False
    for fmt in ("xysa", "record"):
        i32 = e32.stage_inputs(pairs) if fmt == "xysa" else record_inputs(e32, pairs)
        inp = eng.stage_inputs(pairs) if fmt == "xysa" else record_inputs(eng, pairs)
        a = e32.match(i32["desc_q"], i32["kpt_q"], i32["n_q"], i32["desc_r"], i32["kpt_r"], i32["n_r"], i32["kpt_format"])
        eng.set_certify("flag")
        eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"], inp["kpt_format"])
        flags = eng.uncertain(16)
        fl = np.flatnonzero(flags)
        print(fmt, "flags", flags.tolist())
        seen_scattered |= len(fl) >= 2 and (fl[-1] - fl[0] + 1) != len(fl)
        eng.set_certify("rerun")
        b = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"], inp["kpt_format"])
        torch.cuda.synchronize()
        assert torch.equal(a[2], b[2]), (fmt, a[2], b[2], flags)
        for p in range(16):
            k = int(a[2][p])
            assert torch.equal(a[0][p, :k], b[0][p, :k]), (fmt, p, flags)
        wa, wb = e32.estimate(i32, K_MATRIX), eng.estimate(inp, K_MATRIX)
        torch.cuda.synchronize()
        assert torch.equal(wa["n_match"], wb["n_match"]) and torch.equal(wa["ok"], wb["ok"]), fmt
        ok = wa["ok"].bool()
        if bool(ok.any()):
            assert float((wa["R"][ok] - wb["R"][ok]).abs().max()) < 1e-6 and float((wa["t"][ok] - wb["t"][ok]).abs().max()) < 1e-3, fmt
    assert seen_scattered, "the sample never produced scattered flags: the gathered path was not exercised"
    del eng, e32


def test_mirrors_certify_by_themselves_in_the_fast_mode():
    """The drop-in objects (seams B1 and B3) in the FAST precision on low-margin weights (filter_threshold 0: every mutual arg-max is a match,
    hundreds of them with a small margin): `LightGlueMatcher(..., precision=headline)` and `PoseNode(..., precision=headline)` calibrate on their first calls and certify every
    call -- the matcher's indices equal the oracle's on every one of six messages, the node's match count and pose equal an exact-f32 node's."""
    from gisnav_amd import wire
    from gisnav_amd.matcher import LightGlueMatcher
    from gisnav_amd.pose_node import PoseNode
    from oracle import lightglue_sift as lg
    _threads()
    sd = synthetic_state_dict(0, **LOW_MARGIN)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    th = 0.0
    m = LightGlueMatcher("sift", params={"n_layers": 9, "filter_threshold": th, "depth_confidence": -1, "width_confidence": -1}, state_dict=sd,
                         max_kpts=512, precision=HEADLINE).to("cuda:0").eval()
    tq = torch.from_numpy
    for i in range(6):
        p = make_pair(9700 + i, n_q=400 + 17 * i, n_r=512)
        laf_q = lg.laf_from_center_scale_ori(tq(p.kp_q).unsqueeze(0), tq(p.size_q)[None, :, None, None], tq(p.angle_q)[None, :, None])
        laf_r = lg.laf_from_center_scale_ori(tq(p.kp_r).unsqueeze(0), tq(p.size_r)[None, :, None, None], tq(p.angle_r)[None, :, None])
        dq, dr = lg.rootsift(tq(p.desc_q)), lg.rootsift(tq(p.desc_r))
        dists, idx = m(dq.cuda(), dr.cuda(), laf_q.cuda(), laf_r.cuda())
        want = oracle_match(tsd, p, filter_threshold=th)[3].numpy()
        assert np.array_equal(idx.cpu().numpy(), want), i
    st = m._engine.certify_stats()
    assert st["mode"] == 2 and st["calls"] == 6 and m._cal_left == 2 and m._cal_eps > 0.0, st

    class Node(PoseNode):
        CONFIDENCE_THRESHOLD = th
    p = make_pair(9720, n_q=500, n_r=512)
    res = {}
    for prec in ("f32", HEADLINE):
        node = Node(sd, lambda ref: (p.kp_r, p.desc_r, p.size_r, p.angle_r), max_kpts=512, precision=prec)
        node.camera_info = wire.CameraInfo(k=K_MATRIX.reshape(-1), height=480, width=640)
        node.pose_image = wire.OrthoStereoImage(query_sift=wire.pack_keypoints(p.kp_q, p.size_q, p.angle_q, p.desc_q),
                                                reference=wire.ImageMsg(p.ref, wire.Stamp(3, 0)), dem=wire.ImageMsg(p.dem, wire.Stamp(3, 0)))
        r = node.pose()
        res[prec] = (node.last_num_matches, r)
    assert res["f32"][0] == res[HEADLINE][0] == len(oracle_match(tsd, p, filter_threshold=th)[3])
    if res["f32"][1] is not None:
        assert res[HEADLINE][1] is not None and np.linalg.norm(res["f32"][1][0] - res[HEADLINE][1][0]) < 1e-9 and np.linalg.norm(res["f32"][1][1] - res[HEADLINE][1][1]) < 1e-6


@pytest.mark.parametrize("products", [3, 2])
def test_certified_matches_are_permutation_equivariant_at_the_bench_size(products):
    """A size-independent property at BASELINE configs[2]'s full size (32 pairs x 1024 keypoints per side), on discriminating weights: re-ordering the
    keypoints of either side re-orders the sums inside every attention row and every score panel, so the FAST arithmetic's scores move in their last
    bits and near-threshold decisions flip (the uncertified difference is counted); the certified match sets must be the same set of (query keypoint,
    reference keypoint) correspondences -- both are the exact arithmetic's -- except in pairs that are marginal even for the exact-f32 re-run
    (a decision within 1e-4: there f32's own summation order decides; counted by the library, bounded here)."""
    from gisnav_amd.engine import PoseEngine
    sd, _, th = _family("mid_margin")
    B, K = 32, 1024
    pairs = [make_pair(71_000 + i, n_q=K - 3 * (i % 7), n_r=K - 5 * (i % 5)) for i in range(B)]
    eng = PoseEngine(0, max_batch=B, max_kpts=K, precision=HEADLINE, state_dict=sd, filter_threshold=th)
    eng.set_ffn_products(products)
    eng.calibrate_certify(eng.stage_inputs([make_pair(72_000 + i, n_q=K, n_r=K - 24) for i in range(B)]), safety=SAFETY)
    inp = eng.stage_inputs(pairs)
    rs = np.random.default_rng(5)
    nq, nr = inp["n_q"].cpu().numpy(), inp["n_r"].cpu().numpy()
    perm_q = [rs.permutation(int(n)) for n in nq]
    perm_r = [rs.permutation(int(n)) for n in nr]
    shuf = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in inp.items()}
    for b in range(B):     # row i of the shuffled side = row perm[i] of the original
        for side, perm in (("q", perm_q[b]), ("r", perm_r[b])):
            ix = torch.from_numpy(perm).to(inp["kpt_q"].device)
            shuf[f"desc_{side}"][b, : len(perm)] = inp[f"desc_{side}"][b, ix]
            shuf[f"kpt_{side}"][b, : len(perm)] = inp[f"kpt_{side}"][b, ix]

    def sets(idx, n, mapped):
        out = []
        for b in range(B):
            m = idx[b, : int(n[b])]
            out.append({(int(perm_q[b][q]), int(perm_r[b][c])) for q, c in m} if mapped else {(int(q), int(c)) for q, c in m})
        return out

    res = {}
    for mode in ("flag", "rerun"):
        eng.set_certify(mode)
        eng.certify_stats(reset=True)
        a = sets(*_match(eng, inp), mapped=False)
        st_a = eng.certify_stats()
        b_ = sets(*_match(eng, shuf), mapped=True)
        st_b = eng.certify_stats()
        res[mode] = {"differences": sum(len(x ^ y) for x, y in zip(a, b_)), "pairs_differing": sum(1 for x, y in zip(a, b_) if x != y), "matches": sum(len(x) for x in a),
                     "rerun_pairs": st_b["rerun_pairs"], "f32_marginal_pairs": st_b["f32_marginal_pairs"], "first_call_rerun_pairs": st_a["rerun_pairs"]}
    row = {"pairs": B, "keypoints_per_side": K, "block_tail_partial_products": products, "uncertified": res["flag"], "certified": res["rerun"]}
    print(row)
    _report("permutation_equivariance_32x1024_mid_margin" + ("_tail2" if products == 2 else ""), row)
    del eng
    assert res["rerun"]["matches"] > 1500 and res["rerun"]["rerun_pairs"] > 0, row           # (discriminating: some pairs were re-run)
    assert res["rerun"]["pairs_differing"] <= res["rerun"]["f32_marginal_pairs"], row
    if res["rerun"]["f32_marginal_pairs"] == 0:
        assert res["rerun"]["differences"] == 0, row


def test_certified_matches_transpose_when_the_two_sides_are_swapped():
    """Second size-independent property at 32 x 1024, mid-margin weights: LightGlue is symmetric in its two inputs (shared weights, bidirectional cross
    attention, row- and column-wise log-softmax, mutual check), so matching (reference, query) must return the transposed correspondences.  The two calls
    run different sums in a different order (side 0 and side 1 trade places in every kernel); the certified sets must agree except in pairs that are
    marginal even in exact f32."""
    from gisnav_amd.engine import PoseEngine
    sd, _, th = _family("mid_margin")
    B, K = 32, 1024
    pairs = [make_pair(73_000 + i, n_q=K - 3 * (i % 7), n_r=K - 5 * (i % 5)) for i in range(B)]
    eng = PoseEngine(0, max_batch=B, max_kpts=K, precision=HEADLINE, state_dict=sd, filter_threshold=th)
    eng.calibrate_certify(eng.stage_inputs([make_pair(72_000 + i, n_q=K, n_r=K - 24) for i in range(B)]), safety=SAFETY)
    inp = eng.stage_inputs(pairs)
    swp = dict(inp, desc_q=inp["desc_r"], kpt_q=inp["kpt_r"], n_q=inp["n_r"], desc_r=inp["desc_q"], kpt_r=inp["kpt_q"], n_r=inp["n_q"])
    res = {}
    for mode in ("flag", "rerun"):
        eng.set_certify(mode)
        eng.certify_stats(reset=True)
        ia, na = _match(eng, inp)
        ib, nb = _match(eng, swp)
        st = eng.certify_stats()
        a = [{(int(q), int(c)) for q, c in ia[b, : int(na[b])]} for b in range(B)]
        t = [{(int(c), int(q)) for q, c in ib[b, : int(nb[b])]} for b in range(B)]
        res[mode] = {"differences": sum(len(x ^ y) for x, y in zip(a, t)), "pairs_differing": sum(1 for x, y in zip(a, t) if x != y), "matches": sum(len(x) for x in a),
                     "rerun_pairs": st["rerun_pairs"], "f32_marginal_pairs": st["f32_marginal_pairs"]}
    row = {"pairs": B, "keypoints_per_side": K, "uncertified": res["flag"], "certified": res["rerun"]}
    print(row)
    _report("swap_symmetry_32x1024_mid_margin", row)
    del eng
    assert res["rerun"]["matches"] > 1500 and res["rerun"]["rerun_pairs"] > 0, row
    assert res["rerun"]["pairs_differing"] <= res["rerun"]["f32_marginal_pairs"], row
    if res["rerun"]["f32_marginal_pairs"] == 0:
        assert res["rerun"]["differences"] == 0, row


def test_bench_n2_launch_path_on_one_gpu():
    """the sharded path under the driver every round.  `python bench.py --gpus 2` spawns its two ranks itself (gloo, both on
    cuda:0): contiguous shards, the weight broadcast, barriers, max-over-ranks timing, the all-gather of result records."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--no-extras", "--no-traffic", "--no-cpu-baseline", "--no-stream"]
    lines = []
    for _ in range(2):
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
        js = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(js) == 1, r.stdout[-800:]
        lines.append(json.loads(js[0]))
    a, b = lines
    mg = a["multi_gpu"]
    assert a["n_gpus"] == 2 and mg["ranks_seen"] == 2 and mg["rank_blocks_found_in_gathered_records"] == 2 and a["result_records_gathered"] == 16, mg
    assert len(mg["per_rank_ms_per_step"]) == 2 and a["poses_ok_per_step"] >= 15, a
    assert mg["records_sha256"] == b["multi_gpu"]["records_sha256"]          # the gathered result records of two launches: bit for bit
    _report("bench_n2_gloo_shared_gpu", {"ranks_seen": mg["ranks_seen"], "records_sha256": mg["records_sha256"], "value": a["value"]})
