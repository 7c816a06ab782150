"""-m gpu tests of the round-2 fused kernels against the paths they replace (both run through the C ABI on the same inputs):

  * k_head_fused (two sweeps that recompute the similarity tiles, gn_match_head.hip) vs the similarity GEMM + five passes: ragged
    and tiny keypoint counts (tile / row-block boundaries, the _no_match rule n < 2), one pair (column splits S > 1) and many
    pairs (S = 1), both operand formats (hm16 of the f16x2 mode, f32);
  * k_qkv (gn_qkv.hip) vs the tiled GEMM with the rotary / scale / bf16 / V^T epilogues: the matcher's outputs with knob 19 on
    and off (k_qkv is forced at the small batch so the 128-token shape runs), and the bf16 q | k rows and V^T panels themselves.
"""
import numpy as np
import pytest
import torch

from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu


def _match(eng, inp):
    idx, score, n_match = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    torch.cuda.synchronize()
    return idx.clone(), score.clone(), n_match.clone()


@pytest.mark.parametrize("prec", ["f16x2_bf16_attn", "f32"])
@pytest.mark.parametrize("shape", [[(256, 256)], [(63, 65), (129, 127), (1, 200), (200, 1), (5, 7), (256, 191), (2, 2), (97, 256)]])
def test_fused_head_equals_unfused_head_on_ragged_and_tiny_pairs(prec, shape):
    from gisnav_amd.engine import PoseEngine
    sd = synthetic_state_dict(0)
    pairs = [make_pair(900 + i, n_q=nq, n_r=nr) for i, (nq, nr) in enumerate(shape)]
    eng = PoseEngine(0, max_batch=len(pairs), max_kpts=256, precision=prec, state_dict=sd)
    inp = eng.stage_inputs(pairs)
    fused = _match(eng, inp)
    stats_f = {k: eng.debug_read(k, len(pairs) * 256) for k in ("rowmax", "rowlog", "colmax", "collog")}
    eng.lib.gn_debug_set_variant(eng.ctx, 16, 0)      # similarity GEMM + k_row_stats / k_col_stats / k_row_argmax / k_col_argmax / k_compact
    ref = _match(eng, inp)
    eng.lib.gn_debug_set_variant(eng.ctx, 16, 1)
    assert torch.equal(fused[2], ref[2]), (fused[2], ref[2])
    for b, (nq, nr) in enumerate(shape):
        k = int(ref[2][b])
        assert k == 0 if (nq < 2 or nr < 2) else k >= 0
        assert torch.equal(fused[0][b, :k], ref[0][b, :k]), b
        if k:
            assert (fused[1][b, :k] - ref[1][b, :k]).abs().max().item() < 1e-5
        if nq >= 2 and nr >= 2:
            for name, n in (("rowmax", nq), ("rowlog", nq), ("colmax", nr), ("collog", nr)):
                u = eng.debug_read(name, len(pairs) * 256).reshape(len(pairs), 256)[b, :n]
                assert np.abs(u - stats_f[name].reshape(len(pairs), 256)[b, :n]).max() < 1e-4, (name, b)
    again = _match(eng, inp)                           # the arrival counters are back at zero: a second fused call gives the same
    assert torch.equal(again[2], fused[2]) and all(torch.equal(again[0][b, :int(fused[2][b])], fused[0][b, :int(fused[2][b])]) for b in range(len(pairs)))


def test_qkv_kernel_equals_projection_gemm():
    from gisnav_amd.engine import PoseEngine
    sd = synthetic_state_dict(0)
    pairs = [make_pair(950 + i, n_q=256 - 9 * i, n_r=256 - 17 * i) for i in range(4)]
    eng = PoseEngine(0, max_batch=4, max_kpts=256, precision="f16x2_bf16_attn", state_dict=sd)
    inp = eng.stage_inputs(pairs)
    T = 4 * 2 * 256
    out, panels = {}, {}
    eng.lib.gn_debug_set_variant(eng.ctx, 27, 3)       # all three partial products, like the GEMM (the default drops x_m . w_h: checked below)
    for knob in (0, 2):                                # 0: k_gemm_p2 with the bf16 epilogues, 2: k_qkv forced at this (small) batch size
        eng.lib.gn_debug_set_variant(eng.ctx, 19, knob)
        for stop in (3 + knob // 2, 6 + knob // 2):    # after the first self projection / the first cross projection (k_qkv adds k_rot_table to the launch count)
            eng.lib.gn_debug_set_variant(eng.ctx, 4, stop)
            _match(eng, inp)
            panels[(knob, stop - knob // 2)] = (eng.debug_read("qkb", T * 256).copy(), eng.debug_read("vtb", T * 128).copy())
        eng.lib.gn_debug_set_variant(eng.ctx, 4, 0)
        out[knob] = _match(eng, inp)
    eng.lib.gn_debug_set_variant(eng.ctx, 19, 1)
    assert torch.equal(out[0][2], out[2][2])
    for b in range(4):
        k = int(out[0][2][b])
        assert torch.equal(out[0][0][b, :k], out[2][0][b, :k])
        assert (out[0][1][b, :k] - out[2][1][b, :k]).abs().max().item() < 1e-5
    for stop in (3, 6):
        for a, c in zip(panels[(0, stop)], panels[(2, stop)]):
            a16, c16 = a.view(np.uint16), c.view(np.uint16)
            same = np.mean(a16 == c16)
            assert same > 0.999, (stop, same)          # bf16 outputs of the same f32 arithmetic: identical up to the rounding of sums taken in another order
            fa = (a16.astype(np.uint32) << 16).view(np.float32); fc = (c16.astype(np.uint32) << 16).view(np.float32)
            assert np.abs(fa - fc).max() <= 2.0 ** -7 * max(1.0, float(np.abs(fa).max()))
    # the shipped default: TWO partial products (x_h w_h + x_h w_m).  The dropped term is 2^-12 of every x, below the 16-bit rounding of q / k / v: most
    # outputs keep their bits, the rest move by one unit in the last place, and the correspondences do not change
    eng.lib.gn_debug_set_variant(eng.ctx, 27, 2)
    eng.lib.gn_debug_set_variant(eng.ctx, 19, 2)
    for stop in (4, 7):
        eng.lib.gn_debug_set_variant(eng.ctx, 4, stop)
        _match(eng, inp)
        for a, c in zip(panels[(2, stop - 1)], (eng.debug_read("qkb", T * 256), eng.debug_read("vtb", T * 128))):
            a16, c16 = a.view(np.uint16), c.view(np.uint16)
            assert np.mean(a16 == c16) > 0.8, (stop, np.mean(a16 == c16))
            fa = (a16.astype(np.uint32) << 16).view(np.float32); fc = (c16.astype(np.uint32) << 16).view(np.float32)
            assert np.abs(fa - fc).max() <= 2.0 ** -7 * max(1.0, float(np.abs(fa).max()))
    eng.lib.gn_debug_set_variant(eng.ctx, 4, 0)
    two = _match(eng, inp)
    eng.lib.gn_debug_set_variant(eng.ctx, 19, 1)
    assert torch.equal(two[2], out[2][2]) and all(torch.equal(two[0][b, : int(two[2][b])], out[2][0][b, : int(two[2][b])]) for b in range(4))
