"""Developer tool: the fused block tail (k_ffn_fused) against the three-launch form after the FIRST ffn of the schedule."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402


def hm16_decode(raw_u32, rows, cols):
    h = raw_u32.view(np.float16).reshape(rows, cols // 16, 2, 16).astype(np.float32)
    return (h[:, :, 0, :] + h[:, :, 1, :]).reshape(rows, cols)


def main():
    sd = synthetic_state_dict(0)
    eng = PoseEngine(0, max_batch=1, max_kpts=256, precision="f16x2_bf16_attn", state_dict=sd)
    inp = eng.stage_inputs([make_pair(7, n_q=256, n_r=256)])
    args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    T = 512
    res = {}
    for stop in (4, 5):
        for mode in (0, 3):
            eng.lib.gn_debug_set_variant(eng.ctx, 10, mode)
            eng.lib.gn_debug_set_variant(eng.ctx, 4, stop)
            eng.match(*args)
            torch.cuda.synchronize()
            res[(stop, mode)] = hm16_decode(eng.debug_read("x_p", T * 256, np.uint32), T, 256), hm16_decode(eng.debug_read("msg_p", T * 256, np.uint32), T, 256)
    x_in, msg = res[(4, 0)]
    print("inputs identical:", np.array_equal(res[(4, 0)][0], res[(4, 3)][0]), np.array_equal(msg, res[(4, 3)][1]), "|x|", np.abs(x_in).max(), "|msg|", np.abs(msg).max())
    a, b = res[(5, 0)][0], res[(5, 3)][0]
    print("ffn out: ref max", np.abs(a).max(), "delta(ref - x_in) max", np.abs(a - x_in).max(), "fused - ref max", np.abs(a - b).max())
    d = np.abs(a - b)
    print("err by 64-row block:", d.reshape(8, 64, 256).max(axis=(1, 2)))
    print("err by 32-col block:", d.reshape(512, 8, 32).max(axis=(0, 2)))
    print("err by row within block (first block):", d[:64].max(axis=1).round(6))
    print("err by col (first 64):", d[:, :64].max(axis=0).round(6))
    # reference FFN in numpy f64 from the decoded inputs
    W1, b1 = sd["transformers.0.self_attn.ffn.0.weight"].astype(np.float64), sd["transformers.0.self_attn.ffn.0.bias"].astype(np.float64)
    g, be = sd["transformers.0.self_attn.ffn.1.weight"].astype(np.float64), sd["transformers.0.self_attn.ffn.1.bias"].astype(np.float64)
    W2, b2 = sd["transformers.0.self_attn.ffn.3.weight"].astype(np.float64), sd["transformers.0.self_attn.ffn.3.bias"].astype(np.float64)
    xin = np.concatenate([x_in, msg], 1).astype(np.float64)
    h = xin @ W1.T + b1
    h = (h - h.mean(1, keepdims=True)) / np.sqrt(h.var(1, keepdims=True) + 1e-5) * g + be
    from math import erf
    h = 0.5 * h * (1 + np.vectorize(erf)(h / np.sqrt(2)))
    y = x_in + h @ W2.T + b2
    print("vs f64 reference: three-launch", np.abs(a - y).max(), "fused", np.abs(b - y).max(), " (ffn delta magnitude", np.abs(y - x_in).max(), ")")


if __name__ == "__main__":
    main()
