"""Developer tool: k_ffn128 (128 tokens per workgroup, knob 14 = 128) against k_ffn_fused (knob 14 = 64) in one process:
final features / correspondences of a bench-sized call, per-kernel HIP-event times, and (optionally) the phase stamps of k_ffn128.
usage: ffn128_ab.py [batch] [stamps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 1024
eng = PoseEngine(0, max_batch=B, max_kpts=N, precision=os.environ.get("GN_AB_PREC", "f16x2_f16_attn"), state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i) for i in range(B)])
args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
T = B * 2 * N
res = {}
for shape, comp in ((64, 0), (128, 0), (64, 1), (128, 1), (128, 0), (128, 1)):
    eng.lib.gn_debug_set_variant(eng.ctx, 14, shape)
    eng.lib.gn_debug_set_variant(eng.ctx, 28, comp)
    for _ in range(2):
        idx, score, n = (v.cpu().numpy().copy() for v in eng.match(*args))
    x = eng.debug_read("x", T * 256).copy()
    eng.set_kernel_timing(400)
    for _ in range(4):
        eng.match(*args)
    torch.cuda.synchronize()
    rows = eng.kernel_table()
    eng.set_kernel_timing(0)
    step = sum(r["ms"] for r in rows) / 4
    big = sorted(rows, key=lambda r: -r["ms"])[:4]
    print(f"shape {shape} composed {comp}: " + ", ".join(f"{r['name'][:22]} {1000 * r['ms'] / r['launches']:.2f} us" for r in big) + f"; all kernels {step:.3f} ms per call", flush=True)
    res.setdefault((shape, comp), (idx, score, n, x))
(i0, s0, n0, x0), (i1, s1, n1, x1) = res[(64, 0)], res[(128, 1)]
for key in ((128, 0), (64, 1)):
    xk = res[key][3]
    print(f"  {key}: max rel diff of the final features vs (64, 0): {np.abs(xk - x0).max() / np.abs(x0).max():.3e}; indices identical: {all(np.array_equal(i0[b, : n0[b]], res[key][0][b, : res[key][2][b]]) for b in range(B))}")
rel = np.abs(x1 - x0).max() / np.abs(x0).max()
same = all(np.array_equal(i0[b, : n0[b]], i1[b, : n1[b]]) for b in range(B)) and np.array_equal(n0, n1)
print(f"final features: max rel diff {rel:.3e}; finite {np.isfinite(x1).all()}; matches per pair {n0[:4]} / {n1[:4]}; indices identical: {same}")
names = ["prologue", "gemm0", "publish msg", "gemm1 msg", "gemm1 x", "ln stats", "gelu q0", "gemm2+gelu", "yt store", "epilogue"]
order = [0, 1, 2, 3, 10, 4, 5, 6, 7, 8, 9]      # stamp 10 sits between the two halves of GEMM 1
for abl in ([int(v) for v in sys.argv[2:]] if len(sys.argv) > 2 else []):
    eng.lib.gn_debug_set_variant(eng.ctx, 14, 128)
    eng.lib.gn_debug_set_variant(eng.ctx, 28, 1)
    eng.lib.gn_debug_set_variant(eng.ctx, 12, abl)
    eng.lib.gn_debug_set_variant(eng.ctx, 4, 5)       # stop after the first FFN launch: the stamps in `sim` are not overwritten by the head
    eng.match(*args)
    torch.cuda.synchronize()
    nb = T // 128
    ts = eng.debug_read("sim", nb * 4 * 12 * 2, np.uint32).view(np.int64).reshape(nb, 4, 12)[:, :, order]
    d = np.diff(ts, axis=2).astype(np.float64)
    print(f"k_ffn128<{abl}> phase cycles, median over blocks (wave 0) / median of max over waves:")
    for k, nm in enumerate(names):
        print(f"  {nm:12s} {np.median(d[:, 0, k]):9.0f}   {np.median(d[:, :, k].max(axis=1)):9.0f}")
    tot = ts[:, 0, -1] - ts[:, 0, 0]
    raw = eng.debug_read("sim", nb * 4 * 12 * 2, np.uint32).view(np.int64).reshape(nb, 4, 12)
    print("  stamp 11 - stamp 5 (first pass of the exposed GELU quarters):", np.median(raw[:, 0, 11] - raw[:, 0, 5]))
    if abl & 128:
        t2 = eng.debug_read("sim", (nb * 4 * 12 + nb * 4 * 16) * 2, np.uint32).view(np.int64)[nb * 4 * 12:].reshape(nb, 4, 16)
        full = np.concatenate([raw[:, :, 3:4], t2], axis=2)
        print("  GEMM 1 per k-tile 4.. (wave 0, median):", " ".join(f"{v:.0f}" for v in np.median(np.diff(t2[:, 0, 4:], axis=1), axis=0)))
        q4 = np.concatenate([raw[:, :, 6:7], t2[:, :, :4]], axis=2)
        print("  GEMM 2 per quarter (wave 0, median):", " ".join(f"{v:.0f}" for v in np.median(np.diff(q4[:, 0, :], axis=1), axis=0)))
    print("  block total median", np.median(tot), " kernel span (max end - min start)", ts[:, :, -1].max() - ts[:, :, 0].min(), flush=True)
