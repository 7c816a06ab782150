"""Developer tool: k_attn_pw (knob 1 = 70: one wave per SIMD, 64 queries per wave, pinned order) against k_attn16_v5 (knob 1 = 4) in one
process: (a) the attention op alone on random q / k / v with ragged key counts -- outputs compared with each other and with fp64 on the
fp16-rounded operands; (b) a bench-sized matcher call -- correspondences, final features, per-kernel HIP-event times.
usage: attn_pw_ab.py [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
VARS = [int(v) for v in os.environ.get("GN_AB_VARS", "4,70").split(",")]
N = 1024
eng = PoseEngine(0, max_batch=B, max_kpts=N, precision=os.environ.get("GN_AB_PREC", "f16x2_f16_attn"), state_dict=synthetic_state_dict(0))
dev = torch.device("cuda", 0)


def ref64(q, k, v, nkv, cross):
    q, k, v = (t.half().double() for t in (q, k, v))
    out = torch.zeros_like(q)
    for bs in range(q.shape[0]):
        kv = bs ^ 1 if cross else bs
        n = int(nkv[kv])
        for h in range(4):
            sl = slice(64 * h, 64 * h + 64)
            s = (q[bs, :, sl] * 0.125) @ k[kv, :n, sl].T
            out[bs, :, sl] = torch.softmax(s, dim=-1) @ v[kv, :n, sl]
    return out


g = torch.Generator(device="cpu").manual_seed(5)
for npad, BS, boost in ((256, 4, 1.0), (512, 6, 1.0), (2048, 4, 1.0), (2048, 2, 6.0), (1024, 2, 40.0)):
    q, k, v = (torch.randn(BS, npad, 256, generator=g).to(dev) for _ in range(3))
    k[:, npad // 2:] *= boost
    nkv = torch.tensor([npad, npad - 37, 5, npad // 2 + 2, 64, 129][:BS], dtype=torch.int32, device=dev)
    for cross in (False, True):
        outs = {}
        for var in (4, 70):
            eng.lib.gn_debug_set_variant(eng.ctx, 1, var)
            outs[var] = eng.debug_attention(q, k, v, nkv, cross, 0.125).double()
            again = eng.debug_attention(q, k, v, nkv, cross, 0.125).double()
            if not torch.equal(outs[var], again):
                bad = (outs[var] != again).nonzero()
                print(f"  variant {var}: NOT repeatable: {len(bad)} elements differ, first {bad[:3].tolist()}, max diff {float((outs[var] - again).abs().max()):.3e}")
        eng.lib.gn_debug_set_variant(eng.ctx, 1, 4)
        r = ref64(q, k, v, nkv, cross)
        e = {var: float((outs[var] - r).abs().max() / r.abs().max()) for var in outs}
        print(f"npad {npad} BS {BS} boost {boost} cross {int(cross)}: rel err vs fp64  v5 {e[4]:.2e}  pw {e[70]:.2e};  pw vs v5 {float((outs[70] - outs[4]).abs().max() / r.abs().max()):.2e}; finite {bool(torch.isfinite(outs[70]).all())}", flush=True)

inp = eng.stage_inputs([make_pair(i) for i in range(B)])
args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
T = B * 2 * N
res = {}
for var in VARS + VARS:
    eng.lib.gn_debug_set_variant(eng.ctx, 1, var)
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
    print(f"variant {var}: " + ", ".join(f"{r['name'][:26]} {1000 * r['ms'] / r['launches']:.2f} us" for r in big) + f"; all kernels {step:.3f} ms per call", flush=True)
    res.setdefault(var, (idx, score, n, x))
eng.lib.gn_debug_set_variant(eng.ctx, 1, 4)
i0, s0, n0, x0 = res[VARS[0]]
for var in VARS[1:]:
    i1, s1, n1, x1 = res[var]
    same = all(np.array_equal(i0[b, : n0[b]], i1[b, : n1[b]]) for b in range(B)) and np.array_equal(n0, n1)
    print(f"variant {var} vs {VARS[0]}: final features max rel diff {np.abs(x1 - x0).max() / np.abs(x0).max():.3e}; finite {np.isfinite(x1).all()}; matches {n0[:4]} / {n1[:4]}; indices identical: {same}")
for abl in [int(v) for v in os.environ.get("GN_AB_STAMPS", "").split(",") if v]:
    eng.lib.gn_debug_set_variant(eng.ctx, 1, 1000 + abl)
    eng.lib.gn_debug_set_variant(eng.ctx, 4, 3)       # stop after the first attention launch (input projection, k_qkv, attention)
    eng.match(*args)
    torch.cuda.synchronize()
    nwg = (N // 256) * 4 * 2 * B
    ts = eng.debug_read("sim", nwg * 4 * 8 * 2, np.uint32).view(np.int64).reshape(nwg, 4, 8)
    d = np.diff(ts[:, :, :6], axis=2).astype(np.float64)
    print(f"k_attn_pw<{abl} | 8>: " + ", ".join(f"{nm} {np.median(d[:, 0, k]):.0f}" for k, nm in enumerate(["prologue", "sub-tile 0", "tile loop", "-", "rows out"]) if nm != "-")
          + f"; per 64 keys {np.median(d[:, 0, 2] / np.maximum(ts[:, 0, 6], 1)):.0f}; workgroup {np.median(ts[:, 0, 5] - ts[:, 0, 0]):.0f} cycles", flush=True)
eng.lib.gn_debug_set_variant(eng.ctx, 4, 0)
eng.lib.gn_debug_set_variant(eng.ctx, 1, 4)
