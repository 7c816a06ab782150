"""Developer study: k_qkv with TWO partial products (x_h w_h + x_h w_m; the x_m w_h product dropped -- developer knob 27 = 2)
against the three-product split, in the headline precision: index mismatches against the oracle on low- / mid-margin weights over 6 x 4 pairs, and
the kernel's time at batch 32.  Writes gpurun_out/qkv2_study.json."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import oracle_match  # noqa: E402
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
from test_gpu_parity2 import LOW_MARGIN, MID_MARGIN, _mismatches  # noqa: E402

out = {}
for name, kw, th in (("low_margin", LOW_MARGIN, 0.0), ("mid_margin", MID_MARGIN, 0.01)):
    sd = synthetic_state_dict(0, **kw)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    res = {3: [0, 0], 2: [0, 0]}
    eng = PoseEngine(0, max_batch=4, max_kpts=512, precision="f16x2_f16_attn", state_dict=sd, filter_threshold=th)
    eng.lib.gn_debug_set_variant(eng.ctx, 19, 2)      # k_qkv at this small batch too
    for rep in range(6):
        pairs = [make_pair(400 + 10 * rep + i, n_q=512 - 31 * i, n_r=512 - 17 * i) for i in range(4)]
        ref = [oracle_match(tsd, p, filter_threshold=th) for p in pairs]
        inp = eng.stage_inputs(pairs)
        for prod in (3, 2):
            eng.lib.gn_debug_set_variant(eng.ctx, 27, prod)
            idx, score, n_match = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
            torch.cuda.synchronize()
            for b, (_, _, _, oidx) in enumerate(ref):
                d, t = _mismatches(idx[b].cpu().numpy(), int(n_match[b]), oidx.numpy())
                res[prod][0] += d; res[prod][1] += t
    eng.lib.gn_debug_set_variant(eng.ctx, 27, 3)
    out[name] = {"three_products": {"index_mismatches": res[3][0], "oracle_matches": res[3][1]}, "two_products": {"index_mismatches": res[2][0], "oracle_matches": res[2][1]}}
    print(name, out[name], flush=True)
    del eng
B = 32
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_f16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i) for i in range(B)])
args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
base = None
for prod in (3, 2, 3, 2):
    eng.lib.gn_debug_set_variant(eng.ctx, 27, prod)
    for _ in range(2):
        idx, score, n = (v.cpu().numpy().copy() for v in eng.match(*args))
    eng.set_kernel_timing(400)
    for _ in range(4):
        eng.match(*args)
    torch.cuda.synchronize()
    rows = eng.kernel_table(); eng.set_kernel_timing(0)
    q = {r["name"]: round(1000 * r["ms"] / r["launches"], 2) for r in rows if r["name"].startswith("k_qkv")}
    step = sum(r["ms"] for r in rows) / 4
    if prod == 3 and base is None:
        base = (idx, n)
    same = all(np.array_equal(base[0][b, : base[1][b]], idx[b, : n[b]]) for b in range(B)) and np.array_equal(base[1], n)
    out.setdefault("batch32", []).append({"products": prod, "k_qkv_us": q, "all_kernels_ms": round(step, 3), "indices_identical_to_three_products": bool(same)})
    print(out["batch32"][-1], flush=True)
eng.lib.gn_debug_set_variant(eng.ctx, 27, 3)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "qkv2_study.json"), "w"), indent=1)
