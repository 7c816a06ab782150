"""Developer tool: step time of gn_estimate with 1 / 2 / 4 sub-batch streams (the PnP of one group runs under the matcher of the next)."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import K_MATRIX, make_pair
from gisnav_amd.weights import synthetic_state_dict
B = 32
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i) for i in range(B)])
out = eng.alloc_outputs(B)
for nsub in (1, 2, 4, 1):
    eng.set_substreams(nsub)
    for _ in range(3): eng.estimate(inp, K_MATRIX, out=out)
    eng.flush(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.estimate(inp, K_MATRIX, out=out)
    eng.flush(); torch.cuda.synchronize()
    print("substreams", nsub, "ms/step", (time.perf_counter() - t0) / 20 * 1e3, "ok", int(out["ok"].sum()))
for nsub, dj in ((2, True), (2, False)):
    eng.set_substreams(nsub, deferred_join=dj)
    for _ in range(3): eng.estimate(inp, K_MATRIX, out=out)
    eng.flush(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.estimate(inp, K_MATRIX, out=out)
    eng.flush(); torch.cuda.synchronize()
    print("substreams", nsub, "deferred join", dj, "ms/step", (time.perf_counter() - t0) / 20 * 1e3, "ok", int(out["ok"].sum()))
