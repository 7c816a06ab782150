"""Developer experiment: gn_set_substreams(n) on one 32-pair context (results must stay identical; throughput?)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import K_MATRIX, make_pair
from gisnav_amd.weights import synthetic_state_dict

eng = PoseEngine(0, max_batch=32, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i) for i in range(32)])
ref = None
for n in (1, 2, 4, 8, 4, 2, 1):
    eng.set_substreams(n)
    out = eng.alloc_outputs(32)
    for _ in range(3):
        eng.estimate(inp, K_MATRIX, out=out)
    eng.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.estimate(inp, K_MATRIX, out=out)
    eng.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    cur = {k: v.cpu().numpy().copy() for k, v in out.items()}
    if ref is None:
        ref = cur
    same = all(np.array_equal(ref[k], cur[k]) for k in ref)
    print(f"substreams {n}: {dt * 1e3:.3f} ms/step -> {32 / dt:.0f} pairs/s  identical={same}", flush=True)
