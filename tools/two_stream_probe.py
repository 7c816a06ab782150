"""Developer experiment: one 32-pair context on one stream vs two 16-pair contexts on two streams (are the lockstep
load / compute / store phases of the GEMMs hidden by running two half-batches out of phase?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import K_MATRIX, make_pair
from gisnav_amd.weights import synthetic_state_dict

sd = synthetic_state_dict(0)
pairs = [make_pair(i) for i in range(32)]


def run(nctx, steps=20):
    per = 32 // nctx
    engs = [PoseEngine(0, max_batch=per, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=sd) for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    inps = [e.stage_inputs(pairs[i * per:(i + 1) * per]) for i, e in enumerate(engs)]
    outs = [e.alloc_outputs(per) for e in engs]
    torch.cuda.synchronize()
    def step():
        for e, s, i, o in zip(engs, streams, inps, outs):
            with torch.cuda.stream(s):
                e.estimate(i, K_MATRIX, out=o)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{nctx} context(s) x {per} pairs: {dt * 1e3:.3f} ms per 32 pairs -> {32 / dt:.0f} pairs/s", flush=True)
    del engs


for n in (1, 4, 8, 4, 1):
    run(n)
