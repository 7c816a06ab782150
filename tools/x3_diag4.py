import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
pairs = [make_pair(i) for i in range(32)]
T = 32 * 2 * 1024
ref = None
for prec in ("bf16_attn", "f32x3_bf16_attn"):
    eng = PoseEngine(0, max_batch=32, max_kpts=1024, precision=prec, state_dict=sd)
    inp = eng.stage_inputs(pairs)
    xs = []
    for rep in range(6):
        eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        torch.cuda.synchronize()
        xs.append(eng.debug_read("x", T * 256).copy())
    sc = np.abs(xs[-1]).max()
    print(prec, "run r vs last run, max abs diff / max|x|:", [float(np.abs(x - xs[-1]).max() / sc) for x in xs[:-1]], flush=True)
    if ref is None:
        ref = xs[-1]
    else:
        print("   vs bf16_attn(v3) last run:", [float(np.abs(x - ref).max() / sc) for x in xs], flush=True)
    del eng
