import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import synthetic_state_dict
prec = sys.argv[1]
sd = synthetic_state_dict(0)
pairs = [make_pair(i) for i in range(32)]
T = 32 * 2 * 1024
eng = PoseEngine(0, max_batch=32, max_kpts=1024, precision=prec, state_dict=sd)
inp = eng.stage_inputs(pairs)
a = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
if len(sys.argv) > 3:
    eng.lib.gn_debug_set_variant(eng.ctx, 1, int(sys.argv[3]))
for nl in (9,):
    eng.set_num_layers(nl)
    eng.match(*a); torch.cuda.synchronize()
    ref = eng.debug_read("x", T * 256).view(np.uint32).reshape(T, 256).copy()
    nbad = 0
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    for r in range(R):
        eng.match(*a); torch.cuda.synchronize()
        x = eng.debug_read("x", T * 256).view(np.uint32).reshape(T, 256)
        d = x != ref
        if d.any():
            nbad += 1
            rows = np.nonzero(d.any(axis=1))[0]
            cols = np.nonzero(d.any(axis=0))[0]
            print(f"  layers={nl} rep {r}: {int(d.sum())} words, {len(rows)} rows [{rows[0]}..{rows[-1]}] slots {sorted(set(rows // 1024))}, cols [{cols[0]}..{cols[-1]}] ({len(cols)})")
    print(f"{prec} layers={nl}: {nbad} of {R} runs differ from the first")
