import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
pairs = [make_pair(i) for i in range(32)]
T = 32 * 2 * 1024
ref = {}
for prec, var in (("f32", 3), ("f32", 5)):
    eng = PoseEngine(0, max_batch=32, max_kpts=1024, precision=prec, state_dict=sd)
    eng.lib.gn_debug_set_variant(eng.ctx, 0, var)
    inp = eng.stage_inputs(pairs)
    eng.set_num_layers(1)
    eng.lib.gn_debug_set_variant(eng.ctx, 4, 2)
    name, n = ("qkv", T * 768) if prec == "f32" else ("qkb", T * 256)
    snaps = []
    for rep in range(4):
        eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        torch.cuda.synchronize()
        snaps.append(eng.debug_read(name, n).copy())
    if var == 3:
        ref[prec] = snaps[0]
    nd = [int((snaps[0].view(np.int32) != snaps[r].view(np.int32)).sum()) for r in (1, 2, 3)]
    print(prec, "gemm variant", var, name, "differing words run-to-run:", nd, flush=True)
    if nd[0]:
        w = np.nonzero(snaps[0].view(np.int32) != snaps[1].view(np.int32))[0]
        per_row = 768 if prec == "f32" else 256   # words per token row (qkb: 512 bf16 = 256 words)
        rows = np.unique(w // per_row)
        print("   rows", rows[:12], "n rows", len(rows), "cols of first row", (w[w // per_row == rows[0]] % per_row)[:16])
        if prec == "f32":
            for wi in w[:6]:
                r_, c_ = wi // per_row, wi % per_row
                base = r_ * per_row + (c_ & ~3)
                print("     row", r_, "col4", c_ & ~3, "v3", ref[prec][base:base + 4], "run0", snaps[0][base:base + 4], "run1", snaps[1][base:base + 4], "run3", snaps[3][base:base + 4])
    del eng
