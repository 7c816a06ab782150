"""Developer tool: the two-query-tiles-per-wave attention variant (knob 1 = 45) against the default: same matches, ctx rows within bf16 noise."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import synthetic_state_dict
B = 4
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i, n_q=1024 - 37 * i, n_r=1024 - 11 * i) for i in range(B)])
args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
res = {}
for v in (4, 45):
    eng.lib.gn_debug_set_variant(eng.ctx, 1, v)
    idx, score, n = eng.match(*args); torch.cuda.synchronize()
    res[v] = (idx.clone(), score.clone(), n.clone(), eng.debug_read("x_p", B * 2 * 1024 * 256, np.uint32).copy())
a, b = res[4], res[45]
print("n_match", a[2].tolist(), b[2].tolist())
for i in range(B):
    k = int(a[2][i]); print(i, "indices equal", torch.equal(a[0][i, :k], b[0][i, :k]), "score diff", float((a[1][i, :k] - b[1][i, :k]).abs().max()))
print("x_p identical words", float(np.mean(a[3] == b[3])))
