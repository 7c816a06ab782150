import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import synthetic_state_dict
B = 32
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i) for i in range(B)])
args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
T = B * 2 * 1024
eng.lib.gn_debug_set_variant(eng.ctx, 4, 3)   # stop after prep, input_proj, first k_qkv
ref = None
for r in range(6):
    eng.match(*args); torch.cuda.synchronize()
    qk = eng.debug_read("qkb", T * 256, np.uint32).reshape(T, 256).copy()     # bf16 pairs: [T][512 bf16]
    vt = eng.debug_read("vtb", T * 128, np.uint32).copy()
    if ref is None: ref = (qk, vt); continue
    dq = qk != ref[0]; dv = vt != ref[1]
    rows = np.nonzero(dq.any(axis=1))[0]; cols = np.nonzero(dq.any(axis=0))[0]
    print(f"run {r}: qk words differing {int(dq.sum())} (rows {len(rows)}, first {rows[:6]}, word-cols {cols[:12]} .. n={len(cols)}), vt words differing {int(dv.sum())}")
