import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
pairs = [make_pair(i) for i in range(32)]
T = 32 * 2 * 1024
eng = PoseEngine(0, max_batch=32, max_kpts=1024, precision="f32x3_bf16_attn", state_dict=sd)
inp = eng.stage_inputs(pairs)
eng.set_num_layers(1)
# launches: 1 input_proj(x) 2 Wqkv(qkb,vtb) 3 self attn(ctx) 4 out_proj(msg) 5 ffn0(h) 6 ffn3(x) 7 cross proj(qkb,vtb) 8 cross attn(ctx) 9 to_out(msg) 10 ffn0(h) 11 ffn3(x)
outs = {1: ["x"], 2: ["qkb", "vtb"], 3: ["ctx"], 4: ["msg"], 5: ["h"], 6: ["x"], 7: ["qkb", "vtb"], 8: ["ctx"], 9: ["msg"], 10: ["h"], 11: ["x"]}
size = {"x": T * 256, "qkb": T * 256, "vtb": T * 128, "ctx": T * 256, "msg": T * 256, "h": T * 512}
for stop in range(1, 12):
    eng.lib.gn_debug_set_variant(eng.ctx, 4, stop)
    snaps = []
    for rep in range(6):
        eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        torch.cuda.synchronize()
        if rep >= 2:
            snaps.append({k: eng.debug_read(k, size[k]).copy() for k in outs[stop]})
    res = {k: [int((snaps[0][k].view(np.int32) != s[k].view(np.int32)).sum()) for s in snaps[1:]] for k in outs[stop]}
    print("stop_after", stop, res, flush=True)
