import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import oracle_match
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import synthetic_state_dict
sd0 = synthetic_state_dict(0)
sd = dict(sd0); sd["input_proj.weight"] = sd0["input_proj.weight"] * np.float32(4.0e5)
tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
pairs = [make_pair(500 + i, n_q=256 - 9 * i, n_r=250) for i in range(2)]
ref = [oracle_match(tsd, p) for p in pairs]
for prec, guard in (("f32", "flag"), ("bf16_attn", "flag"), ("f32x3_bf16_attn", "flag"), ("f16x2_bf16_attn", "sync")):
    eng = PoseEngine(0, max_batch=2, max_kpts=256, precision=prec, state_dict=sd, guard=guard)
    inp = eng.stage_inputs(pairs)
    idx, score, n = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    torch.cuda.synchronize()
    x = eng.debug_read("x", 2 * 2 * 256 * 256)
    for b, (_, _, sc, oidx) in enumerate(ref):
        k = int(n[b])
        a = {(int(q), int(r)) for q, r in idx[b, :k].cpu().numpy()}
        bb = {(int(q), int(r)) for q, r in oidx.numpy()}
        print(prec, guard, "pair", b, "k", k, "oracle", len(bb), "symdiff", len(a ^ bb), "x finite", np.isfinite(x).all(), "xmax", np.abs(x[np.isfinite(x)]).max(), "status", eng.guard_status())
    del eng
