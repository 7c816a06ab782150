import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
pairs = [make_pair(i) for i in range(32)]
T = 32 * 2 * 1024
snap = {}
for var in (3, 5):
    eng = PoseEngine(0, max_batch=32, max_kpts=1024, precision="f32", state_dict=sd)
    eng.lib.gn_debug_set_variant(eng.ctx, 0, var)
    inp = eng.stage_inputs(pairs)
    eng.set_num_layers(1)
    eng.lib.gn_debug_set_variant(eng.ctx, 4, 2)
    eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    torch.cuda.synchronize()
    snap[var] = eng.debug_read("qkv", T * 768).reshape(T, 768).copy()
    cos = eng.debug_read("cos", T * 32).reshape(T, 32).copy(); sin = eng.debug_read("sin", T * 32).reshape(T, 32).copy()
    del eng
good, bad = snap[3], snap[5]
d = np.abs(good - bad)
idx = np.argwhere(d > 1e-3)
print("n wrong", len(idx))
rows = np.unique(idx[:, 0])
for r in rows[:6]:
    cols = idx[idx[:, 0] == r][:, 1]
    print("row", r, "row%128", r % 128, "ncols", len(cols), "cols", cols[:6], "...", cols[-3:])
    c = cols[0]
    c2 = c & ~1
    f = (c2 & 63) >> 1
    ox, oy = good[r, c2], good[r, c2 + 1]
    cs, sn = cos[r, f], sin[r, f]
    vx = ox * cs + oy * sn; vy = -ox * sn + oy * cs       # inverse rotation -> pre-rotary values
    wrong = bad[r, c]
    # hypotheses: rotation with cos/sin of another row
    best = None
    for dr in range(-70, 71):
        rr = r + dr
        if rr < 0 or rr >= T: continue
        for df in range(-2, 3):
            ff = f + df
            if ff < 0 or ff > 31: continue
            cand = (vx * cos[rr, ff] - vy * sin[rr, ff]) if c == c2 else (vy * cos[rr, ff] + vx * sin[rr, ff])
            e = abs(cand - wrong)
            if best is None or e < best[0]: best = (e, dr, df)
    print("   col", c, "good", good[r, c], "bad", wrong, "v", vx, vy, "best stale-cos/sin hypothesis (err, drow, df):", best)
    # hypothesis: value belongs to another row's correct output at same col
    e2 = np.abs(good[max(0, r - 70): r + 71, c] - wrong)
    print("   nearest other-row same-col match: drow", int(np.argmin(e2)) - min(70, r), "err", float(e2.min()))
