"""Developer tool: s_memtime phase stamps of the two k_head_fused sweeps (knob 17), kilo-cycles."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i) for i in range(B)])
args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
eng.match(*args)
eng.lib.gn_debug_set_variant(eng.ctx, 17, 1)
if len(sys.argv) > 2:
    eng.lib.gn_debug_set_variant(eng.ctx, 18, int(sys.argv[2]))
eng.match(*args)
torch.cuda.synchronize()
nrb = 1024 // 128
ts = eng.debug_read("sim", 2 * B * nrb * 8 * 8 * 2, np.uint32).view(np.int64).reshape(2, B, nrb * 8, 8).astype(np.float64) / 1000.0
names = ["prologue (row operand, tile 0)", "column tiles", "row merge", "ticket", "last block: row + column merge", "last block: compaction"]
for sw in range(2):
    t = ts[sw]
    print(f"sweep {sw + 1}: kilo-cycles, median over {B * nrb} workgroups (last-block phases: over the {B} last blocks)")
    for k, n in enumerate(names[: 5 if sw == 0 else 6]):
        d = t[:, :, k + 1] - t[:, :, k]
        ok = (t[:, :, k + 1] > 0) & (t[:, :, k] > 0) & (d > 0) & (d < 1e6)
        print(f"  {n:32s} {np.median(d[ok]) if ok.any() else float('nan'):9.2f}   n={int(ok.sum())}")
    print(f"  kernel span {t[t > 0].max() - t[:, :, 0].min():9.2f}")
