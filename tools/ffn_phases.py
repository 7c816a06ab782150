"""Developer tool: per-phase shader-clock stamps of k_ffn_fused (ablation 8), from the last FFN launch of one bench-sized call."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i) for i in range(B)])
args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
eng.match(*args)
eng.lib.gn_debug_set_variant(eng.ctx, 12, 8)
eng.lib.gn_debug_set_variant(eng.ctx, 4, 5)       # stop after the first FFN launch: the stamps in `sim` are not overwritten by the head
eng.match(*args)
torch.cuda.synchronize()
TOK = 64 if B * 2 * 1024 // 64 >= 256 else 32          # the launcher's shape rule (gn_ffn.hip)
nb = B * 2 * 1024 // TOK
ts = eng.debug_read("sim", nb * 8 * 8 * 2, np.uint32).view(np.int64).reshape(nb, 8, 8)
d = np.diff(ts, axis=2).astype(np.float64)
names = ["prologue", "gemm1", "ln+gelu", "publish", "gemm2", "barrier", "epilogue"]
print("phase cycles, median over blocks (wave 0) / max over waves:")
for k, n in enumerate(names):
    print(f"  {n:9s} {np.median(d[:, 0, k]):9.0f}   {np.median(d[:, :, k].max(axis=1)):9.0f}")
tot = ts[:, 0, 7] - ts[:, 0, 0]
print("block total median", np.median(tot), " kernel span (max end - min start)", ts[:, :, 7].max() - ts[:, :, 0].min())
starts = np.sort(ts[:, 0, 0] - ts[:, :, 0].min())
print("block start times (cycles) quartiles:", starts[[0, nb // 4, nb // 2, 3 * nb // 4, -1]])
