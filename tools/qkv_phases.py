"""Developer tool: s_memtime phase stamps of k_qkv (knob 20): the first launch of one bench-sized call (self block), kilo-cycles."""
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
eng.lib.gn_debug_set_variant(eng.ctx, 19, 2)      # force k_qkv whatever the batch size
eng.lib.gn_debug_set_variant(eng.ctx, 20, 1)
eng.lib.gn_debug_set_variant(eng.ctx, 4, int(sys.argv[2]) if len(sys.argv) > 2 else 3)       # stop after the first projection launch (prep, input_proj, k_qkv)
eng.match(*args)
torch.cuda.synchronize()
nb = B * 2 * 1024 // 128
ts = eng.debug_read("sim", nb * 8 * 8 * 2, np.uint32).view(np.int64).reshape(nb, 8, 8).astype(np.float64) / 1000.0
names = ["token tile -> LDS", "pass 0 k-loop", "pass 0 epilogue", "pass 1 k-loop", "pass 1 epilogue", "pass 2 k-loop", "pass 2 epilogue"]
d = np.diff(ts, axis=2)
print("k_qkv phases, kilo-cycles: median over workgroups (wave 0) / max over waves (median)")
for k, n in enumerate(names):
    print(f"  {n:20s} {np.median(d[:, 0, k]):9.2f} {np.median(d[:, :, k].max(axis=1)):9.2f}")
print("  workgroup total     ", np.median(ts[:, 0, 7] - ts[:, 0, 0]), " kernel span", ts[:, :, 7].max() - ts[:, :, 0].min())
st = np.sort(ts[:, 0, 0] - ts[:, :, 0].min())
print("  workgroup start times (quartiles):", st[[0, nb // 4, nb // 2, 3 * nb // 4, -1]])
