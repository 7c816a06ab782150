"""Developer tool: s_memtime phase stamps of k_pnp_hyp / k_pnp_refine (knob 15) on the bench workload (kilo-cycles)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
FLAT = len(sys.argv) > 2 and sys.argv[2] == "flat"     # flat DEM: the planar (homography) start instead of the DLT
inp = eng.stage_inputs([make_pair(i, flat_dem=FLAT) for i in range(B)])
out = eng.alloc_outputs(B)
eng.estimate(inp, K_MATRIX, out=out)
eng.lib.gn_debug_set_variant(eng.ctx, 15, 1)
eng.estimate(inp, K_MATRIX, out=out)
torch.cuda.synchronize()
ts = eng.debug_read("sim", (B * 16 + B) * 16 * 2, np.uint32).view(np.int64).reshape(-1, 16)
sweeps = ts[: B * 16].reshape(B, 16, 16)[:, :10, 8]
hyp = ts[: B * 16].reshape(B, 16, 16)[:, :10, :8].astype(np.float64) / 1000.0     # s_memtime ticks at the shader clock here: kilo-cycles
names = ["setup+MtM", "sym_eig 12x12", "null space + L", "3 x svd (lockstep)", "3 x gauss-newton", "3 x R,t", "choose + score all points"]
d = np.diff(hyp, axis=2)
print(f"k_pnp_hyp phases, kilo-cycles: median / max over {B * 10} hypothesis waves   (inliers: {out['n_inliers'].cpu().numpy()[:8]})")
for k, n in enumerate(names):
    print(f"  {n:20s} {np.median(d[:, :, k]):8.2f} {d[:, :, k].max():8.2f}")
print(f"  {'total':20s} {np.median(hyp[:, :, 7] - hyp[:, :, 0]):8.2f} {(hyp[:, :, 7] - hyp[:, :, 0]).max():8.2f}")
print(f"  sym_eig sweeps: median {np.median(sweeps):.0f}, max {sweeps.max()}")
ref = ts[B * 16: B * 16 + B, :6].astype(np.float64) / 1000.0
d = np.diff(ref, axis=1)
print("k_pnp_refine phases, kilo-cycles: median / max over pairs")
for k, n in enumerate(["select", "PCA", "init (planar / DLT)", "Levenberg-Marquardt", "Rodrigues"]):
    print(f"  {n:20s} {np.median(d[:, k]):8.2f} {d[:, k].max():8.2f}")
print(f"  {'total':20s} {np.median(ref[:, 5] - ref[:, 0]):8.2f}")
