"""Developer tool: bitwise repeatability of the matcher over many runs (guards the packed-f32 arithmetic in k_ffn_fused and every other
kernel against timing-dependent results): the residual stream after 9 layers and the matches must be identical every time."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

B = 32
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision=os.environ.get("GN_PREC", "f16x2_f16_attn"), state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i, n_q=1024 - (i % 5) * 17, n_r=1024 - (i % 3) * 29) for i in range(B)])
args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
ref = None
bad = 0
for r in range(runs):
    idx, score, n = eng.match(*args)
    torch.cuda.synchronize()
    x = eng.debug_read("x", B * 2 * 1024 * 256).view(np.uint32).copy()
    nn = n.cpu().numpy().copy()
    valid = np.arange(idx.shape[1])[None, :] < nn[:, None]            # the lists are filled up to n_match only
    cur = (x, np.where(valid[:, :, None], idx.cpu().numpy(), 0), np.where(valid, score.cpu().numpy().view(np.uint32), 0), nn)
    if ref is None:
        ref = cur
        continue
    same = all(np.array_equal(a, b) for a, b in zip(ref, cur))
    if not same:
        bad += 1
        print(f"run {r}: differs in {int((ref[0] != cur[0]).sum())} words of x, matches equal: {np.array_equal(ref[1], cur[1])}")
print(f"{runs} runs, {bad} differing from the first")
sys.exit(1 if bad else 0)
