"""Developer tool: per-kernel launch times (HIP events) under different values of one developer knob, in ONE process on ONE box.
usage: knob_ab.py <knob> <value> [<value> ...]      e.g.  knob_ab.py 22 0 5 0 5   (start stagger of k_qkv / k_ffn_fused off / on)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

B = int(os.environ.get("GN_AB_BATCH", "32"))
knob = int(sys.argv[1])
values = [int(v) for v in sys.argv[2:]]
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i) for i in range(B)])
args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
for _ in range(3):
    eng.match(*args)
for v in values:
    eng.lib.gn_debug_set_variant(eng.ctx, knob, v)
    eng.match(*args)
    eng.set_kernel_timing(400)
    for _ in range(4):
        eng.match(*args)
    torch.cuda.synchronize()
    rows = eng.kernel_table()
    eng.set_kernel_timing(0)
    step = sum(r["ms"] for r in rows) / 4
    big = sorted(rows, key=lambda r: -r["ms"])[:4]
    print(f"knob {knob} = {v}: " + ", ".join(f"{r['name'].split('<')[0]}<{r['name'].split('<')[1][:14] if '<' in r['name'] else ''} {1000 * r['ms'] / r['launches']:.2f} us" for r in big)
          + f"; all kernels {step:.3f} ms per call")
