"""Developer tool: time the attention kernel under developer variants (knob 1) in ONE process on ONE box (HIP events per launch).
usage: attn_ab.py [variant ...]   default: 4 (default kernel) 46 (v_pk_fma_f32 in front of the exponentials) 4 46"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

B = 32
variants = [int(v) for v in sys.argv[1:]] or [4, 46, 4, 46]
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i) for i in range(B)])
args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
for _ in range(3):
    eng.match(*args)
for v in variants:
    eng.lib.gn_debug_set_variant(eng.ctx, 1, v)
    eng.match(*args)
    eng.set_kernel_timing(400)
    for _ in range(4):
        eng.match(*args)
    torch.cuda.synchronize()
    rows = {r["name"]: r for r in eng.kernel_table()}
    eng.set_kernel_timing(0)
    att = [r for n, r in rows.items() if "attn" in n]
    step = sum(r["ms"] for r in rows.values()) / 4
    print(f"variant {v}: " + ", ".join(f"{r['name']} {1000 * r['ms'] / r['launches']:.2f} us x {r['launches'] // 4}" for r in att) + f"; all kernels {step:.3f} ms per call")
