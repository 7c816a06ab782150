"""Developer tool: per-kernel average durations of one 32-pair step with/without a debug knob (torch profiler free:
uses the library's kernel timing classes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import K_MATRIX, make_pair
from gisnav_amd.weights import synthetic_state_dict

B = 32
eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i) for i in range(B)])
out = eng.alloc_outputs(B)
for which, val in [(9, 0), (9, 1), (9, 0)]:
    eng.lib.gn_debug_set_variant(eng.ctx, which, val)
    for _ in range(2):
        eng.estimate(inp, K_MATRIX, out=out)
    eng.set_kernel_timing(93 * 5)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        eng.estimate(inp, K_MATRIX, out=out)
    e.record(); torch.cuda.synchronize()
    g = eng.kernel_stats(0); a = eng.kernel_stats(1)
    eng.set_kernel_timing(0)
    print(f"knob {which}={val}: step {s.elapsed_time(e) / 5:.3f} ms  gemm {g['ms'] / 5:.3f} ms/step  attn {a['ms'] / 5:.3f} ms/step", flush=True)
