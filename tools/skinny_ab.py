"""Developer tool: A/B of the small-grid kernels' variants (knob 33 = 1 + 16 * variant, gn_skinny.hip) in ONE process on one box (boxes differ by
~10 % in clock): per-call wall time of `reps` interleaved rounds per variant.   python tools/skinny_ab.py [batch] [variants ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
dev = torch.device("cuda", 0)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variants = [int(v) for v in sys.argv[2:]] or list(range(8))
eng = PoseEngine(0, max_batch=b, max_kpts=1024, precision="f16x2_f16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i, n_q=1024, n_r=1024) for i in range(b)])
out = eng.alloc_outputs(b)
best = {v: 1e9 for v in variants + [-1]}
for rep in range(5):
    for v in variants + [-1]:
        eng.lib.gn_debug_set_variant(eng.ctx, 33, 0 if v < 0 else 1 + 16 * v)
        elapsed, _ = bench.timed_steps(eng, inp, out, 200, 20, dev)
        best[v] = min(best[v], elapsed / 200 * 1e3)
for v in variants + [-1]:
    print(f"batch {b} variant {v:2d}{' (skinny off)' if v < 0 else ''}: {best[v]:.4f} ms per call (best of 5 x 200)", flush=True)
