"""Developer tool: host time to ENQUEUE one gn_estimate call (no synchronisation inside the loop) next to the wall time per call: is the host
thread or the GPU the limit at small batches?   python tools/host_enqueue.py [batch] [--overlap]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
b = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
eng = PoseEngine(0, max_batch=b, max_kpts=1024, precision="f16x2_f16_attn", state_dict=synthetic_state_dict(0))
if "--overlap" in sys.argv:
    eng.set_overlap(True)
inp = eng.stage_inputs([make_pair(i, n_q=1024, n_r=1024) for i in range(b)])
out = eng.alloc_outputs(b)
for _ in range(30):
    eng.estimate(inp, bench.K_MATRIX, out=out)
eng.flush(); torch.cuda.synchronize()
for n in (1, 4, 16, 64, 256):
    t0 = time.perf_counter()
    for _ in range(n):
        eng.estimate(inp, bench.K_MATRIX, out=out)
    t1 = time.perf_counter()
    eng.flush(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"batch {b}: {n:4d} calls: enqueue {1e3 * (t1 - t0) / n:.4f} ms per call, enqueue + drain {1e3 * (t2 - t0) / n:.4f} ms per call", flush=True)
