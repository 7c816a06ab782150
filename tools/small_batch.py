"""Developer tool: latency of small batches (1, 2, 4 pairs; the reference's operating point is 1) in the headline precision, timed like bench.py's extra."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
dev = torch.device("cuda", 0)
sd = synthetic_state_dict(0)
for b in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    r = bench.run_extra(0, sd, f"batch-{b}", b, 1024, "f16x2_f16_attn", 300, 30, dev)
    print(f"batch {b}: {r['ms_per_step']:.4f} ms per call, {r['value']:.1f} pairs/s, poses ok {r['poses_ok_per_step']}", flush=True)
