"""Developer tool: wall-clock latency of ONE ROS message through the B3 shim (`PoseNode.estimate`: wire bytes -> (R, t) as numpy), the way the reference's
callback is driven: 1024-keypoint frames against a cached tile (same stamp), headline precision.  Compare with bench.py's batch-1 line, which times the
device work alone.   python tools/bench_pose_node.py [n_messages]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd import wire  # noqa: E402
from gisnav_amd.pose_node import PoseNode  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
n_msg = int(sys.argv[1]) if len(sys.argv) > 1 else 300
p = make_pair(3, n_q=1024, n_r=1024)
node = PoseNode(synthetic_state_dict(0), lambda ref: (p.kp_r, p.desc_r, p.size_r, p.angle_r), max_kpts=1024, precision="f16x2_f16_attn")
cam = wire.CameraInfo(k=K_MATRIX.reshape(-1), height=480, width=640)
msgs = []
for i in range(8):        # eight different frames over the same tile
    q = make_pair(3, n_q=1024, n_r=1024) if i == 0 else make_pair(3 + 0, n_q=1024 - 8 * i, n_r=1024)
    msgs.append(wire.OrthoStereoImage(query_sift=wire.pack_keypoints(q.kp_q, q.size_q, q.angle_q, q.desc_q), reference=wire.ImageMsg(p.ref, wire.Stamp(1, 0)),
                                      dem=wire.ImageMsg(p.dem, wire.Stamp(100 + i, 0))))      # (upstream re-stamps the DEM with every message: stereo_node.py:272)
for i in range(20):
    r = node.estimate(cam, msgs[i % 8])
assert r is not None
torch.cuda.synchronize()
ts = []
for i in range(n_msg):
    t0 = time.perf_counter()
    r = node.estimate(cam, msgs[i % 8])
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print(f"PoseNode.estimate, {n_msg} messages: median {np.median(ts):.3f} ms, mean {ts.mean():.3f} ms, p95 {np.percentile(ts, 95):.3f} ms per message "
      f"({1e3 / np.median(ts):.0f} messages/s); last pose ok {r is not None}, matches {node.last_num_matches}")
