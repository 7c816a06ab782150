"""Developer tool: soak -- many messages through the node shim, the seams and LoFTR, device memory and host RSS before / after (a node runs for hours).
   python tools/soak.py [messages]"""
import os, sys, time, gc
import numpy as np, torch, psutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd import wire  # noqa: E402
from gisnav_amd.matcher import LightGlueMatcher  # noqa: E402
from gisnav_amd.pose import compute_pose, init  # noqa: E402
from gisnav_amd.pose_node import PoseNode  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.upload import PinnedUploader  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
proc = psutil.Process()


def snap():
    gc.collect(); torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 2**20, proc.memory_info().rss / 2**20


sd = synthetic_state_dict(0)
pairs = [make_pair(200 + i, n_q=1024 - 16 * i, n_r=1024) for i in range(6)]
node = PoseNode(sd, lambda ref: (pairs[0].kp_r, pairs[0].desc_r, pairs[0].size_r, pairs[0].angle_r), max_kpts=1024, precision="f16x2_f16_attn")
cam = wire.CameraInfo(k=K_MATRIX.reshape(-1), height=480, width=640)
msgs = [wire.OrthoStereoImage(query_sift=wire.pack_keypoints(p.kp_q, p.size_q, p.angle_q, p.desc_q), reference=wire.ImageMsg(pairs[0].ref, wire.Stamp(1 + i // 3, 0)),
                              dem=wire.ImageMsg(pairs[0].dem, wire.Stamp(1 + i // 3, 0))) for i, p in enumerate(pairs)]
for i in range(200): node.estimate(cam, msgs[i % 6])
d0, r0 = snap(); t0 = time.perf_counter(); ok = 0
for i in range(N): ok += node.estimate(cam, msgs[i % 6]) is not None
d1, r1 = snap()
print(f"PoseNode.estimate x {N}: {1e3 * (time.perf_counter() - t0) / N:.3f} ms per message, poses {ok}; device memory {d0:.0f} -> {d1:.0f} MB, host RSS {r0:.0f} -> {r1:.0f} MB", flush=True)


class Cam:
    k = K_MATRIX.reshape(-1)


m = LightGlueMatcher("sift", params={"filter_threshold": 0.5, "depth_confidence": -1, "width_confidence": -1}, state_dict=sd, max_kpts=1024, precision="f16x2_f16_attn").to("cuda:0").eval()
init(0, 1024); up = PinnedUploader("cuda:0"); p = pairs[0]
laf = np.zeros((1024, 2, 3), np.float32); laf[:, 0, 0] = p.size_r; laf[:, 1, 1] = p.size_r; laf[:, :, 2] = p.kp_r
lq = np.zeros((len(p.kp_q), 2, 3), np.float32); lq[:, 0, 0] = p.size_q; lq[:, 1, 1] = p.size_q; lq[:, :, 2] = p.kp_q
def seams():
    d, idx = m(up("dq", p.desc_q), up("dr", p.desc_r), up("lq", lq)[None], up("lr", laf)[None])
    idx = idx.cpu().numpy()
    return compute_pose(Cam, p.kp_q[idx[:, 0]], p.kp_r[idx[:, 1]], p.dem) if len(idx) >= 15 else None
for _ in range(100): seams()
d0, r0 = snap(); t0 = time.perf_counter()
for _ in range(N // 3): seams()
d1, r1 = snap()
print(f"seams B1 + B2 x {N // 3}: {1e3 * (time.perf_counter() - t0) / (N // 3):.3f} ms per frame; device memory {d0:.0f} -> {d1:.0f} MB, host RSS {r0:.0f} -> {r1:.0f} MB", flush=True)
from gisnav_amd import loftr_synthetic as olf  # noqa: E402
from gisnav_amd.loftr import LoFTR  # noqa: E402
i0, i1 = olf.synthetic_pair(1, 480, 640)
data = {"image0": i0.cuda(), "image1": i1.cuda()}
lf = LoFTR(state_dict=olf.synthetic_state_dict(0), arithmetic="split_fp16").to("cuda:0").eval()
for _ in range(20): lf(data)
d0, r0 = snap(); t0 = time.perf_counter()
for _ in range(N // 12): out = lf(data)
d1, r1 = snap()
print(f"LoFTR 640x480 x {N // 12}: {1e3 * (time.perf_counter() - t0) / (N // 12):.2f} ms per pair, {len(out['confidence'])} matches; device memory {d0:.0f} -> {d1:.0f} MB, host RSS {r0:.0f} -> {r1:.0f} MB")
