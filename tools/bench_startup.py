"""Developer tool: what a node pays ONCE -- context creation (weight upload, fp16 splits, MFMA fragment layouts), the first message, device memory
held -- for the matcher seam, the PoseNode shim and the LoFTR mirror.   python tools/bench_startup.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()


def used_mb():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 2**20


from gisnav_amd import wire  # noqa: E402
from gisnav_amd.matcher import LightGlueMatcher  # noqa: E402
from gisnav_amd.pose_node import PoseNode  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
sd = synthetic_state_dict(0)
p = make_pair(3, n_q=1024, n_r=1024)
for prec in ("f32", "f16x2_f16_attn"):
    for kmax in (1024, 4096):
        m0 = used_mb(); t0 = time.perf_counter()
        m = LightGlueMatcher("sift", params={"filter_threshold": 0.5, "depth_confidence": -1, "width_confidence": -1}, state_dict=sd, max_kpts=kmax, precision=prec).to("cuda:0").eval()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"LightGlueMatcher(...).to(device), precision {prec}, max_kpts {kmax}: {1e3 * (t1 - t0):.0f} ms, {used_mb() - m0:.0f} MB of device memory", flush=True)
        del m
        torch.cuda.empty_cache()
t0 = time.perf_counter()
node = PoseNode(sd, lambda ref: (p.kp_r, p.desc_r, p.size_r, p.angle_r), max_kpts=4096, precision="f16x2_f16_attn")
t1 = time.perf_counter()
cam = wire.CameraInfo(k=K_MATRIX.reshape(-1), height=480, width=640)
msg = wire.OrthoStereoImage(query_sift=wire.pack_keypoints(p.kp_q, p.size_q, p.angle_q, p.desc_q), reference=wire.ImageMsg(p.ref, wire.Stamp(1, 0)), dem=wire.ImageMsg(p.dem, wire.Stamp(1, 0)))
r = node.estimate(cam, msg); t2 = time.perf_counter()
r = node.estimate(cam, msg); t3 = time.perf_counter()
print(f"PoseNode(...): {1e3 * (t1 - t0):.0f} ms; first message (new tile: features staged, buffers pinned) {1e3 * (t2 - t1):.1f} ms; second {1e3 * (t3 - t2):.2f} ms", flush=True)
from gisnav_amd import loftr_synthetic as olf  # noqa: E402
from gisnav_amd.loftr import LoFTR  # noqa: E402
i0, i1 = olf.synthetic_pair(1, 480, 640)
data = {"image0": i0.cuda(), "image1": i1.cuda()}
m0 = used_mb(); t0 = time.perf_counter()
lf = LoFTR(state_dict=olf.synthetic_state_dict(0)).to("cuda:0").eval()
out = lf(data); torch.cuda.synchronize(); t1 = time.perf_counter()
out = lf(data); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"LoFTR 640x480: first call (context, weights, graph capture) {1e3 * (t1 - t0):.0f} ms, second {1e3 * (t2 - t1):.1f} ms, {used_mb() - m0:.0f} MB of device memory")
