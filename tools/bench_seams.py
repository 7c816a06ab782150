"""Developer tool: wall-clock latency of one frame through seams B1 + B2 the way pose_node.py:254-308 drives them after the two-import change
(INTEGRATION.md section 2): numpy keypoints / descriptors -> torch tensors on the device -> LightGlueMatcher(...) -> matched points back to numpy ->
compute_pose -> (R, t).  Timed twice: with the reference's own upload lines (`torch.tensor(array).to(device)`, fresh pageable memory per call) and with
those four lines going through `gisnav_amd.upload.PinnedUploader`.   python tools/bench_seams.py [n]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.matcher import LightGlueMatcher  # noqa: E402
from gisnav_amd.pose import compute_pose, init  # noqa: E402
from gisnav_amd.upload import PinnedUploader  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402


class Cam:
    k = K_MATRIX.reshape(-1)


n_msg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
m = LightGlueMatcher("sift", params={"filter_threshold": 0.5, "depth_confidence": -1, "width_confidence": -1}, state_dict=synthetic_state_dict(0),
                     max_kpts=1024, precision="f16x2_f16_attn").to(dev).eval()
init(0, 1024)
p = make_pair(3, n_q=1024, n_r=1024)


def laf(kp, size, ang):   # laf_from_center_scale_ori as pose_node.py:266-283 builds it (host side of the reference, kept as it is)
    a = np.deg2rad(ang)
    L = np.zeros((len(kp), 2, 3), np.float32)
    L[:, 0, 0], L[:, 0, 1], L[:, 1, 0], L[:, 1, 1] = size * np.cos(a), size * np.sin(a), -size * np.sin(a), size * np.cos(a)
    L[:, :, 2] = kp
    return L


up = PinnedUploader(dev)


def one(pinned):
    t = [time.perf_counter()]
    if pinned:
        dq, dr = up("desc_q", p.desc_q), up("desc_r", p.desc_r)
        lq, lr = up("laf_q", laf(p.kp_q, p.size_q, p.angle_q))[None], up("laf_r", laf(p.kp_r, p.size_r, p.angle_r))[None]
    else:
        dq = torch.tensor(p.desc_q).to(dev); dr = torch.tensor(p.desc_r).to(dev)                      # pose_node.py:254-265 (the reference's own host code)
        lq = torch.tensor(laf(p.kp_q, p.size_q, p.angle_q)).to(dev)[None]; lr = torch.tensor(laf(p.kp_r, p.size_r, p.angle_r)).to(dev)[None]
    t.append(time.perf_counter())
    dists, idx = m(dq, dr, lq, lr)                                                                # seam B1
    t.append(time.perf_counter())
    idx = idx.cpu().numpy()                                                                       # pose_node.py:296-297
    t.append(time.perf_counter())
    r = compute_pose(Cam, p.kp_q[idx[:, 0]], p.kp_r[idx[:, 1]], p.dem) if len(idx) >= 15 else None   # seam B2
    t.append(time.perf_counter())
    return r, np.diff(t) * 1e3


for pinned in (False, True):
    for _ in range(10):
        r, _ = one(pinned)
    assert r is not None
    T = np.array([one(pinned)[1] for _ in range(n_msg)])
    tot = T.sum(1)
    print(f"seams B1 + B2, uploads {'through PinnedUploader' if pinned else 'as the reference writes them (pageable)'}: {n_msg} frames of 1024 keypoints, "
          f"median {np.median(tot):.3f} ms, p95 {np.percentile(tot, 95):.3f} ms, frames above 5 ms: {int((tot > 5).sum())}")
    for k, name in enumerate(("uploads (pose_node.py:254-265)", "LightGlueMatcher.__call__ (B1)", "idx.cpu() (pose_node.py:296-297)", "compute_pose (B2)")):
        print(f"   {name}: median {np.median(T[:, k]):.3f} ms, calls above 5 ms: {int((T[:, k] > 5).sum())} (max {T[:, k].max():.1f} ms)")
