"""Generate the golden fixtures of tests/golden/ with the ORACLE (oracle/*.py).

The reference's own path (kornia 0.7.2 + cv2) cannot be imported in the build container and the
reference holds no golden vectors for it, so these fixtures pin the restatement --
"parity unpinned" against the real packages.  Run from the repo root:

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
from oracle import lightglue_sift as lg  # noqa: E402
from oracle import pnp_ransac as pr  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def weights_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()


def main():
    torch.set_num_threads(1)  # fixed reduction order for the committed numbers
    sd = synthetic_state_dict(0)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    tq = torch.from_numpy
    for name, (seed, nq, nr) in {"lightglue_seed0_q96_r80": (7, 96, 80), "lightglue_seed0_q200_r256": (8, 200, 256)}.items():
        p = make_pair(seed, n_q=nq, n_r=nr)
        taps = {}
        mq, mr, sc, idx = lg.pose_node_match(tsd, tq(p.kp_q), tq(p.desc_q), tq(p.size_q), tq(p.angle_q),
                                             tq(p.kp_r), tq(p.desc_r), tq(p.size_r), tq(p.angle_r), taps=taps)
        R, t = pr.compute_pose(K_MATRIX.reshape(-1), mq.numpy(), mr.numpy(), p.dem)
        layer_sums = np.array([[taps[f"layer{i}_0"].double().sum().item(), taps[f"layer{i}_1"].double().sum().item()] for i in range(9)])
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            kp_q=p.kp_q, desc_q=p.desc_q, size_q=p.size_q, angle_q=p.angle_q,
            kp_r=p.kp_r, desc_r=p.desc_r, size_r=p.size_r, angle_r=p.angle_r, dem=p.dem,
            idx=idx.numpy(), scores=sc.numpy(), mkp_q=mq.numpy(), mkp_r=mr.numpy(),
            layer_sums=layer_sums, x_final_0=taps["layer8_0"][0].numpy(), x_final_1=taps["layer8_1"][0].numpy(),
            sim=taps["sim"][0].numpy(), R=R, t=t, K=K_MATRIX)
        print(name, "matches", len(idx))
    # PnP fixtures with gross outliers, non-planar and planar
    for name, (seed, flat) in {"pnp_outliers_dem": (21, False), "pnp_outliers_flat": (22, True)}.items():
        p = make_pair(seed, flat_dem=flat)
        q = np.nonzero(p.gt_q2r >= 0)[0][:300]
        mq, mr = p.kp_q[q].copy(), p.kp_r[p.gt_q2r[q]]
        rs = np.random.default_rng(seed)
        mq[:60] = np.column_stack([rs.uniform(0, 640, 60), rs.uniform(0, 480, 60)]).astype(np.float32)
        x, y = np.floor(mr).astype(int).T
        obj = np.hstack((mr, p.dem[y, x].reshape(-1, 1))).astype(np.float32)
        ok, r, t, inl = pr.solve_pnp_ransac(obj, mq, K_MATRIX, 10)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), obj=obj, img=mq, K=K_MATRIX, rvec=r, tvec=t,
                            R=pr.rodrigues_vec2mat(r), inliers=inl, R_gt=p.R_gt, t_gt=p.t_gt, dem=p.dem, mkp_r=mr)
        print(name, "inliers", len(inl))
    rng = pr.CvRNG()
    stream = [rng.next() for _ in range(32)]
    rng = pr.CvRNG()
    subsets = [pr.get_subset(rng, 100) for _ in range(3)]
    with open(os.path.join(HERE, "cv_rng.json"), "w") as f:
        json.dump({"seed": "0xFFFFFFFFFFFFFFFF", "next_u32": stream, "subsets_count100": subsets,
                   "weights_seed0_sha256": weights_digest(sd)}, f, indent=1)


def make_vo_fixture():
    """tests/golden/vo_knn_seed5_q300_r280.npz: TwistNode 2-NN + ratio test on one synthetic pair (oracle/bf_knn.py)."""
    from oracle import bf_knn
    p = make_pair(5, n_q=300, n_r=280)
    idx, dist = bf_knn.knn_match2(p.desc_q, p.desc_r)
    pairs, pd = bf_knn.ratio_test(idx, dist)
    np.savez_compressed(os.path.join(HERE, "vo_knn_seed5_q300_r280.npz"), desc_q=p.desc_q.astype(np.uint8),
                        desc_r=p.desc_r.astype(np.uint8), nn_idx=idx, nn_dist=dist, pairs=pairs, pair_dist=pd)


def make_stereo_fixture():
    """tests/golden/stereo_rot_seed3.npz: StereoNode gray + DEM stack, rotated 23.5 deg and centre-cropped (oracle/stereo_warp.py)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from test_stereo import _tile
    from oracle import stereo_warp as sw
    bgr, dem = _tile(3, 300, 400)
    ref, do, minv = sw.stereo_reference(bgr, dem, 23.5, (180, 240))
    np.savez_compressed(os.path.join(HERE, "stereo_rot_seed3.npz"), bgr=bgr, dem=dem, angle=np.float64(23.5), crop=np.array([180, 240]),
                        ref=ref, dem_out=do, minv=minv)


def make_sift_fixture():
    """tests/golden/sift_blobs_seed2.npz: cv2.SIFT_create().detectAndCompute on a 96x128 blob image (oracle/sift.py)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from test_sift import blob_image
    from oracle import sift as osift
    img = blob_image(2, 96, 128)
    kp, size, ang, resp, octv, desc = osift.detect_and_compute(img)
    np.savez_compressed(os.path.join(HERE, "sift_blobs_seed2.npz"), image=img, kp=kp, size=size, angle=ang, response=resp, octave=octv,
                        desc=desc.astype(np.uint8))


if __name__ == "__main__":
    main()
    make_vo_fixture()
    make_stereo_fixture()
    make_sift_fixture()
