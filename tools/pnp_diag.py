"""Developer tool: trace RANSAC hypothesis by hypothesis, GPU kernel vs oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from oracle import pnp_ransac as pr  # noqa: E402


def oracle_trace(obj, img, A):
    obj64, img64 = obj.astype(np.float64), img.astype(np.float64)
    und = np.column_stack([(img64[:, 0] - A[0, 2]) / A[0, 0], (img64[:, 1] - A[1, 2]) / A[1, 1]])
    rng = pr.CvRNG()
    out = []
    for it in range(10):
        idx = pr.get_subset(rng, len(obj), 5)
        R, t = pr.epnp(obj64[idx], und[idx])
        rvec = pr.rodrigues_mat2vec(R)
        proj = pr.project_points(obj64, rvec, t, A).astype(np.float32)
        d = img - proj
        err = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
        out.append((idx, int((err <= np.float32(64)).sum()), float(np.sort(err)[-5:].mean())))
    return out


def main():
    dev = torch.device("cuda", 0)
    eng = PoseEngine(0, max_batch=1, max_kpts=1024)
    for pi in [int(a) for a in sys.argv[1:]] or [12, 13]:
        flat = pi >= 1000
        p = make_pair(pi % 1000, flat_dem=flat)
        q = np.nonzero(p.gt_q2r >= 0)[0]
        mq, mr = p.kp_q[q], p.kp_r[p.gt_q2r[q]]
        x, y = np.floor(mr).astype(int).T
        obj = np.hstack((mr, p.dem[y, x].reshape(-1, 1))).astype(np.float32)
        tr = oracle_trace(obj, mq, K_MATRIX)
        print(f"pair {pi}: n={len(obj)} oracle goods per hypothesis:", [g for _, g, _ in tr])
        to, tg = torch.from_numpy(obj[None]).to(dev), torch.from_numpy(mq[None]).to(dev)
        n = torch.tensor([len(obj)], dtype=torch.int32, device=dev)
        gl = []
        for it in range(1, 11):
            R, t, ninl, ok = eng.pnp_ransac(to, tg, n, K_MATRIX, iterations=it)
            gl.append(int(ninl[0]))
        print("   gpu best-so-far after k hypotheses:", gl)
        best = 0
        ol = []
        niters = 10
        for k, (_, g, _) in enumerate(tr):
            if k >= niters:
                ol.append(best); continue
            if g > max(best, 4):
                best = g
                niters = pr.ransac_update_num_iters(0.99, (len(obj) - g) / len(obj), 5, niters)
            ol.append(best)
        print("   oracle best-so-far               :", ol)
        R, t, ninl, ok = eng.pnp_ransac(to, tg, n, K_MATRIX, iterations=10)
        oko, ro, t_o, inl = pr.solve_pnp_ransac(obj, mq, K_MATRIX, 10)
        print("   final: gpu |R-Rgt|", float(np.linalg.norm(R[0].cpu().numpy() - p.R_gt)), "oracle |R-Rgt|",
              float(np.linalg.norm(pr.rodrigues_vec2mat(ro) - p.R_gt)), "gpu ninl", int(ninl[0]), "oracle ninl", len(inl))


if __name__ == "__main__":
    main()
