"""Developer tool: EPnP kernel vs oracle on the RANSAC subsets of a synthetic pair."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from oracle import pnp_ransac as pr  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    eng = PoseEngine(0, max_batch=1, max_kpts=128)
    np.set_printoptions(precision=6, linewidth=200, suppress=False)
    for pi in [int(a) for a in sys.argv[1:]] or [20]:
        p = make_pair(pi)
        q = np.nonzero(p.gt_q2r >= 0)[0]
        mq, mr = p.kp_q[q], p.kp_r[p.gt_q2r[q]]
        x, y = np.floor(mr).astype(int).T
        obj = np.hstack((mr, p.dem[y, x].reshape(-1, 1))).astype(np.float32).astype(np.float64)
        img = mq.astype(np.float64)
        A = K_MATRIX
        und = np.column_stack([(img[:, 0] - A[0, 2]) / A[0, 0], (img[:, 1] - A[1, 2]) / A[1, 1]])
        rng = pr.CvRNG()
        subsets = [pr.get_subset(rng, len(obj), 5) for _ in range(10)]
        pws = np.stack([obj[s] for s in subsets]); us = np.stack([und[s] for s in subsets])
        tp, tu = torch.from_numpy(pws).to(dev), torch.from_numpy(us).to(dev)
        out = torch.zeros((10, 64), dtype=torch.float64, device=dev)
        rc = eng.lib.gn_debug_epnp(eng.ctx, 10, C.c_void_p(tp.data_ptr()), C.c_void_p(tu.data_ptr()), C.c_void_p(out.data_ptr()), eng._stream())
        assert rc == 0
        o = out.cpu().numpy()
        for k, s in enumerate(subsets):
            R, t = pr.epnp(pws[k], us[k])
            dR = np.linalg.norm(o[k, :9].reshape(3, 3) - R)
            print(f"pair {pi} hyp {k}: |dR| {dR:.2e} |dt| {np.linalg.norm(o[k, 9:12] - t):.2e}  gpu cand errs {o[k, 12:15]}")
            if dR > 1e-6:
                print("    gpu betas", o[k, 15:27].reshape(3, 4))
                print("    gpu eig (asc)", o[k, 27:39])
                print("    gpu rho", o[k, 39:45], " L0", o[k, 45:55])
                print("    gpu t", o[k, 9:12], " oracle t", t)


if __name__ == "__main__":
    main()
