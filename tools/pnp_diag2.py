"""Developer tool: the planar-PnP sweep of tests/test_gpu_parity2.py, seed by seed, with a per-hypothesis trace of the seeds
where the kernel and the oracle disagree (inlier count or pose)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from oracle import pnp_ransac as pr  # noqa: E402


def case(seed):
    p = make_pair(seed, n_q=256, n_r=256, flat_dem=True)
    q = np.nonzero(p.gt_q2r >= 0)[0]
    mq, mr = p.kp_q[q].copy(), p.kp_r[p.gt_q2r[q]]
    rs = np.random.default_rng(seed)
    no = int(len(q) * rs.uniform(0.0, 0.35))
    mq[:no] = np.column_stack([rs.uniform(0, 640, no), rs.uniform(0, 480, no)]).astype(np.float32)
    mq[no:] += rs.normal(0, rs.uniform(0.0, 1.5), (len(q) - no, 2)).astype(np.float32)
    obj = np.hstack((mr, np.zeros((len(mr), 1), np.float32))).astype(np.float32)
    return obj, mq


def main():
    dev = torch.device("cuda", 0)
    eng = PoseEngine(0, max_batch=1, max_kpts=256)
    A = K_MATRIX
    bad = []
    for seed in range(1000, 1224):
        obj, mq = case(seed)
        oko, r, t, inl = pr.solve_pnp_ransac(obj, mq, A, 10)
        to, tg = torch.from_numpy(obj[None]).to(dev), torch.from_numpy(mq[None]).to(dev)
        n = torch.tensor([len(obj)], dtype=torch.int32, device=dev)
        R, tt, ninl, ok = eng.pnp_ransac(to, tg, n, A)
        if not oko:
            continue
        dR = float(np.linalg.norm(R[0].cpu().numpy() - pr.rodrigues_vec2mat(r)))
        if int(ninl[0]) != len(inl) or dR > 1e-8:
            bad.append(seed)
            print(f"seed {seed}: n={len(obj)} gpu ninl {int(ninl[0])} oracle {len(inl)} dR {dR:.3e}")
            obj64, img64 = obj.astype(np.float64), mq.astype(np.float64)
            und = np.column_stack([(img64[:, 0] - A[0, 2]) / A[0, 0], (img64[:, 1] - A[1, 2]) / A[1, 1]])
            rng = pr.CvRNG()
            best = 0
            for it in range(10):
                idx = pr.get_subset(rng, len(obj), 5)
                dbg = {}
                Ro, to_ = pr.epnp(obj64[idx], und[idx], dbg=dbg)
                rv = pr.rodrigues_mat2vec(Ro)
                proj = pr.project_points(obj64, rv, to_, A).astype(np.float32)
                d = mq - proj
                err = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
                good = int((err <= np.float32(64)).sum())
                pw = torch.from_numpy(np.ascontiguousarray(obj64[idx])).to(dev)
                us = torch.from_numpy(np.ascontiguousarray(und[idx])).to(dev)
                o = torch.zeros(64, dtype=torch.float64, device=dev)
                eng.lib.gn_debug_epnp(eng.ctx, 1, pw.data_ptr(), us.data_ptr(), o.data_ptr(), eng._stream())
                torch.cuda.synchronize()
                og = o.cpu().numpy()
                Rg, tgp = og[:9].reshape(3, 3), og[9:12]
                rvg = pr.rodrigues_mat2vec(Rg)
                projg = pr.project_points(obj64, rvg, tgp, A).astype(np.float32)
                dg = mq - projg
                errg = (dg[:, 0] * dg[:, 0] + dg[:, 1] * dg[:, 1]).astype(np.float32)
                goodg = int((errg <= np.float32(64)).sum())
                near = int(((np.abs(err - 64) < 1e-3)).sum())
                print(f"   hyp {it}: subset {idx} oracle good {good} gpu-epnp good {goodg} |dR_epnp| {np.linalg.norm(Rg - Ro):.2e} |dt| {np.linalg.norm(tgp - np.ravel(to_)):.2e} "
                      f"errs within 1e-3 of 64: {near}  repr errs (oracle N=1,2,3): {[float(c[2]) for c in dbg['cands']]}")
    print("bad seeds:", bad)


if __name__ == "__main__":
    main()
