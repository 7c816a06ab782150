"""Stage-by-stage GPU-vs-oracle diagnostic (developer tool; prints, never asserts).

    python tools/gpu_diag.py [--n 256] [--precision f32]
"""
import argparse
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd import _lib  # noqa: E402
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
from oracle import lightglue_sift as lg  # noqa: E402
from oracle import pnp_ransac as pr  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--batch", type=int, default=2)
    args = ap.parse_args()
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    print("device:", torch.cuda.get_device_name(0), flush=True)
    sd = synthetic_state_dict(0)
    eng = PoseEngine(0, max_batch=args.batch, max_kpts=args.n, precision=args.precision, state_dict=sd)
    np_ = eng.kmax

    # --- GEMM unit
    A = torch.randn(256, 512, device=dev); W = torch.randn(384, 512, device=dev); b = torch.randn(384, device=dev)
    Y = eng.debug_gemm(A, W, b)
    ref = A.double() @ W.double().T + b.double()
    print("gemm rel err", rel(Y.cpu().numpy(), ref.cpu().numpy()), flush=True)
    # asymmetric identity check
    A = torch.zeros(128, 128, device=dev); A[torch.arange(128), torch.arange(128)] = 1.0
    W = torch.arange(128 * 128, device=dev, dtype=torch.float32).reshape(128, 128)
    Y = eng.debug_gemm(A, W, None)
    print("gemm identity exact:", bool(torch.equal(Y, W.T.contiguous())), flush=True)

    # --- attention unit
    BS, n = 4, np_
    q = torch.randn(BS, n, 256, device=dev); k = torch.randn(BS, n, 256, device=dev); v = torch.randn(BS, n, 256, device=dev)
    nkv = torch.tensor([n, n - 37, 5, max(n // 2, 2)], dtype=torch.int32, device=dev)
    for cross in (False, True):
        out = eng.debug_attention(q, k, v, nkv, cross, 0.125)
        errs = []
        for bs in range(BS):
            kvs = bs ^ 1 if cross else bs
            m = int(nkv[kvs])
            qq = q[bs].double().reshape(n, 4, 64).transpose(0, 1) * 0.125
            kk = k[kvs, :m].double().reshape(m, 4, 64).transpose(0, 1)
            vv = v[kvs, :m].double().reshape(m, 4, 64).transpose(0, 1)
            o = torch.softmax(qq @ kk.transpose(1, 2), -1) @ vv
            o = o.transpose(0, 1).reshape(n, 256)
            errs.append(rel(out[bs].cpu().numpy(), o.cpu().numpy()))
        print(f"attention cross={cross} rel errs", errs, flush=True)

    # --- full matcher, per layer
    pairs = [make_pair(i, n_q=args.n - 8 * i, n_r=args.n - 3 * i) for i in range(args.batch)]
    inp = eng.stage_inputs(pairs)
    tsd = {k_: torch.from_numpy(v_) for k_, v_ in sd.items()}
    taps_all = []
    t0 = time.time()
    for p in pairs:
        taps = {}
        tq = torch.from_numpy
        res = lg.pose_node_match(tsd, tq(p.kp_q), tq(p.desc_q), tq(p.size_q), tq(p.angle_q),
                                 tq(p.kp_r), tq(p.desc_r), tq(p.size_r), tq(p.angle_r), taps=taps)
        taps["res"] = res
        taps_all.append(taps)
    print(f"oracle time {time.time() - t0:.2f}s", flush=True)
    T = args.batch * 2 * np_
    for nl in (1, 2, 9):
        eng.set_num_layers(nl)
        idx, score, n_match = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        torch.cuda.synchronize()
        x = eng.debug_read("x", T * 256).reshape(args.batch, 2, np_, 256)
        if nl == 1:
            desc = eng.debug_read("desc", T * 128).reshape(args.batch, 2, np_, 128)
            cos = eng.debug_read("cos", T * 32).reshape(args.batch, 2, np_, 32)
            for b, p in enumerate(pairs):
                tp = taps_all[b]
                nq, nr = len(p.kp_q), len(p.kp_r)
                print(f" pair {b} rootsift rel", rel(desc[b, 0, :nq], lg.rootsift(torch.from_numpy(p.desc_q)).numpy()),
                      "cos rel", rel(cos[b, 0, :nq], tp["enc0"][0, 0, 0, :, ::2].numpy()),
                      rel(cos[b, 1, :nr], tp["enc1"][0, 0, 0, :, ::2].numpy()), flush=True)
        for b, p in enumerate(pairs):
            tp = taps_all[b]
            nq, nr = len(p.kp_q), len(p.kp_r)
            print(f" layers={nl} pair {b}: x0 rel {rel(x[b, 0, :nq], tp[f'layer{nl - 1}_0'][0].numpy()):.3e} "
                  f"x1 rel {rel(x[b, 1, :nr], tp[f'layer{nl - 1}_1'][0].numpy()):.3e}", flush=True)
    # final: matches
    sim = eng.debug_read("sim", args.batch * np_ * np_).reshape(args.batch, np_, np_)
    nm = n_match.cpu().numpy()
    for b, p in enumerate(pairs):
        tp = taps_all[b]
        nq, nr = len(p.kp_q), len(p.kp_r)
        mq, mr, sc, oidx = tp["res"]
        gi = idx[b, : nm[b]].cpu().numpy()
        same = gi.shape == tuple(oidx.shape) and bool((gi == oidx.numpy()).all())
        print(f" pair {b}: sim rel {rel(sim[b, :nq, :nr], tp['sim'][0].numpy()):.3e}  K gpu {nm[b]} oracle {len(oidx)} identical idx: {same}",
              flush=True)
        if same and len(oidx):
            print("   score max abs diff", float(np.abs(score[b, : nm[b]].cpu().numpy() - sc.numpy()[:, 0]).max()), flush=True)

    # --- full estimate
    out = eng.estimate(inp, K_MATRIX)
    torch.cuda.synchronize()
    Rg, tg = out["R"].cpu().numpy(), out["t"].cpu().numpy()
    print("ok", out["ok"].cpu().numpy(), "n_match", out["n_match"].cpu().numpy(), "n_inl", out["n_inliers"].cpu().numpy(), flush=True)
    for b, p in enumerate(pairs):
        mq, mr, sc, oidx = taps_all[b]["res"]
        t0 = time.time()
        o = pr.compute_pose(K_MATRIX.reshape(-1), mq.numpy(), mr.numpy(), p.dem)
        dt = time.time() - t0
        Ro, to = o
        print(f" pair {b}: |dR| {np.linalg.norm(Rg[b] - Ro):.3e} |dt|/|t| {np.linalg.norm(tg[b] - to) / np.linalg.norm(to):.3e} "
              f"(vs gt: dR {np.linalg.norm(Rg[b] - p.R_gt):.2e} dt {np.linalg.norm(tg[b] - p.t_gt) / np.linalg.norm(p.t_gt):.2e}) oracle pnp {dt * 1e3:.1f} ms",
              flush=True)
    # timing
    eng.set_stage_timing(True)
    for _ in range(2):
        eng.estimate(inp, K_MATRIX, out=out)
    torch.cuda.synchronize()
    print("stage ms", eng.stage_ms(), flush=True)
    t0 = time.time()
    eng.set_stage_timing(False)
    for _ in range(5):
        eng.estimate(inp, K_MATRIX, out=out)
    torch.cuda.synchronize()
    print(f"estimate: {(time.time() - t0) / 5 * 1e3:.2f} ms per batch of {args.batch}", flush=True)


if __name__ == "__main__":
    main()
