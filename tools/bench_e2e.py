#!/usr/bin/env python
"""Frames -> pose from pixels on one MI355X: batched SIFT on B camera frames + B map tiles (gn_sift_detect_and_compute_batch),
then the matcher + PnP on the B pairs (gn_estimate), everything resident in HBM.  One JSON line; timing tool for DESIGN.md 7 --
the headline metric stays bench.py's (matcher + PnP on pre-extracted features, as BASELINE.json defines it).
    python tools/bench_e2e.py [--batch 32] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gisnav_amd import _lib  # noqa: E402
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.sift import SIFT  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
from test_sift import blob_image  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32, help="frame/tile pairs per step")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kpts", type=int, default=1024)
    ap.add_argument("--size", type=int, nargs=2, default=[480, 640])
    ap.add_argument("--no-active", dest="active", action="store_false", help="always pad the matcher to --kpts (skip gn_set_active_kpts)")
    args = ap.parse_args()
    B, (H, W) = args.batch, args.size
    eng = PoseEngine(0, max_batch=B, max_kpts=args.kpts, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
    sift = SIFT(engine=eng, max_keypoints=args.kpts)
    # a camera frame and the map tile it looks at: two crops of one scene, 11 x 8 pixels apart
    frames, tiles = [], []
    for b in range(B):
        big = blob_image(100 + b, H + 16, W + 16, n=1500)
        tiles.append(big[:H, :W]); frames.append(big[8:H + 8, 11:W + 11])
    imgs = torch.as_tensor(np.stack(frames + tiles), device=eng.device)
    dem = torch.zeros((B, H, W), dtype=torch.uint8, device=eng.device)
    out = eng.alloc_outputs(B)

    def step():
        kpt, _, _, desc, n = sift.detect_and_compute_batch_device(imgs)
        nd = torch.as_tensor(n, device=eng.device)
        if args.active:
            eng.set_active_kpts(int(n.max()))                   # run the matcher at round_up(max keypoints, 128) slots instead of --kpts
        inp = dict(desc_q=desc[:B], kpt_q=kpt[:B], n_q=nd[:B], desc_r=desc[B:], kpt_r=kpt[B:], n_r=nd[B:], dem=dem, kpt_format=_lib.GN_KPT_XYSA)
        eng.estimate(inp, K_MATRIX, out=out)
        return n

    for _ in range(args.warmup):
        n = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    t1 = time.perf_counter()
    for _ in range(args.steps):
        sift.detect_and_compute_batch_device(imgs)
    torch.cuda.synchronize()
    ms_sift = (time.perf_counter() - t1) / args.steps * 1e3
    print(json.dumps({"metric": "frame/tile pairs per second from pixels (SIFT on both images + LightGlue matcher + PnP)", "value": round(B / ms * 1e3, 1),
                      "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
                      "ms_sift_per_step": round(ms_sift, 3), "higher_is_better": True, "data": "synthetic", "dtype": "f32 SIFT; f16x2 / bf16-attention matcher",
                      "config": {"workload": f"{B} pairs of {H}x{W} u8 images, {int(n.mean())} keypoints per image on average (max {int(n.max())}), "
                                             f"matcher padded to {eng.set_active_kpts(int(n.max())) if args.active else args.kpts}; random-init matcher weights (timing only)"}}), flush=True)


if __name__ == "__main__":
    main()
