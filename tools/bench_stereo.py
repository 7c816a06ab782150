#!/usr/bin/env python
"""Measurement of the StereoNode reference-raster preparation on one MI355X next to the oracle
on the host.  One JSON line.   python tools/bench_stereo.py [--steps 200] [--size 1024 1365] [--crop 480 640]"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.stereo import stereo_reference  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, nargs=2, default=[1024, 1365], help="orthoimage tile H W")
    ap.add_argument("--crop", type=int, nargs=2, default=[480, 640], help="camera resolution H W")
    args = ap.parse_args()
    H, W = args.size
    rng = np.random.default_rng(0)
    bgr_h = rng.integers(0, 256, (H, W, 3)).astype(np.uint8); dem_h = rng.integers(0, 64, (H, W)).astype(np.uint8)
    eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f32")
    bgr = torch.as_tensor(bgr_h, device=eng.device); dem = torch.as_tensor(dem_h, device=eng.device)
    crop = (args.crop[0], args.crop[1])
    for i in range(args.warmup):
        stereo_reference(eng, bgr, dem, 5.0 * i, crop)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for i in range(args.steps):
        stereo_reference(eng, bgr, dem, 5.0 * (i % 72), crop)
    e.record(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = s.elapsed_time(e) / args.steps
    from oracle import stereo_warp as sw
    times = []
    for i in range(5):
        t1 = time.perf_counter(); sw.stereo_reference(bgr_h, dem_h, 5.0 * i, crop); times.append(time.perf_counter() - t1)
    alg_bytes = crop[0] * crop[1] * (4 + 2)      # the source window (BGR + DEM bytes under the crop) in, two u8 rasters out
    line = {"metric": "StereoNode reference rasters/sec (BGR2GRAY + DEM stack + rotate + centre crop)", "value": round(args.steps / wall, 1),
            "unit": "rasters/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 fixed point (bit-exact)", "data": "synthetic",
            "config": {"workload": f"{H}x{W} BGR orthoimage + u8 DEM -> {crop[0]}x{crop[1]} reference + DEM, one raster per call"},
            "roofline": {"kernel": "k_rotate_crop<fused gray>", "bound": "hbm", "achieved": round(alg_bytes / (ms * 1e-3) / 1e9, 2), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(alg_bytes / (ms * 1e-3) / 1e9 / 8000.0, 5), "traffic": None,
                         "note": "one 0.3-Mpixel launch per call: launch-latency bound, the node prepares one raster per 5-degree heading change"},
            "cpu_baseline": {"value": round(1.0 / float(np.median(times)), 2), "unit": "rasters/s", "cores": 1, "kind": "port",
                             "sample": f"5 calls, median; numpy restatement of cvtColor + warpAffine + crop (oracle/); cpu={platform.processor() or platform.machine()}"}}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
