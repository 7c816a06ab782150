#!/usr/bin/env python
"""Benchmark of the MI355X PoseNode hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--kpts 1024] [--precision bf16_attn]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the whole hot path -- RootSIFT, 9-layer LightGlue("sift") matcher, dual-softmax
mutual-NN match head, matched-point gather + DEM lift, PnP-RANSAC + refinement -- over one batch of
`--batch` synthetic 640x480 frame<->tile pairs (1024 SIFT keypoints per side) that are already resident
in HBM.  Pairs are independent, so N ranks each process their own contiguous shard of N*batch pairs
(weak scaling, no data-path collective); value = pairs all ranks processed / max-over-ranks time.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gisnav_amd import dist as gdist  # noqa: E402
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense (the 2:1-sparsity figure is never used)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E, ~8 TB/s
GEMM_LAUNCHES_PER_STEP = 1 + 9 * 8 + 2  # input_proj + 9 x (4 proj + 2 x 2 ffn) + final_proj + sim
ATTN_LAUNCHES_PER_STEP = 9 * 2           # one self + one (two-sided) cross attention launch per layer


def cpu_baseline(state_dict, kpts: int, seconds_budget: float = 20.0):
    """The oracle (restated reference: torch-CPU LightGlue-sift + numpy solvePnPRansac) timed on this
    box's host cores, on a bounded sample of the same workload."""
    from oracle import lightglue_sift as lg
    from oracle import pnp_ransac as pr
    # host threads actually used: the CPUs this process may run on, capped at 32 (beyond that the
    # 1024x256-sized CPU GEMMs of one pair only lose time to synchronisation)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 32)))
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state_dict.items()}
    tq = torch.from_numpy
    times, poses = [], 0
    i = 0
    t_start = time.perf_counter()
    while True:
        p = make_pair(10_000 + i, n_q=kpts, n_r=kpts)
        t0 = time.perf_counter()
        mq, mr, _, _ = lg.pose_node_match(sd, tq(p.kp_q), tq(p.desc_q), tq(p.size_q), tq(p.angle_q),
                                          tq(p.kp_r), tq(p.desc_r), tq(p.size_r), tq(p.angle_r))
        if len(mq) >= 15:
            poses += pr.compute_pose(K_MATRIX.reshape(-1), mq.numpy(), mr.numpy(), p.dem) is not None
        dt = time.perf_counter() - t0
        if i >= 1:  # first pair is warm-up
            times.append(dt)
        i += 1
        if (len(times) >= 3 and time.perf_counter() - t_start > seconds_budget) or time.perf_counter() - t_start > 6 * seconds_budget:
            break
    med = float(np.median(times))
    return {"value": round(1.0 / med, 4), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} synthetic 640x480 pairs ({kpts} kpts/side) after 1 warm-up pair, median; "
                      f"torch-CPU fp32 LightGlue-sift restatement + numpy solvePnPRansac restatement (oracle/); "
                      f"cpu={platform.processor() or platform.machine()}"}


def measured_gemm_traffic(precision: str):
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (bench.py cannot run under
    the profiler itself); None if the summary is absent."""
    name = {"f32x3_bf16_attn": "r01_pmc_hbm_traffic_f32x3.json", "f16x2_bf16_attn": "r01_pmc_hbm_traffic_f16x2.json"}.get(
        precision, "r01_pmc_hbm_traffic.json")
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            d = json.load(f)
        return int(d["k_gemm_f32_all_variants"]["hbm_bytes_per_launch_corrected"]), os.path.relpath(path, ROOT)
    except (OSError, KeyError, ValueError):
        return None, None


def measured_mfma_busy(precision: str):
    """Matrix-pipe busy fraction of the dominant GEMM kernel from the committed PMC pass (call-weighted), or None."""
    if precision != "f16x2_bf16_attn":
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_mfma_busy_f16x2.json")) as f:
            ks = json.load(f)["kernels"]
        rows = [(v["dispatches"], v["mfma_busy"]) for k, v in ks.items() if k.startswith("k_gemm")]
        return round(sum(n * b for n, b in rows) / max(sum(n for n, _ in rows), 1), 4)
    except (OSError, KeyError, ValueError):
        return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="pairs per GPU per step")
    ap.add_argument("--kpts", type=int, default=1024)
    ap.add_argument("--precision", default="f16x2_bf16_attn", choices=["f32", "bf16_attn", "f32x3_bf16_attn", "f16x2_bf16_attn"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true", help="run the PnP stage of step n on a second stream beside the matcher of step n+1 "
                                                           "(gn_set_overlap; measured gain < 1 %: the 256-VGPR GEMM waves leave no room for co-resident PnP waves)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only for single-GPU dry runs)")
    ap.add_argument("--debug-variant", action="append", default=[], metavar="WHICH:VALUE", help="developer knob: gn_debug_set_variant(which, value) before the run (timing experiments)")
    ap.add_argument("--share-gpu", action="store_true", help="dry-run aid: every rank uses cuda:0 (with --backend gloo)")
    args = ap.parse_args()

    rank, local_rank, world = gdist.init(args.backend)
    if args.share_gpu:
        local_rank = 0
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # weights: seeded synthetic on rank 0, broadcast once over RCCL (no checkpoint is available offline)
    sd = synthetic_state_dict(0)
    if world > 1:
        if rank != 0:
            sd = {k: np.zeros_like(v) for k, v in sd.items()}
        sd = gdist.broadcast_state_dict(sd, dev, src=0)
    eng = PoseEngine(local_rank, max_batch=args.batch, max_kpts=args.kpts, precision=args.precision, state_dict=sd)

    for kv in args.debug_variant:
        which, value = (int(v) for v in kv.split(":"))
        eng.lib.gn_debug_set_variant(eng.ctx, which, value)

    # this rank's contiguous shard of the global batch, staged into HBM before the timed region
    shard = gdist.shard_range(args.batch * world, rank, world)
    pairs = [make_pair(i, n_q=args.kpts, n_r=args.kpts) for i in shard]
    inp = eng.stage_inputs(pairs)
    out = eng.alloc_outputs(len(pairs))
    torch.cuda.synchronize()

    if args.overlap:
        eng.set_overlap(True)      # PnP of step n beside the matcher of step n+1 (HIP streams); flushed before the closing sync
    for _ in range(args.warmup):
        eng.estimate(inp, K_MATRIX, out=out)
    eng.flush()
    eng.set_kernel_timing((GEMM_LAUNCHES_PER_STEP + ATTN_LAUNCHES_PER_STEP) * args.steps)
    gdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.estimate(inp, K_MATRIX, out=out)
    eng.flush()
    torch.cuda.synchronize()
    gdist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = gdist.max_over_ranks(elapsed, dev)
    kstats = eng.kernel_stats(0)
    astats = eng.kernel_stats(1)
    eng.set_kernel_timing(0)

    n_ok = float(out["ok"].sum().item())
    n_ok_all = gdist.sum_over_ranks(n_ok, dev)
    n_match_mean = float(out["n_match"].float().mean().item())
    rec = gdist.gather_records(gdist.pack_records(shard.start, out))  # fixed-size result records, 128 B/pair

    if rank == 0:
        total_pairs = args.batch * world * args.steps
        pairs_per_s = total_pairs / elapsed
        ach = kstats["flops"] / (kstats["ms"] * 1e-3) / 1e12 if kstats["ms"] > 0 else 0.0
        x3 = args.precision == "f32x3_bf16_attn"
        h2 = args.precision == "f16x2_bf16_attn"
        mult = 6 if x3 else 3 if h2 else 1      # matrix-pipe flops issued per algorithmic flop
        peak = PEAK_BF16_MFMA_TFLOPS / mult if (x3 or h2) else PEAK_F32_MFMA_TFLOPS
        traffic, traffic_src = measured_gemm_traffic(args.precision) if (args.batch == 32 and args.kpts == 1024) else (None, None)
        # which roof binds these launches: arithmetic intensity (algorithmic flops per compulsory HBM byte) x 8 TB/s against the matrix pipe
        alg_bytes = kstats["bytes"] / max(kstats["launches"], 1)
        alg_flops = kstats["flops"] / max(kstats["launches"], 1)
        intensity = alg_flops / max(alg_bytes, 1.0)
        hbm_roof_tflops = intensity * PEAK_HBM_GBS * 1e9 / 1e12
        gbs = kstats["bytes"] / (kstats["ms"] * 1e-3) / 1e9 if kstats["ms"] > 0 else 0.0
        if hbm_roof_tflops < peak:
            gemm_roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                         "note": (f"the GEMMs of this path are skinny (K, N in 256..768): {intensity:.0f} algorithmic flops per compulsory HBM byte puts their "
                                  f"HBM roof at {hbm_roof_tflops:.0f} TFLOP/s, below the {peak:.0f} TFLOP/s matrix-pipe roof, so HBM is the binding roof; "
                                  "achieved = algorithmic bytes (A, W, every output array, residual rows, rotary tables -- each once) / HIP-event time; the 18 ffn.0 launches "
                                  "of a step also carry LayerNorm + GELU in their epilogue (k_gemm_p2ln), work that is not counted as bytes or flops"),
                         "algorithmic_mb_per_launch": round(alg_bytes / 1e6, 1), "flops_per_byte": round(intensity, 1),
                         "hbm_roof_tflops": round(hbm_roof_tflops, 1)}
        else:
            gemm_roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "note": f"matrix-pipe roof {peak:.0f} TFLOP/s is below the HBM roof ({intensity:.0f} flops per byte x 8 TB/s = {hbm_roof_tflops:.0f} TFLOP/s)",
                         "algorithmic_mb_per_launch": round(alg_bytes / 1e6, 1), "flops_per_byte": round(intensity, 1),
                         "hbm_roof_tflops": round(hbm_roof_tflops, 1), "achieved_gbs": round(gbs, 1)}
        line = {
            "metric": "matched frame-pairs/sec + PnP poses/sec, 640x480 cam-vs-tile",
            "value": round(pairs_per_s, 2),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32",
                      "bf16_attn": "f32 projections/FFN/match-head + bf16 MFMA attention (f32 accumulate)",
                      "f32x3_bf16_attn": "f32-accurate projections/FFN/match-head (each f32 operand split exactly into 3 bf16 terms, "
                                         "6 bf16 MFMA partial products, f32 accumulate) + bf16 MFMA attention (f32 accumulate)",
                      "f16x2_bf16_attn": "f32-accurate projections/FFN/match-head (each f32 operand split into 2 fp16 terms = 22 "
                                         "significant bits, 3 fp16 MFMA partial products, f32 accumulate; error vs fp64 <= the f32 "
                                         "MFMA path's) + bf16 MFMA attention (f32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE {'configs[1] (batch-1' if args.batch == 1 else 'configs[2]/[3] (batch-' + str(args.batch)} per GPU): "
                            f"640x480 pairs, {args.kpts} SIFT kpts/side, "
                            f"LightGlue-sift 9 layers + dual-softmax head + GPU PnP-RANSAC(10 it, 8 px) + LM refine",
                "pairs_per_gpu_per_step": args.batch,
                "global_pairs_per_step": args.batch * world,
                "keypoints_per_side": args.kpts,
                "precision": args.precision,
                "parallelism": f"pair-sharded x{world} (no data-path collective)",
                "pose_stage_overlap": bool(args.overlap),
                "weights": "seeded synthetic, kornia sift_lightglue state-dict layout",
            },
            "poses_per_s": round(n_ok_all * args.steps / elapsed, 2),
            "poses_ok_per_step": int(n_ok_all),
            "mean_matches_per_pair": round(n_match_mean, 1),
            "result_records_gathered": int(rec.shape[0]),
            "roofline": {
                "kernel": ("k_gemm_f32x3 (projection/FFN/similarity GEMM: 3 x bf16 split, 6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 block)"
                           if x3 else "k_gemm_f16x2 (projection/FFN/similarity GEMM: 2 x fp16 split, 3 x v_mfma_f32_32x32x16_f16 per 32x32x16 block)"
                           if h2 else "k_gemm_f32_v3 (projection/FFN/similarity GEMM on v_mfma_f32_32x32x2_f32)"),
                **gemm_roof,
                "matrix_pipe": {"achieved_tflops": round(ach, 2), "peak_tflops": round(peak, 1), "frac": round(ach / peak, 4),
                                "note": (f"algorithmic 2*M*N*K flops / HIP-event time; the kernel issues {mult} 16-bit MFMA flops per algorithmic "
                                         f"flop, so the matrix-pipe ceiling is 2500 TF dense / {mult}"
                                         if (x3 or h2) else "algorithmic 2*M*N*K flops / HIP-event time vs the f32 MFMA peak")},
                "executed_mfma_tflops": round(ach * mult, 1),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "mfma_busy_pmc": measured_mfma_busy(args.precision) if (args.batch == 32 and args.kpts == 1024) else None,
                "launches_timed": int(kstats["launches"]),
                "avg_launch_us": round(kstats["ms"] * 1e3 / max(kstats["launches"], 1), 2),
                "algorithmic_gflop_per_launch": round(kstats["flops"] / max(kstats["launches"], 1) / 1e9, 3),
            },
        }
        a_ach = astats["flops"] / (astats["ms"] * 1e-3) / 1e12 if astats["ms"] > 0 else 0.0
        a_peak = PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_BF16_MFMA_TFLOPS
        line["roofline_attention"] = {   # second-largest kernel; the QK^T / PV contractions north_star singles out
            "kernel": "k_attn_f32" if args.precision == "f32" else "k_attn_bf16_v5",
            "bound": "mfma", "achieved": round(a_ach, 2), "peak": a_peak, "unit": "TFLOP/s", "frac": round(a_ach / a_peak, 4),
            "launches_timed": int(astats["launches"]),
            "avg_launch_us": round(astats["ms"] * 1e3 / max(astats["launches"], 1), 2),
            "algorithmic_gflop_per_launch": round(astats["flops"] / max(astats["launches"], 1) / 1e9, 3),
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(sd, args.kpts)
        print(json.dumps(line), flush=True)
    gdist.barrier()


if __name__ == "__main__":
    main()
