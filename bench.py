#!/usr/bin/env python
"""Benchmark of the MI355X PoseNode hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--kpts 1024] [--precision f16x2_bf16_attn]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the whole hot path -- RootSIFT, 9-layer LightGlue("sift") matcher, dual-softmax mutual-NN match head,
matched-point gather + DEM lift, PnP-RANSAC + refinement -- over one batch of `--batch` synthetic 640x480 frame<->tile pairs
(1024 SIFT keypoints per side) that are already resident in HBM.  Pairs are independent, so N ranks each process their own
contiguous shard of N*batch pairs (weak scaling, no data-path collective); value = pairs all ranks processed / max-over-ranks time.
Rank 0 prints ONE JSON line.

What the line reports, and where each number comes from:
  value / ms_per_step   host clock around K steps, barrier + synchronize on both sides, max over ranks
  end_to_end            value x 80.7 GFLOP (algorithmic work per pair, formula of SURVEY.md:405-408) against the 2.5 PF dense 16-bit MFMA peak
  roofline              the DOMINANT kernel (largest share of the HIP-event time measured in this run, on the launch stream), by the
                        name rocprofv3 prints for it: algorithmic flops per launch / average launch duration, against the same peak
  kernels               the same for every timed kernel (joins row by row with profiles/*kernel_stats*.csv)
  traffic               HBM bytes per step from rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md), collected by
                        re-running this script under the profiler (N = 1 only; --no-traffic skips it), next to the compulsory bytes
  extra_configs         further BASELINE configurations timed in the same process (batch-1 f32 = configs[1] as the survey reads it;
                        batch-32 with exact-f32 MFMA GEMMs + bf16 attention = configs[2] without the split-fp16 GEMMs)
  cpu_baseline          the CPU restatement of the reference on this box's host cores, bounded sample
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import platform
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gisnav_amd import dist as gdist  # noqa: E402
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 / fp16 dense (the 2:1-sparsity figure is never used)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E, ~8 TB/s
MAX_TIMED_LAUNCHES_PER_STEP = 320


def gflop_per_pair(n: int, m: int, d: int = 256, layers: int = 9, d_in: int = 128):
    """Algorithmic work of one pair (2 x MACs), SURVEY.md:405-408; 80.7 GFLOP at n = m = 1024.  Attention part separately."""
    macs = (n + m) * d_in * d + layers * ((n + m) * (4 * d * d + 6 * d * d) + (n + m) * (3 * d * d + 6 * d * d) + 2 * d * (n * n + m * m) + 3 * d * n * m) \
        + (n + m) * d * d + d * n * m
    attn = layers * (2 * d * (n * n + m * m) + 3 * d * n * m) + d * n * m
    return 2.0 * macs / 1e9, 2.0 * attn / 1e9


def compulsory_mb_per_pair(n: int, m: int) -> float:
    """Compulsory HBM I/O of one pair (SURVEY.md:409-410): descriptors + keypoints in, matches / pose out; weights amortised over the batch."""
    return ((n + m) * 128 * 4 + (n + m) * 4 * 4 + min(n, m) * (16 + 4) + 128) / 1e6


def timed_steps(eng, inps, out, steps, warmup, dev, kernel_timing=False):
    """W warm-up steps, then K steps bracketed by barrier + synchronize; returns (elapsed seconds max over ranks, this rank's seconds).
    `inps` is a list of DISTINCT staged batches (all resident in HBM): step i works on batch i mod len(inps)."""
    if isinstance(inps, dict):
        inps = [inps]
    outs = out if isinstance(out, list) else [out]      # (deferred certificate: call n's outputs stay untouched until call n + 1 has returned)
    for i in range(warmup):
        eng.estimate(inps[i % len(inps)], K_MATRIX, out=outs[i % len(outs)])
    eng.flush()
    if kernel_timing:
        eng.set_kernel_timing(MAX_TIMED_LAUNCHES_PER_STEP * steps)
    gdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.estimate(inps[i % len(inps)], K_MATRIX, out=outs[i % len(outs)])
    eng.flush()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    gdist.barrier()
    elapsed = time.perf_counter() - t0
    return gdist.max_over_ranks(elapsed, dev), mine


def kernel_rows(table, steps, precision):
    """Per-kernel roofline rows from the HIP-event table of the run."""
    total_ms = sum(r["ms"] for r in table) or 1.0
    rows = []
    for r in sorted(table, key=lambda r: -r["ms"]):
        n = max(r["launches"], 1)
        attn = r["name"].startswith("k_attn")
        f32_pipe = r["name"].startswith("k_gemm_f32_v3") or r["name"].startswith("k_attn_f32")
        peak = PEAK_F32_MFMA_TFLOPS if f32_pipe else PEAK_16BIT_MFMA_TFLOPS
        two = r["name"].startswith("k_qkv") and r["name"].rstrip(">").endswith(", 2")         # attention input projections: two partial products (gn_qkv.hip)
        mult = 1 if (attn or f32_pipe) else 6 if r["name"].startswith("k_gemm_f32x3") else 2 if two else 3   # matrix-pipe flops issued per algorithmic flop
        if r["name"].startswith("k_ffn128"):      # k_ffn128<ABL, COMP, LOOP, QKV, PROD>: block tail on PROD partial products (+ the next block's projection, 2 products, in the same launch)
            prm = r["name"][r["name"].index("<") + 1:].rstrip(">").split(", ")
            qkv, prod = (int(prm[3]), int(prm[4])) if len(prm) >= 5 else (0, 3)
            tail = 2.0 * (512 * 512 + 256 * 512)
            proj = 2.0 * 256 * (768 if qkv == 1 else 512 if qkv == 2 else 0)
            mult = round((prod * tail + 2 * proj) / (tail + proj), 4)
        tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
        rows.append({"name": r["name"], "launches_per_step": round(n / steps, 2), "avg_launch_us": round(r["ms"] * 1e3 / n, 2),
                     "share_of_timed_kernel_time": round(r["ms"] / total_ms, 4),
                     "algorithmic_gflop_per_launch": round(r["flops"] / n / 1e9, 3), "achieved_tflops": round(tf, 1),
                     "peak_tflops": peak, "frac": round(tf / peak, 4),
                     "mfma_flops_issued_per_algorithmic_flop": mult, "frac_of_issue_ceiling": round(tf * mult / peak, 4),
                     "algorithmic_mb_per_launch": round(r["bytes"] / n / 1e6, 1)})
    return rows


def measure_traffic(args, steps_prof: int = 2, warmup_prof: int = 1, certify_eps: str = ""):
    """HBM bytes per kernel from two rocprofv3 PMC passes over this same script (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass),
    --kernel-trace only.  FETCH_SIZE counts 64 B per 128-B request on gfx950 for wide coalesced reads: doubled (MI355X_MICROARCH.md)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    work = tempfile.mkdtemp(prefix="gn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(work, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", str(steps_prof), "--warmup", str(warmup_prof), "--batch", str(args.batch), "--kpts", str(args.kpts), "--precision", args.precision,
                   "--no-cpu-baseline", "--no-traffic", "--no-extras", "--no-stream", "--no-rccl-check", "--substreams", "1"]     # one pass, full-batch launches: the configuration of the kernel table
            for kv in args.debug_variant:
                cmd += ["--debug-variant", kv]
            cmd += ["--certify-eps", certify_eps] if certify_eps else ["--no-certify"] if args.no_certify else []
            cmd += ["--ffn-products", str(args.ffn_products)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        k = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                        if "gn::" not in k:
                            continue
                        k = k.replace("void ", "").replace("gn::(anonymous namespace)::", "").replace("gn::", "").split("(")[0]
                        v = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
                        e = per.setdefault(k, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                        e[counter][0] += v
                        e[counter][1] += 1
    except (subprocess.TimeoutExpired, OSError) as exc:
        return None, f"rocprofv3 pass failed: {exc}"
    finally:
        shutil.rmtree(work, ignore_errors=True)
    n_steps = steps_prof + warmup_prof
    kernels, total = {}, 0.0
    for k, e in per.items():
        f_kb, f_n = e["FETCH_SIZE"]
        w_kb, w_n = e["WRITE_SIZE"]
        n = max(f_n, w_n, 1)
        by = (2.0 * f_kb + w_kb) * 1024.0           # counters are in KB; corrected fetch = 2 x FETCH_SIZE
        kernels[k] = {"dispatches_per_step": round(n / n_steps, 2), "hbm_mb_per_launch": round(by / n / 1e6, 2)}
        total += by / n_steps
    return {"hbm_mb_per_step": round(total / 1e6, 1), "kernels": kernels,
            "method": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over {n_steps} steps of this script; bytes = 2 x FETCH_SIZE + WRITE_SIZE"}, None


CERT_SAFETY = 4.0


def certify_on(eng, kpts, batch, precision, certify=True, eps=""):
    """The margin certificate for a fast-mode engine (gn_set_certify(2)): eps is CALIBRATED for these weights on one batch of pairs that are not part of any
    timed batch (max |P_mode - P_f32| over the deciding entries x CERT_SAFETY), then every estimate() call synchronises once, reads its per-pair
    flags and re-runs the flagged pairs on the exact-f32 kernels -- inside the timed region.  Returns the calibration record (None: f32 / off)."""
    if precision == "f32" or not certify:
        return None
    if eps:      # profiling pass: the eps a full run calibrated -- "E", or "E3,E2,LEVEL" for the automatic block-tail level (and the level that run settled on)
        v = [float(x) for x in str(eps).split(",")]
        eng.set_certify("rerun", eps=v[0])
        if len(v) == 3:
            eng.set_ffn_level_eps(v[1], v[0], int(v[2]))
        return {"eps": v[0], "measured": None, "safety": None}
    # (as many pairs as a timed call holds: the kernel family, and with it the arithmetic whose error is measured, follows the grid size)
    cal_pairs = [make_pair(900_000 + i, n_q=kpts, n_r=kpts) for i in range(batch)]
    cal = eng.calibrate_certify(eng.stage_inputs(cal_pairs), safety=CERT_SAFETY)
    eng.set_certify("rerun")
    if "eps_two_products" in cal:
        # automatic block-tail level: it starts on three products and moves after a window of 64 certified pairs; let it settle on pairs outside every timed batch
        inp = eng.stage_inputs(cal_pairs)
        out = eng.alloc_outputs(batch)
        for _ in range(max(2, -(-128 // batch))):
            eng.estimate(inp, K_MATRIX, out=out)
        torch.cuda.synchronize()
        eng.certify_stats(reset=True)
    return cal


def certificate_block(eng, cal):
    if cal is None:
        return None
    st = eng.certify_stats()
    lv = eng.ffn_level()
    auto = "eps_two_products" in cal
    return {"block_tail_level": None if not auto else {
                "setting": "automatic (gn_set_ffn_products(0)): two partial products while that flags no more pairs than three would (1 in 64 tolerated), else three",
                "level_now": lv["level"], "certified_calls_on_two_products": lv["calls_two_products"], "certified_calls_on_three_products": lv["calls_three_products"],
                "switches": lv["switches"], "eps_two_products": cal["eps_two_products"], "eps_three_products": cal["eps_three_products"]},
            "profile_eps_arg": ("%r,%r,%d" % (cal["eps_three_products"], cal["eps_two_products"], lv["level"])) if auto else repr(cal["eps"]),
            "mode": "margin guard + exact-f32 re-run of the flagged pairs (gn_set_certify(2)); one stream synchronisation per call, inside the timed region",
            "resolution": cal.get("resolution", "inside every call"), "eps_by_calibration_shape": cal.get("eps_by_calibration_shape"),
            "eps": cal["eps"], "eps_measured_max_dP": cal["measured"], "safety_factor": cal["safety"],
            "calls": st["calls"], "pairs": st["pairs"], "pairs_flagged_margin": st["flagged_margin"], "pairs_flagged_fp16_range": st["flagged_fp16_range"],
            "pairs_rerun_in_f32": st["rerun_pairs"], "rerun_fraction": round(st["rerun_fraction"], 6), "rerun_pairs_marginal_even_in_f32": st["f32_marginal_pairs"]}


def run_extra(local_rank, sd, name, batch, kpts, precision, steps, warmup, dev, certify=True, substreams=1, ffn_products=3, filter_threshold=0.5):
    """One further configuration, timed like the main one (no per-kernel events, no collectives needed: every rank runs it)."""
    eng = PoseEngine(local_rank, max_batch=batch, max_kpts=kpts, precision=precision, state_dict=sd, filter_threshold=filter_threshold)
    if precision.startswith("f16x2"):
        eng.set_ffn_products(ffn_products)
    cal = certify_on(eng, kpts, batch, precision, certify)
    pairs = [make_pair(i, n_q=kpts, n_r=kpts) for i in range(batch)]
    inp = eng.stage_inputs(pairs)
    out = eng.alloc_outputs(batch)
    outs = out
    if substreams > 1:
        eng.set_substreams(substreams)
        if cal is not None:      # as in the headline: the certificate resolved one call later, two alternating output sets
            eng.set_certify("deferred")
            cal["resolution"] = "deferred by one call (gn_set_certify(3)); two alternating output sets"
            outs = [out, eng.alloc_outputs(batch)]
    torch.cuda.synchronize()
    elapsed, _ = timed_steps(eng, inp, outs, steps, warmup, dev)
    ok = int(out["ok"].sum().item())
    cert = certificate_block(eng, cal)
    del eng
    torch.cuda.empty_cache()
    g, _ = gflop_per_pair(kpts, kpts)
    pps = batch * steps / elapsed
    peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_16BIT_MFMA_TFLOPS
    return {"config": name, "batch": batch, "keypoints_per_side": kpts, "precision": precision, "steps": steps, "warmup": warmup,
            "value": round(pps, 2), "unit": "pairs/s (this rank's GPU)", "ms_per_step": round(elapsed / steps * 1e3, 4), "poses_ok_per_step": ok,
            "end_to_end_tflops": round(pps * g / 1e3, 1), "end_to_end_frac_of_peak": round(pps * g / 1e3 / peak, 4), "peak_tflops": peak,
            "block_tail_partial_products": (ffn_products or "automatic") if precision.startswith("f16x2") else None,
            "index_exact": "exact-f32 arithmetic" if precision == "f32" else "certified (margin guard, f32 re-run)" if cert else "tolerance mode", "certificate": cert}


def run_extra_ragged(local_rank, sd, batch, precision, steps, warmup, dev, lo=400, hi=2500, bucket=8, certify=True):
    """A ragged batch: `cv2.SIFT_create()` is unbounded (pose_node.py:122), so N and M differ per message.  `batch` pairs with N, M ~ U(lo, hi)
    per side: (a) ONE gn_estimate call padded to the batch maximum, (b) the length-bucketing scheduler (PoseEngine.estimate_bucketed: groups of `bucket`
    pairs sorted by length, each padded to its own maximum).  Same poses either way."""
    rs = np.random.default_rng(77)
    nq = rs.integers(lo, hi + 1, batch); nr = rs.integers(lo, hi + 1, batch)
    kmax = ((int(max(nq.max(), nr.max())) + 127) // 128) * 128
    eng = PoseEngine(local_rank, max_batch=batch, max_kpts=kmax, precision=precision, state_dict=sd)
    cal = certify_on(eng, 1024, batch, precision, certify)
    if cal is not None:
        # eps belongs to the grid it was measured on (the kernel family follows the grid): a ragged stream runs several -- the padded call and the buckets --
        # so it is measured on each of them and the LARGEST is stated for all
        by_shape = {f"{batch}x1024": cal["eps"]}
        for b, k in ((batch, kmax), (16, 1536), (8, 1536)):
            c2 = eng.calibrate_certify(eng.stage_inputs([make_pair(910_000 + i, n_q=k, n_r=k) for i in range(b)]), safety=CERT_SAFETY)
            by_shape[f"{b}x{k}"] = c2["eps"]
            if c2["eps"] > cal["eps"]:
                cal["eps"], cal["measured"] = c2["eps"], c2["measured"]
        cal["eps_by_calibration_shape"] = {k: round(v, 7) for k, v in by_shape.items()}
        eng.set_certify("rerun", eps=cal["eps"])
    pairs = [make_pair(500 + i, n_q=int(nq[i]), n_r=int(nr[i])) for i in range(batch)]
    inp = eng.stage_inputs(pairs)
    n_q = np.array([len(p.kp_q) for p in pairs]); n_r = np.array([len(p.kp_r) for p in pairs])
    out_a, out_b = eng.alloc_outputs(batch), eng.alloc_outputs(batch)

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps

    # (no set_ragged hint: the library picks the block tail's form from the padding its previous call saw -- the warm-up calls -- DESIGN 12)
    eng.set_active_kpts(int(max(n_q.max(), n_r.max())))
    t_one = timed(lambda: eng.estimate(inp, K_MATRIX, out=out_a))
    eng.set_active_kpts(eng.kmax)
    stats, t_bkt, tried = {}, None, {}
    for bk in (16, 8):      # groups of 16 keep every launch at the size the kernels are tuned for; groups of 8 pad less but leave CUs idle
        st = {}
        def bucketed():
            nonlocal st
            _, st = eng.estimate_bucketed(inp, K_MATRIX, n_q, n_r, bucket_pairs=bk, out=out_b)
        tb = timed(bucketed)
        tried[bk] = {"ms_per_step": round(tb * 1e3, 3), "padding_waste": round(1.0 - st["real_tokens"] / st["padded_tokens_bucketed"], 4)}
        if t_bkt is None or tb < t_bkt:
            t_bkt, stats, bucket = tb, st, bk
    same = bool(torch.equal(out_a["ok"], out_b["ok"]) and torch.equal(out_a["n_match"], out_b["n_match"]) and
                float((out_a["R"] - out_b["R"]).abs().max()) < 1e-6)
    ok = int(out_b["ok"].sum().item())
    cert = certificate_block(eng, cal)
    del eng
    torch.cuda.empty_cache()
    real = stats["real_tokens"]
    return {"config": f"ragged batch: {batch} 640x480 pairs, N and M ~ U({lo}, {hi}) keypoints per side (an unbounded cv2.SIFT_create(), pose_node.py:122), "
                      f"headline precision", "batch": batch, "precision": precision, "steps": steps, "warmup": warmup,
            "keypoints_per_side": {"min": int(min(n_q.min(), n_r.min())), "mean": round(float((n_q.sum() + n_r.sum()) / (2 * batch)), 1), "max": int(max(n_q.max(), n_r.max()))},
            "one_padded_call": {"value": round(batch / t_one, 2), "ms_per_step": round(t_one * 1e3, 3),
                                "padded_tokens": stats["padded_tokens_one_call"], "padding_waste": round(1.0 - real / stats["padded_tokens_one_call"], 4)},
            "length_bucketed": {"value": round(batch / t_bkt, 2), "ms_per_step": round(t_bkt * 1e3, 3), "groups": stats["groups"], "pairs_per_group": bucket,
                                "padded_tokens": stats["padded_tokens_bucketed"], "padding_waste": round(1.0 - real / stats["padded_tokens_bucketed"], 4),
                                "group_sizes_tried": tried},
            "value": round(batch / min(t_one, t_bkt), 2), "unit": "pairs/s (this rank's GPU)", "ms_per_step": round(min(t_one, t_bkt) * 1e3, 3),
            "poses_ok_per_step": ok, "same_results_both_ways": same,
            "index_exact": "certified (margin guard, f32 re-run)" if cert else "tolerance mode", "certificate": cert}


def run_extra_superpoint(local_rank, batch, steps, warmup, dev, h=1080, w=1920, kpts=1024, arithmetic="split_fp16"):
    """BASELINE configs[4] on this rank's GPU: `batch` 1920x1080 frame<->tile pairs from PIXELS -- SuperPoint (exact-f32 convolutions) on
    2 x batch images, LightGlue(features="superpoint") in the headline precision, PnP-RANSAC.  Synthetic textured frames, the tile is a
    shifted crop of the same scene; seeded random weights (timing is what is measured; match quality is not)."""
    from gisnav_amd.superpoint import SuperPoint
    from gisnav_amd import _lib as glib
    sd_m = synthetic_state_dict(0, feature="superpoint", identity_blocks=True)    # blocks = identity: the match is the mutual nearest neighbour of the descriptors
    eng = PoseEngine(local_rank, max_batch=batch, max_kpts=kpts, precision="f16x2_f16_attn", state_dict=sd_m, filter_threshold=0.0, feature="superpoint")
    g = torch.Generator(device="cpu").manual_seed(5)
    conv_sd = {}
    sizes = [1, 64, 64, 128, 128]
    def conv(name, cout, cin, k):
        conv_sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        conv_sd[name + ".bias"] = torch.randn(cout, generator=g) * 0.05
    for b in range(4):
        conv(f"encoder.conv_blocks.{b}.conv_a", sizes[b + 1], sizes[b], 3)
        conv(f"encoder.conv_blocks.{b}.conv_b", sizes[b + 1], sizes[b + 1], 3)
    conv("keypoint_decoder.conv_score_a", 256, 128, 3); conv("keypoint_decoder.conv_score_b", 65, 256, 1)
    conv("descriptor_decoder.conv_descriptor_a", 256, 128, 3); conv("descriptor_decoder.conv_descriptor_b", 256, 256, 1)
    sp = SuperPoint(engine=eng, max_keypoints=kpts, state_dict=conv_sd, arithmetic=arithmetic)
    rs = np.random.default_rng(11)
    base = rs.uniform(0, 1, (batch, h // 8 + 3, w // 8 + 3)).astype(np.float32)
    big = torch.nn.functional.interpolate(torch.from_numpy(base)[:, None], size=(h + 16, w + 16), mode="bicubic", align_corners=False)[:, 0].clamp(0, 1)
    frames = big[:, 8:8 + h, 8:8 + w].contiguous().to(dev)
    tiles = big[:, 16:16 + h, 0:w].contiguous().to(dev)
    imgs = torch.cat([frames, tiles], 0)
    dem = torch.zeros((batch, h, w), dtype=torch.uint8, device=dev)
    Kc = np.array([[205.4696 * 3, 0.0, w / 2], [0.0, 205.4696 * 3, h / 2], [0.0, 0.0, 1.0]])
    eng.set_image_size((float(w), float(h)), (float(w), float(h)))
    out = eng.alloc_outputs(batch)
    t_sp = 0.0

    def step():
        nonlocal t_sp
        torch.cuda.synchronize(); t0 = time.perf_counter()
        kpt, score, desc, n = sp.detect_and_describe_device(imgs)
        torch.cuda.synchronize(); t_sp += time.perf_counter() - t0
        nd = torch.as_tensor(n, device=dev)
        inp = dict(desc_q=desc[:batch], kpt_q=kpt[:batch], n_q=nd[:batch], desc_r=desc[batch:], kpt_r=kpt[batch:], n_r=nd[batch:], dem=dem, kpt_format=glib.GN_KPT_XYSA)
        eng.estimate(inp, Kc, out=out)
        return n
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(); t_sp = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        n = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fp16 = arithmetic == "fp16"
    res = {"config": ("BASELINE configs[4] per GPU with 16-bit convolution operands (what configs[4] names; one fp16 product per block, f32 accumulate -- NOT f32-accurate: ~99 % of the exact path's keypoints): 1920x1080 frames -> SuperPoint -> LightGlue(superpoint), 1024 kpts -> PnP-RANSAC, from pixels" if fp16 else
                      "BASELINE configs[4] per GPU: 1920x1080 frames -> SuperPoint (split-fp16 MFMA convolutions: f32 operands as 2 fp16 terms, 3 products, f32 accumulate) -> LightGlue(superpoint), 1024 kpts -> PnP-RANSAC, from pixels"),
           "batch": batch, "keypoints_per_side": kpts, "precision": ("fp16 (single product) convolutions" if fp16 else "f16x2 convolutions") + " + f16x2_f16_attn matcher", "steps": steps, "warmup": warmup,
           "value": round(batch * steps / elapsed, 2), "unit": "pairs/s (this rank's GPU)", "ms_per_step": round(elapsed / steps * 1e3, 3),
           "superpoint_ms_per_image": round(t_sp / steps / (2 * batch) * 1e3, 3), "superpoint_gflop_per_image": 345.0,
           "superpoint_tflops": round(345.0 / (t_sp / steps / (2 * batch)) / 1e3, 1),
           "superpoint_frac_of_16bit_mfma_peak": round(345.0 / (t_sp / steps / (2 * batch)) / 1e3 / PEAK_16BIT_MFMA_TFLOPS, 4),
           "superpoint_frac_of_issue_ceiling": round((1.0 if fp16 else 3.0) * 345.0 / (t_sp / steps / (2 * batch)) / 1e3 / PEAK_16BIT_MFMA_TFLOPS, 4),
           "mean_keypoints": float(np.mean(n)), "mean_matches": float(out["n_match"].float().mean().item()), "poses_ok": int(out["ok"].sum().item()),
           "note": "random-init networks: keypoints / matches are whatever the untrained detector yields; the model named by configs[4] is not in the reference tree"}
    res["_pose"] = (out["R"].cpu().numpy().copy(), out["t"].cpu().numpy().copy(), out["ok"].cpu().numpy().copy().astype(bool))     # (popped by main: the pose delta between the arithmetics)
    del sp, eng
    torch.cuda.empty_cache()
    return res


def stream_new_pairs(eng, batches_msgs, steps, warmup, dev):
    """PCIe-INCLUSIVE rate of a stream of NEW pairs (never `value`): every step's batch arrives as host bytes -- the raw 532-byte
    KEYPOINT_DTYPE records of OrthoStereoImage.query_sift (pose_node.py:207-213) for both sides + the DEM raster -- and is copied into
    pinned memory and uploaded on a copy stream into one of three device slots while the GPU works on the previous batch (RecordStager);
    k_prep reads the records as they are.  The result records (16 f64 per pair) go back to pinned host memory every step."""
    from concurrent.futures import ThreadPoolExecutor
    from gisnav_amd.engine import RecordStager
    B = len(batches_msgs[0])
    st = RecordStager(eng, max_batch=B, max_kpts=eng.kmax, dem_hw=batches_msgs[0][0][2].shape, depth=3)
    out = eng.alloc_outputs(B)
    host_rec = torch.empty((B, gdist.RECORD_F64), dtype=torch.float64, pin_memory=True)
    nb = len(batches_msgs)
    t_stage = [0.0]

    def stage(i):
        t0 = time.perf_counter()
        r = st.stage(batches_msgs[i % nb])
        t_stage[0] += time.perf_counter() - t0
        return r

    def run(n_steps):
        with ThreadPoolExecutor(max_workers=1) as pool:
            cur = stage(0)
            for i in range(n_steps):
                fut = pool.submit(stage, i + 1) if i + 1 < n_steps else None
                st.wait(cur)
                eng.estimate(cur, K_MATRIX, out=out)
                st.release(cur)
                host_rec.copy_(gdist.pack_records(0, out), non_blocking=True)
                cur = fut.result() if fut else None
        eng.flush()
        torch.cuda.synchronize()

    run(max(warmup, 2))
    t_stage[0] = 0.0
    t0 = time.perf_counter()
    run(steps)
    elapsed = time.perf_counter() - t0
    mb = sum(len(q) + len(r) + d.nbytes for q, r, d in batches_msgs[0]) / 1e6
    return {"pairs_per_s": round(B * steps / elapsed, 2), "ms_per_step": round(elapsed / steps * 1e3, 4), "distinct_batches": nb,
            "host_to_device_mb_per_step": round(mb, 2), "host_stage_ms_per_batch": round(t_stage[0] / steps * 1e3, 3),
            "poses_ok_last_step": int(out["ok"].sum().item())}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: re-exec this script under torch.distributed.run with N ranks (one per GPU,
    backend nccl = RCCL) and return its exit code.  Refuses -- non-zero exit, nothing measured -- when the box has fewer GPUs than N."""
    have = torch.cuda.device_count()
    if have < args.gpus and not args.share_gpu:
        print(f"[bench] --gpus {args.gpus} asked for, but only {have} GPU(s) are visible on this box: refusing to run fewer ranks than "
              f"requested (use --gpus {max(have, 1)}, or --share-gpu --backend gloo for a dry run of the launch path)", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(gdist.free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print("[bench] spawning: " + " ".join(cmd), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def loftr_gflop_per_pair(h: int, w: int, fine: bool = True) -> float:
    """Algorithmic work of one LoFTR pair (2 x MACs): ResNet-FPN backbone on both images, 8 encoder-layer applications per image at 1/8
    resolution, the coarse similarity matrix (the fine level's per-match work is a few MFLOP per match and left out)."""
    p2, p4, p8 = (h // 2) * (w // 2), (h // 4) * (w // 4), (h // 8) * (w // 8)
    macs = p2 * 128 * 49 + 4 * p2 * 128 * 128 * 9
    macs += p4 * 196 * 128 * 9 + 3 * p4 * 196 * 196 * 9 + p4 * 196 * 128
    macs += p8 * 256 * 196 * 9 + 3 * p8 * 256 * 256 * 9 + p8 * 256 * 196 + p8 * 256 * 256
    if fine:
        macs += p4 * 256 * 196 + p4 * 256 * 256 * 9 + p4 * 196 * 256 * 9 + p2 * 196 * 128 + p2 * 196 * 196 * 9 + p2 * 128 * 196 * 9
    macs *= 2                                                            # two images
    macs += 2 * 8 * p8 * (3 * 256 * 256 + 256 * 256 + 512 * 512 + 256 * 512)
    macs += p8 * p8 * 256
    return 2.0 * macs / 1e9


def run_extra_loftr(local_rank, steps, warmup, dev, h=480, w=640, fine=True, graph=True, arithmetic="exact_f32"):
    """BASELINE configs[1] as literally worded: ONE 640x480 pair through the LoFTR matcher in fp32 (gn_loftr_match: ResNet-FPN backbone on the
    exact-f32 matrix instruction, linear-attention transformer, dual-softmax coarse matching, fine level), seeded random weights."""
    from gisnav_amd import loftr_synthetic as olf
    from gisnav_amd.loftr import LoFTR
    sd = olf.synthetic_state_dict(0)
    i0, i1 = olf.synthetic_pair(1, h, w)
    m = LoFTR(state_dict=sd, fine=fine, graph=graph, arithmetic=arithmetic).to(dev).eval()
    data = {"image0": i0.to(dev), "image1": i1.to(dev)}
    for _ in range(warmup):
        out = m(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = m(data)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    g = loftr_gflop_per_pair(h, w, fine)
    res = {"config": f"BASELINE configs[1] as worded: batch-1 {w}x{h} pair, LoFTR matcher (ResNet-FPN + 4x(self, cross) linear attention + dual-softmax coarse matching"
                     f"{' + fine level' if fine else ''}), " + ("fp32 (exact-f32 MFMA)" if arithmetic == "exact_f32" else
                     "f32-ACCURATE split fp16 (each f32 operand as 2 fp16 terms, 3 MFMA products, f32 accumulate; fp16-range guard with exact re-run)"),
           "batch": 1, "precision": "f32" if arithmetic == "exact_f32" else "f16x2 (f32-accurate)", "hip_graph": bool(graph), "steps": steps, "warmup": warmup, "value": round(steps / elapsed, 2), "unit": "pairs/s (this rank's GPU)",
           "ms_per_step": round(elapsed / steps * 1e3, 3), "matches": int(out["keypoints0"].shape[0]), "algorithmic_gflop_per_pair": round(g, 1),
           "end_to_end_tflops": round(g / (elapsed / steps) / 1e3, 1),
           "peak_tflops": PEAK_F32_MFMA_TFLOPS if arithmetic == "exact_f32" else PEAK_16BIT_MFMA_TFLOPS,
           "end_to_end_frac_of_peak": round(g / (elapsed / steps) / 1e3 / (PEAK_F32_MFMA_TFLOPS if arithmetic == "exact_f32" else PEAK_16BIT_MFMA_TFLOPS), 4),
           "mfma_flops_issued_per_algorithmic_flop": 1 if arithmetic == "exact_f32" else 3,
           "note": "the model named by configs[1] is not in the reference tree; random-init weights calibrated so that the dual-softmax has confident mutual maxima"}
    del m
    torch.cuda.empty_cache()
    return res


def precision_guarantee(precision: str, cert=None) -> dict:
    """What this run can say about `north_star`'s "correspondence indices bit-exact", and the evidence behind it -- READ from the report the -m gpu
    parity tests wrote (tests/test_gpu_round6.py -> gpurun_out/parity_r06.json, committed as profiles/r06_parity_report.json), never typed in.  The
    report carries the source digest of the LIBRARY it was measured on (gn_source_digest, compiled into the binary); `same_build` compares it with the
    digest of the library THIS run loaded."""
    from gisnav_amd import _lib
    exact = "exact-f32 arithmetic (empirically index-exact: f32_evidence)" if precision == "f32" else \
        "certified (margin guard, f32 re-run)" if cert else "tolerance mode"
    out = {"mode": precision, "index_exact": exact, "certificate": cert,
           "note": "north_star asks for bit-exact correspondence indices (pose_node.py:285-297 uses them as integers).  The fast mode computes the assignment "
                   "scores with an arithmetic error; with the certificate on, the match head keeps the runner-up of every row / column maximum, flags every "
                   "pair in which a decision lies within eps (calibrated: max |P_mode - P_f32| x safety) of flipping -- or whose activations left the fp16 "
                   "range -- and the flagged pairs are run again on the exact-f32 kernels before the call returns.  Counts below: symmetric differences of "
                   "the match sets against the CPU restatement of the reference (the tests' checker; itself UNPINNED against kornia, DESIGN 2), on the "
                   "headline kernels (asserted from the launch table)",
           "library_source_digest": _lib.library_digest()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_parity_report.json")
    try:
        with open(path) as f:
            rep = json.load(f)
    except (OSError, ValueError) as exc:
        out["report"] = f"not available ({exc.__class__.__name__}): no counts are claimed for this run"
        return out
    out["report"] = "profiles/r06_parity_report.json"
    out["report_source_digest"] = rep.get("source_digest")
    out["same_build"] = rep.get("source_digest") == out["library_source_digest"]
    keep = ("eps", "safety", "cpu_matches", "uncertified_index_mismatches", "certified_index_mismatches", "pairs", "pairs_flagged", "rerun_fraction", "f32_marginal_pairs")
    out["certified_tables"] = {k[len("certified_"):]: {f: v.get(f) for f in keep} for k, v in rep.items() if k.startswith("certified_") and isinstance(v, dict)}
    out["f32_evidence"] = {k[len("f32_"):]: v for k, v in rep.items() if k.startswith("f32_") and isinstance(v, dict)}
    out["source"] = "tests/test_gpu_round6.py (asserted: certified = 0 mismatches and no wrong pair unflagged on every table; f32 = 0 on those tables) + tools/f32_exactness_sweep.py (many weight sets; asserted: the exact-f32 mode differs from the CPU restatement only in pairs its own certificate marks as holding a decision within 1e-4, and the certified fast mode only where the f32 mode does)"
    if not out["same_build"]:
        out["warning"] = "the report was measured on another build than the library this run loaded: re-run `pytest -m gpu tests/test_gpu_round6.py` and copy gpurun_out/parity_r06.json"
    return out


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def cpu_baseline(state_dict, kpts: int, seconds_budget: float = 20.0):
    """The oracle (restated reference: torch-CPU LightGlue-sift + numpy solvePnPRansac) timed on this box's host cores, on a bounded sample of the
    same workload -- twice: with every hardware thread the process may use (`torch.set_num_threads(os.cpu_count())`) and with a
    32-thread cap (the 1024 x 256-sized CPU GEMMs of one pair stop scaling long before 256 threads).  `value` is the FASTER of the two (the baseline
    gets the benefit of the doubt); both are printed."""
    from oracle import lightglue_sift as lg
    from oracle import pnp_ransac as pr
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state_dict.items()}
    tq = torch.from_numpy

    def leg(threads, budget):
        torch.set_num_threads(max(1, threads))
        times, poses, i = [], 0, 0
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
            if (len(times) >= 3 and time.perf_counter() - t_start > budget) or time.perf_counter() - t_start > 6 * budget:
                break
        med = float(np.median(times))
        return {"threads": torch.get_num_threads(), "value": round(1.0 / med, 4), "unit": "pairs/s", "pairs_timed": len(times), "median_s_per_pair": round(med, 4)}

    legs = []
    if avail > 32:
        # every hardware thread, in a child process with a wall-clock limit: on the 256-thread hosts of this pool torch's CPU GEMMs
        # of this size collapse under oversubscription (one pair did not finish in a minute), and a leg that cannot be interrupted would eat the
        # bench's time budget.  The child times the same pairs the same way.
        import subprocess
        code = ("import json, os, sys, time, numpy as np, torch\n"
                f"sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})\n"
                "from oracle import lightglue_sift as lg\nfrom oracle import pnp_ransac as pr\n"
                "from gisnav_amd.synthetic import K_MATRIX, make_pair\nfrom gisnav_amd.weights import synthetic_state_dict\n"
                f"torch.set_num_threads({avail}); kpts = {kpts}; budget = {seconds_budget / 2}\n"
                "sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic_state_dict(0).items()}; tq = torch.from_numpy\n"
                "times = []; i = 0; t_start = time.perf_counter()\n"
                "while True:\n"
                "    p = make_pair(10_000 + i, n_q=kpts, n_r=kpts); t0 = time.perf_counter()\n"
                "    mq, mr, _, _ = lg.pose_node_match(sd, tq(p.kp_q), tq(p.desc_q), tq(p.size_q), tq(p.angle_q), tq(p.kp_r), tq(p.desc_r), tq(p.size_r), tq(p.angle_r))\n"
                "    if len(mq) >= 15: pr.compute_pose(K_MATRIX.reshape(-1), mq.numpy(), mr.numpy(), p.dem)\n"
                "    times.append(time.perf_counter() - t0); i += 1\n"
                "    print(json.dumps(times), flush=True)\n"
                "    if (len(times) >= 4 and time.perf_counter() - t_start > budget) or time.perf_counter() - t_start > 3 * budget: break\n")
        limit = 2.0 * seconds_budget
        row = {"threads": avail, "unit": "pairs/s", "wall_limit_s": limit}
        try:
            t0 = time.perf_counter()
            out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=limit).stdout
        except subprocess.TimeoutExpired as exc:
            out = (exc.stdout.decode() if isinstance(exc.stdout, bytes) else exc.stdout) or ""
            row["note"] = f"stopped at the wall-clock limit after {time.perf_counter() - t0:.0f} s"
        lines = [ln for ln in out.splitlines() if ln.startswith("[")]
        times = json.loads(lines[-1]) if lines else []
        if len(times) >= 2:
            med = float(np.median(times[1:]))
            row.update({"value": round(1.0 / med, 4), "pairs_timed": len(times) - 1, "median_s_per_pair": round(med, 4)})
        else:
            row.update({"value": None, "pairs_timed": 0, "note": row.get("note", "") + f"; {len(times)} pair(s) finished inside the limit (the first is warm-up)"})
        legs.append(row)
    legs.append(leg(min(avail, 32), seconds_budget / 2 if avail > 32 else seconds_budget))
    best = max((r for r in legs if r.get("value")), key=lambda r: r["value"])
    torch.set_num_threads(best["threads"])
    res = {"value": best["value"], "unit": "pairs/s", "cores": best["threads"], "kind": "port",
           "sample": f"{best['pairs_timed']} synthetic 640x480 pairs ({kpts} kpts/side) after 1 warm-up pair, median, at {best['threads']} threads (the faster of "
                     f"the thread counts in by_threads); torch-CPU fp32 LightGlue-sift restatement + numpy solvePnPRansac restatement (oracle/); "
                     f"cpu={_cpu_model()}",
           "by_threads": legs,
           "cpu_model": _cpu_model(), "threads_used": best["threads"], "threads_available": avail, "os_cpu_count": os.cpu_count()}
    try:   # the same host cores on configs[1] as worded (the LoFTR restatement, oracle/loftr.py): one warm-up forward, one timed
        from oracle import loftr as olf
        sdl = olf.synthetic_state_dict(0)
        i0, i1 = olf.synthetic_pair(1, 480, 640)
        olf.loftr_forward(sdl, i0, i1)
        t0 = time.perf_counter()
        r = olf.loftr_forward(sdl, i0, i1)
        dt = time.perf_counter() - t0
        res["loftr_640x480"] = {"value": round(1.0 / dt, 4), "unit": "pairs/s", "ms_per_pair": round(dt * 1e3, 1), "matches": int(len(r["i_ids"])),
                                "sample": "one 640x480 pair after one warm-up, torch-CPU fp32 restatement of kornia's LoFTR (oracle/loftr.py), same threads"}
    except Exception as exc:  # noqa: BLE001   (a reported side number: never fails the line)
        res["loftr_640x480"] = {"error": repr(exc)[:200]}
    return res


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="pairs per GPU per step")
    ap.add_argument("--kpts", type=int, default=1024)
    ap.add_argument("--precision", default="f16x2_f16_attn", choices=["f32", "bf16_attn", "f32x3_bf16_attn", "f16x2_bf16_attn", "f16x2_f16_attn"])
    ap.add_argument("--resident-batches", type=int, default=4, help="distinct staged batches (all resident in HBM) the timed steps rotate over")
    ap.add_argument("--only-loftr", action="store_true", help="time only the LoFTR extra (configs[1] as worded) and print its JSON")
    ap.add_argument("--no-rccl-check", action="store_true", help="skip the world-1 RCCL self-check of the N = 1 run (multi_gpu.rccl_selfcheck_world1)")
    ap.add_argument("--no-stream", action="store_true", help="skip the PCIe-inclusive streaming measurement (pcie_inclusive)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes (HBM bytes per step)")
    ap.add_argument("--no-extras", action="store_true", help="skip extra_configs (batch-1 f32, batch-32 exact-f32 GEMMs)")
    ap.add_argument("--overlap", action="store_true", help="run the PnP stage of step n on a second stream beside the matcher of step n+1 "
                                                           "(gn_set_overlap; measured gain < 1 %%)")
    ap.add_argument("--substreams", type=int, default=-1, help="sub-batch streams inside one gn_estimate call (gn_set_substreams): the batch is cut into that many "
                                                               "groups whose matcher / PnP stages overlap on internal streams, joined before the call returns; "
                                                               "default 2 for batches >= 16 pairs (measured -3 %% step time), else 1")
    ap.add_argument("--deferred-join", action="store_true", help="with --substreams > 1: leave the join of the sub-batch groups to the flush at the end of the timed "
                                                                 "region (gn_set_deferred_join): consecutive steps pipeline inside each group's stream")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--debug-variant", action="append", default=[], metavar="WHICH:VALUE",
                    help="developer knob: gn_debug_set_variant(which, value) before the run (timing experiments; RECORDED in the JSON line, "
                         "a line with a non-empty debug_variant is not a valid measurement)")
    ap.add_argument("--no-certify", action="store_true", help="run the fast precision mode WITHOUT the margin certificate (gn_set_certify): round 5's tolerance-mode "
                                                              "number; the line then says index_exact: tolerance mode")
    ap.add_argument("--ffn-products", type=int, default=0, choices=[0, 2, 3],
                    help="fp16 partial products of the block tail's GEMMs on bulk grids (gn_set_ffn_products): 3 = f32-accurate split (rounds 1-5), 2 = activations' "
                         "high term only, 0 (default) = the level follows the certificate's flags (two products only while they flag no more pairs than three would); "
                         "the fixed levels are reported as extra configurations")
    ap.add_argument("--certify-eps", type=str, default="", help="profiling passes: use this eps instead of calibrating (the calibration's f32 pass and the "
                                                                     "fused-projection self-check are set-up work that a rocprofv3 pass over few steps would count as steps)")
    ap.add_argument("--sync-certify", action="store_true", help="certificate resolved inside every call (one host synchronisation per step) instead of one call later")
    ap.add_argument("--share-gpu", action="store_true", help="dry-run aid: every rank uses cuda:0 (with --backend gloo)")
    args = ap.parse_args()

    if args.only_loftr:
        torch.cuda.set_device(0)
        print(json.dumps(run_extra_loftr(0, args.steps, args.warmup, torch.device("cuda", 0))), flush=True)
        print(json.dumps(run_extra_loftr(0, args.steps, args.warmup, torch.device("cuda", 0), fine=False)), flush=True)
        print(json.dumps(run_extra_loftr(0, args.steps, args.warmup, torch.device("cuda", 0), arithmetic="split_fp16")), flush=True)
        print(json.dumps(run_extra_loftr(0, args.steps, args.warmup, torch.device("cuda", 0), fine=False, arithmetic="split_fp16")), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))            # no launcher around us: become one (N ranks over RCCL), or refuse
    if int(os.environ.get("WORLD_SIZE", 1)) != args.gpus:
        print(f"[bench] WORLD_SIZE={os.environ.get('WORLD_SIZE')} does not match --gpus {args.gpus}: refusing to report a line for a world the "
              f"caller did not ask for", file=sys.stderr)
        sys.exit(2)
    if not args.share_gpu and torch.cuda.device_count() < args.gpus:
        print(f"[bench] --gpus {args.gpus} asked for, but only {torch.cuda.device_count()} GPU(s) are visible on this box", file=sys.stderr)
        sys.exit(2)
    rccl_check = None
    if args.gpus == 1 and args.backend == "nccl" and not args.debug_variant and not args.no_rccl_check and "WORLD_SIZE" not in os.environ:
        # (under a launcher -- WORLD_SIZE set, even to 1 -- the run below IS an nccl process group: its broadcast / gathers are the proof, and a
        #  child that inherits the launcher's rendezvous variables was seen to hang on a 1-GPU box)
        try:   # the collectives of the N > 1 path on a world-1 RCCL group: what a 1-GPU lease can prove.  In a child process: RCCL writes a
            #      banner to the C-level stdout, and this process's stdout carries exactly one line
            launcher_vars = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                             "MASTER_ADDR", "MASTER_PORT")
            env = {k: v for k, v in os.environ.items() if k not in launcher_vars and not k.startswith("TORCHELASTIC_")}
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            r = subprocess.run([sys.executable, "-m", "gisnav_amd.dist", "--selfcheck"], cwd=ROOT, capture_output=True, text=True, timeout=120, env=env)
            js = [l for l in r.stdout.splitlines() if l.startswith("{")]
            rccl_check = json.loads(js[-1]) if js else {"ok": False, "error": (r.stderr or r.stdout)[-300:]}
        except Exception as exc:  # noqa: BLE001
            rccl_check = {"ok": False, "error": repr(exc)[:300]}
    rank, local_rank, world = gdist.init(args.backend)
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # every rank alive, every rank on its own GPU -- or no line at all
    idents = gdist.gather_strings(f"{rank}:{gdist.device_identity(local_rank)}")
    ranks_seen = len({i.split(":", 1)[0] for i in idents})
    gpus_distinct = len({i.split(":", 1)[1] for i in idents})
    if ranks_seen != args.gpus or (gpus_distinct != args.gpus and not args.share_gpu):
        if rank == 0:
            print(f"[bench] {ranks_seen} ranks on {gpus_distinct} distinct GPUs, --gpus {args.gpus} asked for: refusing", file=sys.stderr)
        sys.exit(2)

    # weights: seeded synthetic on rank 0, broadcast once over RCCL (no checkpoint is available offline)
    sd = synthetic_state_dict(0)
    t_b = time.perf_counter()
    if world > 1:
        if rank != 0:
            sd = {k: np.zeros_like(v) for k, v in sd.items()}
        sd = gdist.broadcast_state_dict(sd, dev, src=0)
        torch.cuda.synchronize()
    broadcast_ms = (time.perf_counter() - t_b) * 1e3
    eng = PoseEngine(local_rank, max_batch=args.batch, max_kpts=args.kpts, precision=args.precision, state_dict=sd)
    for kv in args.debug_variant:
        which, value = (int(v) for v in kv.split(":"))
        eng.lib.gn_debug_set_variant(eng.ctx, which, value)
    # default 0: the level follows the certificate.  Two products are ~7 % faster per step and ten times less accurate (eps): under the certificate the indices
    # stay exact either way, but the share of pairs that must be re-run in f32 grows with eps on anything but wide-margin weights -- so the
    # context counts what each level's certificate flags and uses two products only while that costs no extra re-runs (without the certificate: three)
    ffn_products = args.ffn_products
    if args.precision.startswith("f16x2"):
        eng.set_ffn_products(ffn_products)         # (before the calibration: eps is measured for the arithmetic that runs)
    if args.certify_eps:
        eng.lib.gn_debug_set_variant(eng.ctx, 47, 1)      # (a profiling pass: no self-check launches among the steps; the full run's line carries its verdict)
    cert_cal = certify_on(eng, args.kpts, args.batch, args.precision, certify=not args.no_certify, eps=args.certify_eps)

    # this rank's contiguous shard of the global batch, staged into HBM before the timed region
    shard = gdist.shard_range(args.batch * world, rank, world)
    nres = max(1, args.resident_batches)
    # `nres` DISTINCT batches per rank (batch j of rank r = pairs j * B * world + shard), all staged into HBM before the timed region;
    # the timed steps rotate over them, so no step re-reads what the previous one just touched
    pair_sets = [[make_pair(j * args.batch * world + i, n_q=args.kpts, n_r=args.kpts) for i in shard] for j in range(nres)]
    pairs = pair_sets[0]
    torch.cuda.synchronize()
    t_h = time.perf_counter()
    inp = eng.stage_inputs(pairs)
    torch.cuda.synchronize()
    h2d_ms = (time.perf_counter() - t_h) * 1e3     # host packing + PCIe upload of one batch (pageable host memory): what a per-message caller pays
    inps = [inp] + [eng.stage_inputs(ps) for ps in pair_sets[1:]]
    out = eng.alloc_outputs(len(pairs))
    if args.overlap:
        eng.set_overlap(True)
    nsub = args.substreams if args.substreams > 0 else (2 if args.batch >= 16 else 1)
    # Per-kernel figures (roofline, kernel table) come from a pass with ONE stream: with sub-batch streams the launches of two groups
    # run concurrently and a launch's HIP-event duration stops being that kernel's own time.  `value` is timed afterwards, in its own
    # K steps, with the sub-batch streams on.
    eng.set_substreams(1)
    elapsed1, _ = timed_steps(eng, inps, out, args.steps, args.warmup, dev, kernel_timing=True)   # (_ = this rank's own seconds)
    table = eng.kernel_table()
    eng.set_kernel_timing(0)
    if nsub > 1:
        eng.set_substreams(nsub, deferred_join=args.deferred_join)
        outs = out
        if cert_cal is not None and not args.sync_certify:
            # the flags of step n are read, and its flagged pairs re-run, after step n + 1 has been enqueued (gn_set_certify(3)): the host
            # never waits for an idle GPU.  Step n's outputs must then stay untouched for one more step: two output sets, alternating
            eng.set_certify("deferred")
            cert_cal["resolution"] = "deferred by one call (gn_set_certify(3)); two alternating output sets"
            outs = [eng.alloc_outputs(len(pairs)), eng.alloc_outputs(len(pairs))]
        elapsed, mine = timed_steps(eng, inps, outs, args.steps, args.warmup, dev, kernel_timing=False)
        if isinstance(outs, list):
            out = outs[(args.steps - 1) % 2]           # the last timed step's results (final: timed_steps flushed)
            eng.set_certify("rerun")
    else:
        elapsed, mine = elapsed1, _
    tripped, trips = eng.guard_status()
    fused_status = eng.fused_projection_status()
    cert_main = certificate_block(eng, cert_cal)

    # what the block tail of THIS run computed in (the `dtype` text names the three-product form; the automatic level may have chosen two)
    tail_note = ""
    if args.precision.startswith("f16x2"):
        lvl = (cert_main or {}).get("block_tail_level")
        two = lvl["certified_calls_on_two_products"] if lvl else (1 if ffn_products == 2 else 0)
        tot = (lvl["certified_calls_on_two_products"] + lvl["certified_calls_on_three_products"]) if lvl else 1
        if two:
            tail_note = (f" || BLOCK TAIL (ffn.0 -> LayerNorm -> GELU -> ffn.3 with out_proj folded in) in this run: TWO partial products -- the activations' fp16 high "
                         f"term x the 22-bit weights, f32 accumulate, the arithmetic of the attention input projections -- in {two} of {tot} certified calls"
                         + (": the level the margin certificate chose (gn_set_ffn_products(0), DESIGN 10.2)" if lvl else " (gn_set_ffn_products(2))")
                         + "; the correspondence indices are certified against the exact-f32 arithmetic on either level")
    n_ok_all = gdist.sum_over_ranks(float(out["ok"].sum().item()), dev)
    n_match_mean = float(out["n_match"].float().mean().item())
    # the synthetic pairs are built so that every one yields a pose: a step that produced (almost) none has run a broken hot path, and its time is not a measurement
    if not args.debug_variant and n_ok_all < 0.9 * args.batch * world:
        raise SystemExit("bench.py: only %d of %d pairs of the last step produced a pose (mean matches %.1f) -- refusing to print a line for a broken hot path"
                         % (int(n_ok_all), args.batch * world, n_match_mean))
    # the LAST timed step worked on batch (steps - 1) mod nres: its result records, gathered over RCCL, must contain this rank's block
    last = (args.steps - 1) % nres
    mine_rec = gdist.pack_records(last * args.batch * world + shard.start, out)
    rec = gdist.gather_records(mine_rec)  # fixed-size result records, 128 B/pair
    block_ok = bool(torch.equal(rec[rank * len(pairs): (rank + 1) * len(pairs)].cpu(), mine_rec.cpu()))
    blocks_ok = int(gdist.sum_over_ranks(1.0 if block_ok else 0.0, dev))
    import hashlib
    rec_sha = hashlib.sha256(rec.cpu().numpy().tobytes()).hexdigest()
    per_rank_ms = gdist.gather_records(torch.tensor([[mine / args.steps * 1e3] + [0.0] * 15], dtype=torch.float64, device=dev))[:, 0].cpu().tolist()
    stream = None
    if not args.no_stream and not args.debug_variant and world == 1:
        from gisnav_amd import wire
        msgs = [[(wire.pack_keypoints(p.kp_q, p.size_q, p.angle_q, p.desc_q), wire.pack_keypoints(p.kp_r, p.size_r, p.angle_r, p.desc_r), p.dem)
                 for p in ps] for ps in pair_sets]
        stream = stream_new_pairs(eng, msgs, args.steps, args.warmup, dev)
    del eng
    torch.cuda.empty_cache()

    extras = []
    if not args.no_extras and not args.debug_variant and world == 1:   # per-GPU side configurations: reported by the N = 1 line; an N > 1 run times the sharded headline only
        extras.append(run_extra(local_rank, sd, "BASELINE configs[1] as SURVEY.md reads it: batch-1 640x480 pair, f32 everywhere (exact-f32 MFMA GEMMs and attention)",
                                1, args.kpts, "f32", 30, 5, dev))
        extras.append(run_extra(local_rank, sd, "BASELINE configs[2] with exact-f32 MFMA projections/FFN (no fp16 split) + bf16 MFMA attention (a tolerance mode: no certificate)",
                                args.batch, args.kpts, "bf16_attn", 6, 2, dev, certify=False))
        if cert_cal is not None:
            extras.append(run_extra(local_rank, sd, "the headline configuration WITHOUT the certificate (round 5's tolerance-mode number: no per-call synchronisation, no re-run)",
                                    args.batch, args.kpts, args.precision, 20, 3, dev, certify=False, substreams=nsub))
        if cert_cal is not None:
            # what the certificate costs when decisions are NOT far from flipping: the same configuration on the tests' mid-margin weight family
            # (filter_threshold 0.01: tests/test_gpu_round6.py) -- flagged pairs are re-run in exact f32 inside the timed region -- and the opt-in
            # two-product block tail (a third less matrix work in the tail, ten times the eps) on both weight sets
            mid = synthetic_state_dict(0, ffn_out_std=1.2e-3, final_scale=12.0, matchability_bias=2.0, matchability_std=0.05)
            extras.append(run_extra(local_rank, sd, "the headline configuration with the block tail FIXED on three partial products (gn_set_ffn_products(3): the f32-accurate fast pass of rounds 1-5): bench weights",
                                    args.batch, args.kpts, args.precision, 20, 3, dev, substreams=nsub, ffn_products=3))
            extras.append(run_extra(local_rank, sd, "the headline configuration with the block tail FIXED on two partial products (gn_set_ffn_products(2)): bench weights",
                                    args.batch, args.kpts, args.precision, 20, 3, dev, substreams=nsub, ffn_products=2))
            extras.append(run_extra(local_rank, mid, "the headline configuration on MID-MARGIN weights (decisions close to flipping: re-runs inside the timed region), AUTOMATIC block-tail level as in the headline",
                                    args.batch, args.kpts, args.precision, 6, 2, dev, substreams=nsub, ffn_products=0, filter_threshold=0.01))
            extras.append(run_extra(local_rank, mid, "the headline configuration on MID-MARGIN weights, fixed on three products",
                                    args.batch, args.kpts, args.precision, 6, 2, dev, substreams=nsub, ffn_products=3, filter_threshold=0.01))
            extras.append(run_extra(local_rank, mid, "the headline configuration on MID-MARGIN weights, fixed on two products",
                                    args.batch, args.kpts, args.precision, 6, 2, dev, substreams=nsub, ffn_products=2, filter_threshold=0.01))
        extras.append(run_extra(local_rank, sd, f"BASELINE configs[2] in the GUARANTEED mode: batch-{args.batch}, f32 everywhere (exact-f32 MFMA GEMMs and attention) -- "
                                                "the arithmetic the certificate re-runs flagged pairs in; correspondence indices identical to the CPU restatement of the reference on every "
                                                "weight set tested (counts: precision_guarantee.f32_evidence)",
                                args.batch, args.kpts, "f32", 3, 1, dev))
        extras.append(run_extra_ragged(local_rank, sd, args.batch, args.precision, 5, 2, dev, certify=not args.no_certify))
        extras.append(run_extra(local_rank, sd, "batch-1 640x480 pair in the headline precision (latency of one ROS message, SURVEY F5)",
                                1, args.kpts, args.precision, 30, 5, dev, certify=not args.no_certify))
        extras.append(run_extra_loftr(local_rank, 10, 2, dev))
        extras.append(run_extra_loftr(local_rank, 10, 2, dev, arithmetic="split_fp16"))
        extras.append(run_extra_superpoint(local_rank, 4, 2, 1, dev))
        extras.append(run_extra_superpoint(local_rank, 4, 2, 1, dev, arithmetic="fp16"))
        # the same two at configs[3]'s per-GPU batch (32 pairs: the extractor still runs four frames per pass, the matcher and the PnP stage run once)
        extras.append(run_extra_superpoint(local_rank, 32, 2, 1, dev))
        extras.append(run_extra_superpoint(local_rank, 32, 2, 1, dev, arithmetic="fp16"))
        # what the 16-bit convolution operands do to the POSE: the same 32 frame / tile pairs through both extractors, everything
        # behind them identical -- per pair |R_fp16 - R_split|_F and |t_fp16 - t_split| / |t_split| over the pairs that yield a pose both ways
        (Ra, ta, oka), (Rb, tb, okb) = extras[-2]["_pose"], extras[-1]["_pose"]
        both = oka & okb
        if both.any():
            dR = np.linalg.norm((Ra - Rb)[both].reshape(-1, 9), axis=1)
            dt = np.linalg.norm((ta - tb)[both].reshape(-1, 3), axis=1) / np.maximum(np.linalg.norm(ta[both].reshape(-1, 3), axis=1), 1e-12)
            extras[-1]["pose_delta_vs_f32_accurate_extractor"] = {"pairs_with_a_pose_both_ways": int(both.sum()), "pairs": int(len(both)),
                                                                  "dR_frobenius": {"median": float(np.median(dR)), "max": float(dR.max())},
                                                                  "dt_relative": {"median": float(np.median(dt)), "max": float(dt.max())}}
        else:
            extras[-1]["pose_delta_vs_f32_accurate_extractor"] = {"pairs_with_a_pose_both_ways": 0, "pairs": int(len(both))}
        for e in extras:
            e.pop("_pose", None)

    if rank == 0:
        total_pairs = args.batch * world * args.steps
        pairs_per_s = total_pairs / elapsed
        g_pair, g_attn = gflop_per_pair(args.kpts, args.kpts)
        f32_all = args.precision == "f32"
        e2e_peak = PEAK_F32_MFMA_TFLOPS if f32_all else PEAK_16BIT_MFMA_TFLOPS
        e2e_tf = pairs_per_s / world * g_pair / 1e3          # per GPU
        rows = kernel_rows(table, args.steps, args.precision)
        # the dominant kernel: the template instantiations of one kernel are ONE family (k_ffn128<.., 0 / 1 / 2> are the same block tail with
        # no / the self / the cross projection behind it; rocprofv3 lists them as three names) -- the family with the largest share of the
        # timed kernel time, represented by its largest member (whose name and average duration profiles/*kernel_stats*.csv can be checked against)
        fam = {}
        for r_ in rows:
            fam.setdefault(r_["name"].split("<")[0], []).append(r_)
        dom_family = max(fam.values(), key=lambda ms: sum(m["share_of_timed_kernel_time"] for m in ms)) if fam else []
        dom = max(dom_family, key=lambda m: m["share_of_timed_kernel_time"]) if dom_family else None
        traffic, traffic_err = (None, "skipped")
        if not args.no_traffic and world == 1:
            traffic, traffic_err = measure_traffic(args, certify_eps=(cert_main or {}).get("profile_eps_arg", ""))
        comp_mb = compulsory_mb_per_pair(args.kpts, args.kpts) * args.batch + 47.5   # + the f32 weights once per step (cache-resident in practice)
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
                      "f16x2_bf16_attn": "f32-accurate FFN / out-projection / match-head GEMMs (each f32 operand split into 2 fp16 terms = 22 significant bits, 3 fp16 MFMA "
                                         "partial products, f32 accumulate; fp16-range guard active); attention INPUT projections (q, k, v: 18 of 66 launches) on 2 partial "
                                         "products -- the fp16 high term of the activations (11 bits) times the 22-bit weights, outputs rounded to bf16 for the attention; "
                                         "bf16 MFMA attention (f32 softmax / accumulate)",
                      "f16x2_f16_attn": "f32-accurate FFN / out-projection / match-head GEMMs (each f32 operand split into 2 fp16 terms = 22 significant bits, 3 fp16 MFMA "
                                        "partial products, f32 accumulate; fp16-range guard active); attention INPUT projections (q, k, v: 18 of 66 launches) on 2 partial "
                                        "products -- the fp16 high term of the activations (11 bits) times the 22-bit weights, one rounding more than half(fp32 Linear(x)); "
                                        "their outputs are rounded to fp16 for the attention either way (10.5 % of the q/k/v values move by one fp16 ulp, DESIGN 10.3) --; "
                                        "fp16 MFMA attention (q, k, v, p rounded to fp16 like the reference's CUDA SDPA; f32 softmax / accumulate)"}[args.precision] + tail_note,
            "precision_guarantee": precision_guarantee(args.precision, cert_main),
            "data": "synthetic",
            "inputs_resident": True,
            "debug_variant": list(args.debug_variant),
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
                "sub_batch_streams": nsub,
                "single_stream_ms_per_step": round(elapsed1 / args.steps * 1e3, 4),
                "weights": "seeded synthetic, kornia sift_lightglue state-dict layout",
                "same_staged_batch_every_step": nres == 1,
                "distinct_resident_batches_rotated": nres,
            },
            "poses_per_s": round(n_ok_all * args.steps / elapsed, 2),
            "poses_ok_per_step": int(n_ok_all),
            "mean_matches_per_pair": round(n_match_mean, 1),
            "result_records_gathered": int(rec.shape[0]),
            "f16x2_guard": {"tripped_in_last_step": bool(tripped), "trips_observed": int(trips)},
            "fused_projection_selfcheck": {1: "passed: bitwise equal to the separate k_qkv launches on these weights (checked at the first call)",
                                           0: "FAILED: fusion switched off for this context", -1: "not applicable / not run"}.get(fused_status),
            "block_tail_partial_products": (ffn_products or "automatic: see precision_guarantee.certificate.block_tail_level") if args.precision.startswith("f16x2") else None,
            "end_to_end": {"algorithmic_gflop_per_pair": round(g_pair, 2), "attention_gflop_per_pair": round(g_attn, 2),
                           "achieved_tflops_per_gpu": round(e2e_tf, 1), "peak_tflops": e2e_peak, "frac": round(e2e_tf / e2e_peak, 4),
                           "attention_only_frac": round(pairs_per_s / world * g_attn / 1e3 / e2e_peak, 4),
                           "ceiling_pairs_per_s_per_gpu": round(e2e_peak * 1e3 / g_pair, 0),
                           "note": "SURVEY.md 8(d): achieved = pairs/s x algorithmic GFLOP per pair; peak = dense 16-bit MFMA (f32 MFMA in the f32 mode)"},
            "pcie_inclusive": {"pairs_per_s": stream["pairs_per_s"] if stream else None,
                               "frac_of_value": round(stream["pairs_per_s"] / pairs_per_s, 4) if stream else None,
                               "stream": stream,
                               "synchronous_stage_inputs_ms_per_batch": round(h2d_ms, 3),
                               "synchronous_pairs_per_s": round(args.batch * world / (elapsed / args.steps + h2d_ms * 1e-3), 2),
                               "note": "stream: NEW pairs every step from host bytes (raw 532-byte keypoint records + DEM, pinned staging + copy stream, unpacked "
                                       "on the device; result records copied back) -- the reference's per-message boundary (pose_node.py:207-213,254-265) "
                                       "pipelined; synchronous_*: round 2's figure (one blocking host-pack + pageable H2D per step).  Never `value`"},
            "multi_gpu": {"ranks_seen": ranks_seen, "gpus_distinct": gpus_distinct, "devices": idents, "backend": args.backend,
                          "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms], "weight_broadcast_ms": round(broadcast_ms, 2) if world > 1 else None,
                          "records_sha256": rec_sha, "rank_blocks_found_in_gathered_records": blocks_ok,
                          "rccl_selfcheck_world1": rccl_check},
        }
        if dom is not None:
            dom_traffic = None
            if traffic is not None:
                hit = traffic["kernels"].get(dom["name"]) or next((v for k, v in traffic["kernels"].items() if k.startswith(dom["name"].rstrip("<("))), None)
                dom_traffic = int(hit["hbm_mb_per_launch"] * 1e6) if hit else None
            # the lower of the two roofs at this kernel's arithmetic intensity: matrix pipe (its issue ceiling: peak / MFMA flops issued per algorithmic
            # flop) against HBM (algorithmic flops per algorithmic byte x 8 TB/s)
            ai = dom["algorithmic_gflop_per_launch"] * 1e9 / max(dom["algorithmic_mb_per_launch"] * 1e6, 1.0)
            hbm_roof_tf = ai * PEAK_HBM_GBS / 1e3
            mfma_roof_tf = dom["peak_tflops"] / dom["mfma_flops_issued_per_algorithmic_flop"]
            dom_bound = "hbm" if hbm_roof_tf < mfma_roof_tf else "mfma"
            line["roofline"] = {
                "kernel": dom["name"], "bound": dom_bound,
                "achieved": dom["achieved_tflops"], "peak": dom["peak_tflops"], "unit": "TFLOP/s", "frac": dom["frac"],
                "traffic": dom_traffic,
                "avg_launch_us": dom["avg_launch_us"], "launches_per_step": dom["launches_per_step"],
                "algorithmic_gflop_per_launch": dom["algorithmic_gflop_per_launch"],
                "share_of_timed_kernel_time": dom["share_of_timed_kernel_time"],
                "family": {"members": [m["name"] for m in dom_family],
                           "share_of_timed_kernel_time": round(sum(m["share_of_timed_kernel_time"] for m in dom_family), 4),
                           "launches_per_step": round(sum(m["launches_per_step"] for m in dom_family), 2),
                           "achieved_tflops": round(sum(m["algorithmic_gflop_per_launch"] * m["launches_per_step"] for m in dom_family)
                                                    / max(sum(m["avg_launch_us"] * m["launches_per_step"] for m in dom_family), 1e-9) * 1e3, 1),
                           "frac": round(sum(m["algorithmic_gflop_per_launch"] * m["launches_per_step"] for m in dom_family)
                                         / max(sum(m["avg_launch_us"] * m["launches_per_step"] for m in dom_family), 1e-9) * 1e3 / dom["peak_tflops"], 4),
                           "note": "all template instantiations of the dominant kernel together (achieved = their algorithmic flops / their launch time); "
                                   "`frac` above is the largest member's"},
                "runner_up": next(({"kernel": r_["name"], "share_of_timed_kernel_time": r_["share_of_timed_kernel_time"], "avg_launch_us": r_["avg_launch_us"],
                                    "achieved_tflops": r_["achieved_tflops"], "frac": r_["frac"]} for r_ in rows if r_ not in dom_family), None),
                "mfma_flops_issued_per_algorithmic_flop": dom["mfma_flops_issued_per_algorithmic_flop"],
                "frac_of_issue_ceiling": dom["frac_of_issue_ceiling"],
                "roofs_tflops": {"matrix_pipe_issue_ceiling": round(mfma_roof_tf, 1), "hbm_at_this_intensity": round(hbm_roof_tf, 1),
                                 "flops_per_byte": round(ai, 1)},
                "note": "achieved = algorithmic flops per launch (2 x M x N x K of the GEMMs inside the kernel) / average launch duration from HIP events recorded "
                        "around every launch on the launch stream in THIS run; peak = dense 16-bit MFMA.  The split-fp16 arithmetic issues 3 MFMA flops per "
                        "algorithmic flop (2 on the block tail's two-product level and in the attention input projections: mfma_flops_issued_per_algorithmic_flop), so "
                        "the kernel's own issue ceiling is peak / that factor (frac_of_issue_ceiling).  `kernel` is the name rocprofv3 prints "
                        "(profiles/*kernel_stats*.csv).  Measured in this run's single-stream pass (config.single_stream_ms_per_step): with sub-batch "
                        "streams two groups' launches run concurrently and a launch's elapsed time is no longer its own (tools/collect_profiles.sh "
                        "profiles the same single-stream configuration: --substreams 1)."}
        line["kernels"] = rows
        line["traffic"] = {"compulsory_mb_per_step": round(comp_mb, 1),
                           "measured_hbm_mb_per_step": traffic["hbm_mb_per_step"] if traffic else None,
                           "ratio": round(traffic["hbm_mb_per_step"] / comp_mb, 1) if traffic else None,
                           "per_kernel": traffic["kernels"] if traffic else None,
                           "method": traffic["method"] if traffic else f"not measured: {traffic_err}"}
        line["extra_configs"] = extras
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(sd, args.kpts)
        try:   # anything libraries left in the C-level stdout buffer (RCCL's banner) goes out BEFORE the line: the JSON line is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(line), flush=True)
    gdist.barrier()


if __name__ == "__main__":
    main()
