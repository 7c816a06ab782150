"""Developer micro-benchmark: f32 MFMA GEMM variants on the matcher's shapes + pure-MFMA ceiling probe."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n  # ms


def main():
    dev = torch.device("cuda", 0)
    eng = PoseEngine(0, max_batch=1, max_kpts=128)
    lib, ctx = eng.lib, eng.ctx
    for blocks in (512, 1024):
        for sign, label in ((1, "constant operands"), (-1, "random operands")):
            iters = 4000
            ms = timeit(lambda: lib.gn_debug_mfma_probe(ctx, blocks, sign * iters, eng._stream()), n=5, warm=2)
            fl = blocks * 4 * iters * 8 * 2 * 32 * 32 * 2
            print(f"mfma probe blocks={blocks} {label}: {fl / ms / 1e9:.1f} TF", flush=True)
    M = 65536
    shapes = [(256, 128), (768, 256), (256, 256), (512, 256), (512, 512), (256, 512), (1024, 256), (512, 2048)]
    variants = [int(v) for v in (sys.argv[1:] or ["3", "5", "6", "7"])]   # 3 f32 MFMA, 5 f32x3, 6 f16x2 (on-the-fly split), 7 k_gemm_p2
    for variant in variants:
        out_mode = 0
        lib.gn_debug_set_variant(ctx, 8, 1)
        if variant >= 700:         # 70x: k_gemm_p2w timing ablations (g_p2_wide = x)
            lib.gn_debug_set_variant(ctx, 8, variant - 700); variant = 7
        elif variant >= 70:          # 71: k_gemm_p2 with hm16 output only, 72: f32 + hm16 output
            out_mode, variant = variant - 70, 7
        lib.gn_debug_set_variant(ctx, 7, out_mode)
        lib.gn_debug_set_variant(ctx, 0, variant)
        lib.gn_debug_set_variant(ctx, 2, 1 if variant >= 5 else 0)
        tot_ms = tot_fl = 0
        for N, K in shapes:
            A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
            if variant == 7:      # build the operand planes once, then time the GEMM alone
                lib.gn_debug_set_variant(ctx, 6, 0); eng.debug_gemm(A, W, b); lib.gn_debug_set_variant(ctx, 6, 1)
            ms = timeit(lambda: eng.debug_gemm(A, W, b))
            lib.gn_debug_set_variant(ctx, 6, 0)
            fl = 2.0 * M * N * K
            tot_ms += ms; tot_fl += fl
            print(f"variant {variant} M={M} N={N} K={K}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TF", flush=True)
        print(f"variant {variant} out_mode {out_mode} aggregate {tot_fl / tot_ms / 1e9:.1f} TF", flush=True)
    lib.gn_debug_set_variant(ctx, 7, 0)
    variants = [7 if v >= 70 else v for v in variants]
    for (M_, N_, K_) in ((1024, 384, 512), (65536, 512, 512), (65536, 256, 256)):
        A = torch.randn(M_, K_, device=dev); W = torch.randn(N_, K_, device=dev) * 0.05; b = torch.randn(N_, device=dev)
        ref = A.double() @ W.double().T + b.double()
        for v in sorted(set(variants)):
            for planes in ((0, 1) if v == 5 else (0, 1, 14) if v == 6 else (1, 14) if v == 7 else (0,)):
                lib.gn_debug_set_variant(ctx, 0, v); lib.gn_debug_set_variant(ctx, 2, planes)
                errs = []
                for rep in range(3):
                    y = eng.debug_gemm(A, W, b)
                    errs.append(float((y.double() - ref).abs().max() / ref.abs().max()))
                print(f"variant {v} planes={planes} M={M_} N={N_} K={K_}: max abs err vs fp64 / max|ref| = {errs}", flush=True)
        lib.gn_debug_set_variant(ctx, 2, 0)


if __name__ == "__main__":
    main()
