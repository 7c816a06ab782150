import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
dev = torch.device("cuda", 0)
eng = PoseEngine(0, max_batch=1, max_kpts=128)
lib, ctx = eng.lib, eng.ctx
torch.manual_seed(0)
for variant in (3, 5):
    for planes in ((0, 1) if variant == 5 else (0,)):
        lib.gn_debug_set_variant(ctx, 0, variant); lib.gn_debug_set_variant(ctx, 2, planes)
        for (M, N, K) in ((65536, 768, 256), (65536, 256, 512), (65536, 512, 512), (65536, 256, 128)):
            A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
            y0 = eng.debug_gemm(A, W, b).clone()
            nbad = 0; worst = 0.0
            for rep in range(30):
                y = eng.debug_gemm(A, W, b)
                d = (y - y0).abs().max().item()
                if d != 0.0:
                    nbad += 1; worst = max(worst, d)
            print(f"variant {variant} planes {planes} M={M} N={N} K={K}: non-identical reps {nbad}/30 worst {worst:.3e}", flush=True)
