"""Developer tool: LoFTR per 640x480 pair with and without the fine level, both arithmetics (the figures INTEGRATION.md quotes).   python tools/loftr_fine_vs_coarse.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd import loftr_synthetic as olf
from gisnav_amd.loftr import LoFTR
dev = torch.device("cuda", 0)
i0, i1 = olf.synthetic_pair(1, 480, 640)
data = {"image0": i0.to(dev), "image1": i1.to(dev)}
for arith in ("exact_f32", "split_fp16"):
    for fine in (True, False):
        m = LoFTR(state_dict=olf.synthetic_state_dict(0), fine=fine, graph=True, arithmetic=arith).to(dev).eval()
        for _ in range(3): out = m(data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): out = m(data)
        torch.cuda.synchronize()
        print(arith, "fine" if fine else "coarse only", f"{(time.perf_counter() - t0) / 10 * 1e3:.2f} ms", int(out["keypoints0"].shape[0]), flush=True)
        del m
