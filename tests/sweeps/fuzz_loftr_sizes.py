"""Developer tool: LoFTR at a sweep of image sizes (multiples of 8, non-square, tiny to mid) against its oracle, both arithmetics: coarse indices
must be identical, confidences / fine keypoints within the test-suite bars.   python tests/sweeps/fuzz_loftr_sizes.py [HxW ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import loftr as lf  # noqa: E402   (checker, as in tests/)
from gisnav_amd.loftr import LoFTR  # noqa: E402
torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
sd = lf.synthetic_state_dict(0)
sizes = [(32, 32), (40, 72), (64, 48), (88, 136), (104, 104), (120, 248), (168, 96), (200, 312), (256, 256), (72, 400)]
if len(sys.argv) > 1:      # e.g. 960x1280 1080x1920: the large end (the coarse similarity matrix is (H W / 64)^2 floats)
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
bad = 0
for k, (h, w) in enumerate(sizes):
    i0, i1 = lf.synthetic_pair(10 + k, h, w)
    ref = lf.loftr_forward(sd, i0, i1)
    for arith in ("exact_f32", "split_fp16"):
        m = LoFTR(state_dict=sd, arithmetic=arith).to("cuda:0").eval()
        out = m({"image0": i0[None, None].cuda(), "image1": i1[None, None].cuda()}, with_ids=True)
        same = torch.equal(out["i_ids"].cpu(), ref["i_ids"]) and torch.equal(out["j_ids"].cpu(), ref["j_ids"])
        n = len(ref["i_ids"])
        dc = float((out["confidence"].cpu() - ref["confidence"]).abs().max()) if same and n else float("nan")
        dk = float((out["keypoints1"].cpu() - ref["keypoints1"]).abs().max()) if same and n else float("nan")
        ok = same and (n == 0 or (dc < 2e-4 and dk < 2e-3))
        bad += not ok
        print(f"{h}x{w} {arith}: {n} oracle matches, {len(out['i_ids'])} here, indices identical {same}, |dconf| {dc:.2e}, |dkpt1| {dk:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
        del m
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
