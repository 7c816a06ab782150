"""Developer tool: N LoFTR forwards (for rocprofv3 --kernel-trace --stats).  python tools/loftr_profile.py [exact_f32|split_fp16] [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd import loftr_synthetic as olf
from gisnav_amd.loftr import LoFTR
arith = sys.argv[1] if len(sys.argv) > 1 else "split_fp16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
m = LoFTR(state_dict=olf.synthetic_state_dict(0), fine=True, graph=False, arithmetic=arith).to(dev).eval()
i0, i1 = olf.synthetic_pair(1, 480, 640)
data = {"image0": i0.to(dev), "image1": i1.to(dev)}
for _ in range(n):
    out = m(data)
torch.cuda.synchronize()
print("matches", int(out["keypoints0"].shape[0]))
