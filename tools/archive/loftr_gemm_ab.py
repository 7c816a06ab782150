import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd import loftr_synthetic as olf
from gisnav_amd.loftr import LoFTR
from gisnav_amd.engine import PoseEngine
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f32")      # a gn_ctx to reach the process-wide developer knob
dev = torch.device("cuda", 0)
m = LoFTR(state_dict=olf.synthetic_state_dict(0), fine=True, graph=True, arithmetic="exact_f32").to(dev).eval()
i0, i1 = olf.synthetic_pair(1, 480, 640)
data = {"image0": i0.to(dev), "image1": i1.to(dev)}
for thr in (0, 170, 320, 0, 170, 320):
    eng.lib.gn_debug_set_variant(eng.ctx, 41, thr)
    m2 = LoFTR(state_dict=olf.synthetic_state_dict(0), fine=True, graph=True, arithmetic="exact_f32").to(dev).eval()   # (a new context: the graph is captured with the knob in force)
    for _ in range(3): out = m2(data)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = m2(data)
    torch.cuda.synchronize()
    print(f"m64 threshold {thr}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per pair, matches {int(out['keypoints0'].shape[0])}", flush=True)
    del m2
