"""Developer tool: SuperPoint extractor with the first convolution fused into the second (knob 46 = 1) against the two-launch form (0), 4 x 1080p per call, interleaved."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f16x2_f16_attn", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0))
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((4, 1080, 1920), dtype=np.float32)).cuda()
for knob in (0, 1, 0, 1, 0, 1):
    eng.lib.gn_debug_set_variant(eng.ctx, 46, knob)
    for _ in range(2):
        sp.detect_and_describe_device(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        sp.detect_and_describe_device(img)
    torch.cuda.synchronize()
    print(f"knob 46 = {knob}: {(time.perf_counter() - t0) / 8 / 4 * 1e3:.3f} ms per 1080p image", flush=True)
eng.lib.gn_debug_set_variant(eng.ctx, 46, 1)
