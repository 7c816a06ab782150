"""Developer tool: run-to-run repeatability of the SuperPoint pass per convolution family (knob 34), several frame sizes."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f16x2_f16_attn", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0))
rng = np.random.default_rng(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
def check(shape, tag):
    img = torch.from_numpy(rng.random(shape, dtype=np.float32)).cuda()
    maps = []
    for rep in range(reps):
        sp.detect_and_describe_device(img)
        torch.cuda.synchronize()
        maps.append((eng.debug_read("sp_enc", shape[0] * (shape[1] // 8) * (shape[2] // 8) * 128).copy(), eng.debug_read("sp_nms", shape[0] * shape[1] * shape[2]).copy()))
    print(tag, shape, "runs differing from run 0 (encoder / nms map):", sum(1 for m in maps[1:] if not np.array_equal(maps[0][0], m[0])), sum(1 for m in maps[1:] if not np.array_equal(maps[0][1], m[1])), "of", reps - 1, flush=True)
for knob in (2, 1):
    eng.lib.gn_debug_set_variant(eng.ctx, 34, knob)
    check((1, 480, 1920), f"knob 34 = {knob}")
    check((4, 1080, 1920), f"knob 34 = {knob}")
    check((2, 1080, 640), f"knob 34 = {knob}")
