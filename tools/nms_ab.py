import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f16x2_f16_attn", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0))
rng = np.random.default_rng(0)
for shape in ((2, 1080, 1920), (1, 480, 640), (3, 136, 200)):
    img = torch.from_numpy(rng.random(shape, dtype=np.float32)).cuda()
    res = {}
    for knob in (0, 1):
        eng.lib.gn_debug_set_variant(eng.ctx, 36, knob)
        out = sp.detect_and_describe_device(img)
        torch.cuda.synchronize()
        nms = eng.debug_read("sp_nms", shape[0] * shape[1] * shape[2]).copy()
        res[knob] = (nms, [o.cpu().numpy().copy() if hasattr(o, "cpu") else np.array(o) for o in out])
    same = np.array_equal(res[0][0], res[1][0])
    print(shape, "nms maps identical:", same, "nonzero:", int((res[0][0] != 0).sum()), "outputs identical:", all(np.array_equal(a, b) for a, b in zip(res[0][1], res[1][1])))
