import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f16x2_bf16_attn", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0))
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((4, 1080, 1920), dtype=np.float32)).cuda()
ref = None
for knob in (1, 2, 0, 1, 2):
    eng.lib.gn_debug_set_variant(eng.ctx, 21, knob)
    for _ in range(2):
        out = sp.detect_and_describe_device(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        out = sp.detect_and_describe_device(img)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 / 4 * 1e3
    kpt, score, desc, n = out
    k = kpt[0, : n[0]].cpu().numpy(); sc = score[0, : n[0]].cpu().numpy(); d = desc[0, : n[0]].cpu().numpy()
    if knob == 0: ref = (k, sc, d)
    print(f"knob 21 = {knob}: {dt:.3f} ms per 1080p image, n = {list(n)}")
    if ref is not None and knob != 0:
        a = {(float(x), float(y)) for x, y in ref[0][:, :2]}; b = {(float(x), float(y)) for x, y in k[:, :2]}
        print("   keypoints in common with exact f32:", len(a & b), "of", len(a))
