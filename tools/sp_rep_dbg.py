import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f16x2_f16_attn", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0))
shape = (2, 1080, 1920)
cap = max(16384, shape[1] * shape[2] // 16)
img = torch.from_numpy(np.random.default_rng(11).random(shape, dtype=np.float32)).cuda()
eng.lib.gn_debug_set_variant(eng.ctx, 34, 1)
ref = None
for rep in range(16):
    out = sp.detect_and_describe_device(img); torch.cuda.synchronize()
    counts = eng.debug_read("sp_counts", 8).view(np.int32).reshape(2, 4).copy()
    cand = eng.debug_read("sp_cand", 2 * cap * 2).view(np.int32).reshape(2, cap, 2).copy()
    nms = eng.debug_read("sp_nms", shape[0] * shape[1] * shape[2]).reshape(shape).copy()
    sets = []
    for b in range(2):
        n = int(counts[b, 0])
        c = cand[b, :n]
        order = np.lexsort((c[:, 1], c[:, 0]))
        sets.append(c[order])
    kp = out[0].cpu().numpy().copy()
    expect = [int(((nms[b] > 0.005) & (np.arange(shape[1])[:, None] >= 4) & (np.arange(shape[2])[None, :] >= 4)).sum()) for b in range(2)]
    cur = (counts.copy(), sets, kp)
    if ref is None: ref = cur
    same_c = [np.array_equal(ref[1][b], sets[b]) for b in range(2)]
    print(f"rep {rep}: counts {counts[:, :3].tolist()} expected candidates {expect}; candidate sets equal to run 0: {same_c}; keypoints equal: {np.array_equal(ref[2], kp)}; duplicates in list: {[int(len(s) - len(np.unique(s[:, 0]))) for s in sets]}", flush=True)
