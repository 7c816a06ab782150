"""Developer tool: phase stamps (s_memtime) of k_sp_conv_s (knob 34 = 1: the 32-channel-slice form carries the stamps) for SuperPoint layers, 4 x 1080p per call.   python tools/sp_phases.py [layers...]"""
import sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp

layers = [int(x) for x in sys.argv[1:]] or [1, 2, 5, 7]
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f16x2_f16_attn", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0))
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((4, 1080, 1920), dtype=np.float32)).cuda()
eng.lib.gn_debug_set_variant(eng.ctx, 34, 1)
for _ in range(2):
    sp.detect_and_describe_device(img)
torch.cuda.synchronize()
for layer in layers:
    eng.lib.gn_debug_set_variant(eng.ctx, 35, layer)
    sp.detect_and_describe_device(img)
    torch.cuda.synchronize()
    ts = eng.debug_read("sp_ts", 8192 * 32 * 2).view(np.int64).reshape(8192, 32)
    ts = ts[ts[:, 0] != 0]
    t0 = ts[:, 0:1]
    d = (ts - t0).astype(np.float64)
    names = ["start", "dma issued 0", "staged 0"] + [f"tap {t}" for t in range(9)] + ["dma issued 1", "staged 1"] + [f"tap {t}'" for t in range(9)] + ["", "loops done", "end"]
    print(f"layer {layer}: {len(ts)} workgroups stamped; launch span {(ts[:, 25].max() - ts[:, 0].min())} cycles; median cycles since the workgroup's start / since the previous stamp")
    prev = np.zeros(len(ts))
    for k, nm in ((26, "epi barrier"), (27, "row 0 in LDS"), (28, "row 0 stored"), (29, "row 1 stored")):
        if (ts[:, k] != 0).all(): print(f"  {nm:14s} {np.median(d[:, k]):9.0f}   (since loops done: {np.median(d[:, k] - d[:, 24]):8.0f})")
    for k in range(26):
        if k < len(names) and names[k] and (ts[:, k] != 0).all():
            cur = d[:, k]
            print(f"  {names[k]:14s} {np.median(cur):9.0f} {np.median(cur - prev):8.0f}   (p10 {np.percentile(cur - prev, 10):7.0f}, p90 {np.percentile(cur - prev, 90):7.0f})")
            prev = cur
eng.lib.gn_debug_set_variant(eng.ctx, 35, 0)
eng.lib.gn_debug_set_variant(eng.ctx, 34, 2)
