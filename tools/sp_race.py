"""Developer tool: run-to-run differences of the SuperPoint layer outputs, layer by layer (knob 39 stops the extractor behind a layer)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f16x2_f16_attn", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0))
rng = np.random.default_rng(0)
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
img = torch.from_numpy(rng.random((1, H, W), dtype=np.float32)).cuda()
layers = {1: ("sp_y", 2, 64, (4, 16)), 2: ("sp_x", 2, 64, (8, 32)), 3: ("sp_y", 4, 64, (4, 16)), 4: ("sp_x", 4, 128, (8, 32)), 5: ("sp_y", 8, 128, (4, 16)),
          6: ("sp_x", 8, 128, (8, 32)), 7: ("sp_y", 8, 128, (8, 32)), 8: ("sp_x", 8, 256, (8, 32)), 10: ("sp_x", 8, 256, (8, 32))}
for stop, (name, div, C, (th, tw)) in layers.items():
    oh, ow = H // div, W // div
    eng.lib.gn_debug_set_variant(eng.ctx, 39, stop)
    outs = []
    for rep in range(reps):
        try:
            sp.detect_and_describe_device(img)
        except Exception:
            pass      # (the skipped layers leave garbage for the detector: candidate overflow)
        torch.cuda.synchronize()
        outs.append(eng.debug_read(name, oh * ow * C).view(np.uint32).reshape(oh, ow, C).copy())
    bad = [r for r in range(1, reps) if not np.array_equal(outs[0], outs[r])]
    print(f"layer {stop} ({oh} x {ow} x {C}): {len(bad)} of {reps - 1} repeats differ from run 0", flush=True)
    for rep in bad[:2]:
        d = np.argwhere(outs[0] != outs[rep])
        ys, xs, cs = d[:, 0], d[:, 1], d[:, 2]
        tiles = sorted(set(zip((ys // th).tolist(), (xs // tw).tolist())))
        print(f"   rep {rep}: {len(d)} words; tiles {tiles[:6]} n = {len(tiles)}; rows in tile {sorted(set((ys % th).tolist()))} cols in tile {sorted(set((xs % tw).tolist()))[:40]} words {sorted(set(cs.tolist()))[:48]}")
        for (y, x, c) in d[:4].tolist():
            print(f"     (y {y}, x {x}, word {c}): run0 {int(outs[0][y, x, c]):08x} run{rep} {int(outs[rep][y, x, c]):08x}")
    if bad:
        break
eng.lib.gn_debug_set_variant(eng.ctx, 39, 0)
