"""Developer: where the projection fused into the block tail (knob 32) departs from the separate k_qkv launch.
usage: python tools/dbg_fused_qkv.py [pairs] [kpts] [forced]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
forced = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sd = synthetic_state_dict(0)
pairs = [make_pair(400 + i, n_q=N - 31 * i, n_r=N - 17 * i) for i in range(B)]
eng = PoseEngine(0, max_batch=B, max_kpts=N, precision="f16x2_f16_attn", state_dict=sd)
inp = eng.stage_inputs(pairs)
T = 2 * B * N
if forced:
    for k, v in ((14, 128), (1, 70), (19, 2), (27, 2)):
        assert eng.lib.gn_debug_set_variant(eng.ctx, k, v) == 0
for lists in (1,):
    eng.lib.gn_debug_set_variant(eng.ctx, 31, lists)
    for stop_f, stop_s, what in ((4, 5, "self tail -> cross projection"), (6, 8, "cross tail -> self projection"), (8, 11, "second self tail -> cross projection")):
        got = {}
        for fused, stop in ((1, stop_f), (0, stop_s)):
            eng.lib.gn_debug_set_variant(eng.ctx, 32, fused)
            eng.lib.gn_debug_set_variant(eng.ctx, 4, stop)
            eng.set_kernel_timing(50)
            eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
            torch.cuda.synchronize()
            names = [(r["name"], int(r["launches"])) for r in eng.kernel_table()]
            eng.set_kernel_timing(0)
            got[fused] = (eng.debug_read("qkb", T * 256, np.uint32).copy(), eng.debug_read("vtb", T * 128, np.uint32).copy(),
                          eng.debug_read("x_p", T * 256, np.uint32).copy(), names)
        print("lists", lists, what)
        print("  fused   :", got[1][3])
        print("  separate:", got[0][3])
        nv = eng.debug_read("nvalid", 2 * B, np.int32)
        for nm, k, per in (("qkb", 0, 256), ("vtb", 1, 128), ("x_p", 2, 256)):
            a, b = got[1][k], got[0][k]
            d = a != b
            print("  %s: %d of %d words differ" % (nm, int(d.sum()), d.size))
            if nm != "vtb" and d.any():
                rows = np.nonzero(d.reshape(T, per).any(axis=1))[0]
                valid = [r for r in rows if (r % N) < nv[r // N]]
                print("     rows differing: %d (valid among them %d), first %s; columns of first row: %s" % (len(rows), len(valid), rows[:8], np.nonzero(d.reshape(T, per)[rows[0]])[0][:16]))
                if valid:
                    r = valid[0]
                    cols = np.nonzero(d.reshape(T, per)[r])[0][:8]
                    print("     first valid row", r, "cols", cols, "fused", a.reshape(T, per)[r][cols].view(np.float16), "separate", b.reshape(T, per)[r][cols].view(np.float16))
                    allrows = sorted(set(int(x) % 128 for x in rows)); allcols = sorted(set(int(x) for x in np.nonzero(d.reshape(T, per).any(axis=0))[0]))
                    print("     rows mod 128:", allrows[:64], "cols:", allcols[:64])
            if nm == "vtb" and d.any():
                dd = d.reshape(2 * B, 4, 64, N // 2)
                print("     slots", np.nonzero(dd.any(axis=(1, 2, 3)))[0], "heads", np.nonzero(dd.any(axis=(0, 2, 3)))[0], "dims", np.nonzero(dd.any(axis=(0, 1, 3)))[0][:8],
                      "key words", np.nonzero(dd.any(axis=(0, 1, 2)))[0][:16])
eng.lib.gn_debug_set_variant(eng.ctx, 4, 0)
for lists in (1, 2):
    eng.lib.gn_debug_set_variant(eng.ctx, 31, lists)
    for fused in (1, 0):
        eng.lib.gn_debug_set_variant(eng.ctx, 32, fused)
        idx, score, n = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        torch.cuda.synchronize()
        print("lists", lists, "fused", fused, "n_match", n.cpu().numpy(), "precision note:", eng.guard_status())
if os.environ.get("DBG_ROT"):
    eng.lib.gn_debug_set_variant(eng.ctx, 31, 1)
    eng.lib.gn_debug_set_variant(eng.ctx, 32, 1)
    eng.lib.gn_debug_set_variant(eng.ctx, 4, 6)
    eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    torch.cuda.synchronize()
    qk = eng.debug_read("qkb", T * 256, np.uint32).copy().view(np.float16).reshape(T, 512).astype(np.float32)
    rot = eng.debug_read("rot4", T * 64).reshape(16, T, 4)
    want = np.transpose(rot, (1, 0, 2)).reshape(T, 64).astype(np.float16).astype(np.float32)      # [token][4 fg + c]
    for nm, lo in (("q", 0), ("k", 256)):
        for h in range(4):
            got = qk[:, lo + 64 * h: lo + 64 * h + 64]
            bad = got != want
            rows = np.nonzero(bad.any(axis=1))[0]
            print("rot dump", nm, "head", h, ":", int(bad.sum()), "bad; rows mod 128", sorted(set(int(r) % 128 for r in rows))[:40], "cols", sorted(set(int(c) for c in np.nonzero(bad.any(axis=0))[0])))
            if bad.any() and h == 0:
                r = rows[0]; c = np.nonzero(bad[r])[0]
                print("   row", r, "cols", c, "got", got[r][c], "want", want[r][c])
                print("   got as bits", got[r][c].astype(np.float16).view(np.uint16))
    eng.lib.gn_debug_set_variant(eng.ctx, 4, 0)
