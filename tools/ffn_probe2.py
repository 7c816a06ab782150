"""Developer tool: residual stream after n layers, fused block tail vs three-launch form vs the oracle, on a RAGGED golden fixture."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import oracle_match  # noqa: E402
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402


def hm16_decode(raw_u32, rows, cols):
    h = raw_u32.view(np.float16).reshape(rows, cols // 16, 2, 16).astype(np.float32)
    return (h[:, :, 0, :] + h[:, :, 1, :]).reshape(rows, cols)


def main():
    sd = synthetic_state_dict(0)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    p = make_pair(7, n_q=96, n_r=80)
    taps = {}
    mq, mr, sc, oidx = oracle_match(tsd, p, taps=taps)
    eng = PoseEngine(0, max_batch=4, max_kpts=256, precision="f16x2_bf16_attn", state_dict=sd)
    inp = eng.stage_inputs([p])
    args = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    for nl in (1, 2, 9):
        eng.set_num_layers(nl)
        for mode in (0, 3):
            eng.lib.gn_debug_set_variant(eng.ctx, 10, mode)
            idx, score, n = eng.match(*args)
            torch.cuda.synchronize()
            x = eng.debug_read("x", 4 * 2 * 256 * 256).reshape(4, 2, 256, 256)
            e0 = np.abs(x[0, 0, :96] - taps[f"layer{nl - 1}_0"][0].numpy()).max() / np.abs(taps[f"layer{nl - 1}_0"][0].numpy()).max()
            e1 = np.abs(x[0, 1, :80] - taps[f"layer{nl - 1}_1"][0].numpy()).max() / np.abs(taps[f"layer{nl - 1}_1"][0].numpy()).max()
            best = min((float(np.abs(x[0, 0, :96] - taps[kk][0].numpy()).max()), kk) for kk in taps if kk.endswith("_0") and taps[kk].dim() == 3 and taps[kk].shape[1:] == (96, 256))
            print("   closest oracle tap to x side 0:", best, " x max", np.abs(x[0, 0, :96]).max(), "x_p decode err vs f32 x:",
                  np.abs(hm16_decode(eng.debug_read("x_p", 4 * 2 * 256 * 256, np.uint32), 4 * 2 * 256, 256).reshape(4, 2, 256, 256)[0, 0, :96] - x[0, 0, :96]).max())
            k = int(n[0])
            same = k == len(oidx) and np.array_equal(idx[0, :k].cpu().numpy(), oidx.numpy())
            print(f"layers {nl} mode {mode}: rel err side0 {e0:.3e} side1 {e1:.3e}  finite {np.isfinite(x[0]).all()}  matches {k} (oracle {len(oidx)}) identical {same}  guard {eng.guard_status()}")
            if nl == 9 and not same:
                a = {(int(q), int(r)) for q, r in idx[0, :k].cpu().numpy()}
                b = {(int(q), int(r)) for q, r in oidx.numpy()}
                print("   only gpu:", sorted(a - b), " only oracle:", sorted(b - a))
                print("   padded rows finite:", np.isfinite(x[0, 0, 96:]).all(), np.abs(x[0, 0, 96:]).max(), np.abs(x[0, 1, 80:]).max())


if __name__ == "__main__":
    main()
