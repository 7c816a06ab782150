"""How much evidence is behind "GN_PREC_F32 gives the oracle's correspondence indices" -- and behind the certificate.

f32 MFMA summation order differs from torch-CPU's, so exactness of the f32 mode against the CPU restatement is EMPIRICAL.  This sweep runs, for
`seeds` different weight sets of each discriminating family (low-margin, mid-margin, PyTorch-default initialisation), 16 pairs x 1024 keypoints
through (a) the exact-f32 mode and (b) the headline mode with the calibrated certificate in re-run mode, each against the oracle, and counts
matches / index mismatches / flagged pairs.  Writes gpurun_out/f32_sweep_r06.json and merges the totals into gpurun_out/parity_r06.json.

    python tools/f32_exactness_sweep.py [seeds=11] [pairs=16] [block-tail products=2]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import oracle_match  # noqa: E402
from gisnav_amd import _lib  # noqa: E402
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import default_init_state_dict, synthetic_state_dict  # noqa: E402

LOW = dict(ffn_out_std=4.8e-3, final_scale=4.0, matchability_bias=0.0, matchability_std=0.05)
MID = dict(ffn_out_std=1.2e-3, final_scale=12.0, matchability_bias=2.0, matchability_std=0.05)
FAMILIES = {"low_margin": (lambda s: synthetic_state_dict(s, **LOW), 0.0), "mid_margin": (lambda s: synthetic_state_dict(s, **MID), 0.01),
            "default_init": (lambda s: default_init_state_dict(s), 0.0)}


def diff(idx, n, refs, flags=None):
    """symmetric difference of the match sets, summed over the pairs (and, given per-pair flags, the part of it in UNFLAGGED pairs)"""
    tot = unflagged = 0
    for b, r in enumerate(refs):
        a = {(int(q), int(c)) for q, c in idx[b, : int(n[b])]}
        d = len(a ^ {(int(q), int(c)) for q, c in r})
        tot += d
        if flags is not None and not flags[b]:
            unflagged += d
    return tot if flags is None else (tot, unflagged)


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 11
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    products = int(sys.argv[3]) if len(sys.argv) > 3 else 2          # block tail of the headline engine (gn_set_ffn_products)
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    out = {"source_digest": _lib.library_digest(), "pairs_per_seed": B, "keypoints_per_side": 1024, "headline_block_tail_partial_products": products, "families": {}}
    t0 = time.time()
    for fam, (make, th) in FAMILIES.items():
        tot = {"seeds": 0, "pairs": 0, "cpu_matches": 0, "f32_index_mismatches": 0, "f32_pairs_with_a_decision_within_1e-4": 0, "f32_index_mismatches_in_pairs_without_such_a_decision": 0,
               "headline_uncertified_index_mismatches": 0, "headline_certified_index_mismatches": 0, "headline_pairs_flagged": 0, "eps": []}
        for s in range(seeds if fam == "low_margin" else max(2, seeds // 3)):
            sd = make(s)
            tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
            pairs = [make_pair(20_000 + 100 * s + i, n_q=1024 - 7 * (i % 4), n_r=1024 - 11 * (i % 3)) for i in range(B)]
            refs = [oracle_match(tsd, p, filter_threshold=th)[3].numpy() for p in pairs]
            e32 = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f32", state_dict=sd, filter_threshold=th)
            inp = e32.stage_inputs(pairs)
            e32.set_certify("flag")
            idx, _, n = e32.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
            torch.cuda.synchronize()
            f32_flags = e32.uncertain(B)
            d_all, d_unflagged = diff(idx.cpu().numpy(), n.cpu().numpy(), refs, f32_flags)
            tot["f32_index_mismatches"] += d_all
            tot["f32_index_mismatches_in_pairs_without_such_a_decision"] += d_unflagged
            tot["f32_pairs_with_a_decision_within_1e-4"] += int((f32_flags != 0).sum())
            del e32
            eng = PoseEngine(0, max_batch=B, max_kpts=1024, precision="f16x2_f16_attn", state_dict=sd, filter_threshold=th)
            eng.set_ffn_products(products)
            cal = eng.calibrate_certify(eng.stage_inputs([make_pair(30_000 + 100 * s + i, n_q=1024, n_r=1000) for i in range(B)]), safety=4.0)
            inp = eng.stage_inputs(pairs)
            eng.set_certify("flag")
            idx, _, n = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
            torch.cuda.synchronize()
            tot["headline_uncertified_index_mismatches"] += diff(idx.cpu().numpy(), n.cpu().numpy(), refs)
            tot["headline_pairs_flagged"] += int((eng.uncertain(B) != 0).sum())
            eng.set_certify("rerun")
            idx, _, n = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
            torch.cuda.synchronize()
            tot["headline_certified_index_mismatches"] += diff(idx.cpu().numpy(), n.cpu().numpy(), refs)
            del eng
            tot["eps"].append(round(cal["eps"], 6))
            tot["seeds"] += 1; tot["pairs"] += B; tot["cpu_matches"] += sum(len(r) for r in refs)
            print(fam, "seed", s, {k: v for k, v in tot.items() if k != "eps"}, f"{time.time() - t0:.0f} s", flush=True)
        out["families"][fam] = tot
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    sfx = "_tail2" if products == 2 else ""       # (a second run on the two-product block tail adds its own keys; the f32 columns repeat)
    with open(os.path.join(ROOT, "gpurun_out", f"f32_sweep_r06{sfx}.json"), "w") as f:
        json.dump(out, f, indent=1)
    path = os.path.join(ROOT, "gpurun_out", "parity_r06.json")
    rep = {}
    if os.path.exists(path):
        with open(path) as f:
            rep = json.load(f)
    if rep.get("source_digest") != out["source_digest"]:
        rep = {"source_digest": out["source_digest"]}
    for fam, tot in out["families"].items():
        if not sfx:
            rep[f"f32_sweep_16x1024_{fam}"] = {k: v for k, v in tot.items() if k.startswith(("f32_", "cpu_", "seeds", "pairs"))}
        rep[f"certified_sweep_16x1024_{fam}{sfx}"] = {"cpu_matches": tot["cpu_matches"], "uncertified_index_mismatches": tot["headline_uncertified_index_mismatches"],
                                                 "certified_index_mismatches": tot["headline_certified_index_mismatches"], "pairs": tot["pairs"],
                                                 "pairs_flagged": tot["headline_pairs_flagged"], "rerun_fraction": round(tot["headline_pairs_flagged"] / max(tot["pairs"], 1), 4),
                                                 "eps": f"{min(tot['eps'])} .. {max(tot['eps'])} (calibrated per weight set)", "safety": 4.0, "block_tail_partial_products": products}
    with open(path, "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)
    # what must hold: no mismatch of the f32 mode in a pair its own certificate passed; the certified headline mode never worse than the f32 mode it re-runs in
    bad = sum(t["f32_index_mismatches_in_pairs_without_such_a_decision"] + max(0, t["headline_certified_index_mismatches"] - t["f32_index_mismatches"]) for t in out["families"].values())
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "eps"} for k, v in out["families"].items()}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
