"""Developer tool: random ragged batches (keypoint counts anywhere in [2, max_kpts] per side, mixed inside one batch) through the C ABI against the
oracle: correspondence indices must be identical in f32 mode; the fast mode is reported.   python tests/sweeps/fuzz_ragged.py [trials] [max_kpts]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import oracle_match  # noqa: E402   (the oracle is the checker here, as in tests/)
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
K = int(sys.argv[2]) if len(sys.argv) > 2 else 384
sd = synthetic_state_dict(0)
sd_t = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
rng = np.random.default_rng(2026)
engs = {p: PoseEngine(0, max_batch=4, max_kpts=K, precision=p, state_dict=sd) for p in ("f32", "f16x2_f16_attn")}
bad = {p: 0 for p in engs}; tot = {p: 0 for p in engs}; worst_pose = 0.0
for t in range(trials):
    sizes = [(int(rng.integers(2, K + 1)), int(rng.integers(2, K + 1))) for _ in range(int(rng.integers(1, 5)))]
    if t % 3 == 0: sizes[0] = (K, K)
    if t % 4 == 1: sizes[-1] = (int(rng.integers(2, 20)), K)
    pairs = [make_pair(5000 + 10 * t + i, n_q=a, n_r=b) for i, (a, b) in enumerate(sizes)]
    want = [oracle_match(sd_t, p)[3].numpy() for p in pairs]
    for prec, eng in engs.items():
        inp = eng.stage_inputs(pairs)
        idx, score, nm = eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        out = eng.estimate(inp, K_MATRIX)
        torch.cuda.synchronize()
        assert torch.isfinite(out["R"]).all() and torch.isfinite(out["t"]).all()
        for b, w in enumerate(want):
            got = idx[b, : int(nm[b])].cpu().numpy()
            tot[prec] += 1
            if len(got) != len(w) or not np.array_equal(got, w):
                bad[prec] += 1
                print(f"trial {t} pair {b} sizes {sizes[b]} {prec}: {len(got)} matches vs oracle {len(w)}; first difference at "
                      f"{next((i for i in range(min(len(got), len(w))) if not np.array_equal(got[i], w[i])), min(len(got), len(w)))}", flush=True)
    print(f"trial {t}: sizes {sizes} ok", flush=True)
print("pairs with any index difference:", {p: f"{bad[p]} of {tot[p]}" for p in engs})
sys.exit(1 if bad["f32"] else 0)
