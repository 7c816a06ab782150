"""Developer tool: what exactly differs in k_qkv<false>'s output (layer 0, self block) between runs of a library built with the SLP vectoriser.
  GISNAV_AMD_LIB=tools/probes/variants/lib_noslp.so python tools/slp_diag.py save gpurun_out/slp/ref.npy
  GISNAV_AMD_LIB=tools/probes/variants/lib_slp.so   python tools/slp_diag.py diff gpurun_out/slp/ref.npy 40
"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.synthetic import make_pair
from gisnav_amd.weights import synthetic_state_dict
mode, path = sys.argv[1], sys.argv[2]
stop = int(sys.argv[4]) if len(sys.argv) > 4 else 2
sd = synthetic_state_dict(0)
pairs = [make_pair(i) for i in range(32)]
T = 32 * 2 * 1024
eng = PoseEngine(0, max_batch=32, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=sd)
inp = eng.stage_inputs(pairs)
a = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
eng.set_num_layers(1)
eng.lib.gn_debug_set_variant(eng.ctx, 4, stop)
def run():
    try:
        eng.match(*a)
    except Exception:
        pass
    torch.cuda.synchronize()
    return eng.debug_read("qkb", T * 256).view(np.uint16).reshape(T, 512).copy()
def bf(u):
    return (u.astype(np.uint32) << 16).view(np.float32)
if mode == "save":
    q = run(); q2 = run()
    print("repeatable:", bool((q == q2).all()), "nonzero words:", int((q != 0).sum()))
    np.save(path, q)
else:
    ref = np.load(path)
    run(); rot4 = eng.debug_read("rot4", T * 64).reshape(16, T, 4)
    hist_f = np.zeros(512, int); hist_tok = np.zeros(128, int); nb = 0
    for r in range(int(sys.argv[3])):
        q = run()
        d = np.argwhere(q != ref)
        if len(d) == 0: continue
        nb += 1
        for (t, f) in d[:6]:
            p = f & ~3
            print(f"rep {r}: token {t} (slot {t // 1024}, row-in-tile {t % 128}, j {t % 128 // 32}, ql {t % 32}) feature {f} ({'q' if f < 256 else 'k'}, tile {f // 32}, in-tile {f % 32}, e {f % 4})"
                  f" got {bf(q[t, f:f+1])[0]:.6g} want {bf(ref[t, f:f+1])[0]:.6g};  group want {bf(ref[t, p:p+4])} got {bf(q[t, p:p+4])}"
                  f" rot4 {rot4[(f % 64) >> 2, t]}")
            c, s_ = rot4[(f % 64) >> 2, t, 1], rot4[(f % 64) >> 2, t, 3]
            zw = bf(ref[t, p + 2:p + 4]).astype(np.float64); vz = zw[0] * c + zw[1] * s_; vw = zw[1] * c - zw[0] * s_   # un-rotate (c^2 + s^2 = 1)
            c0, s0 = rot4[(f % 64) >> 2, t, 0], rot4[(f % 64) >> 2, t, 2]
            print(f"      v.z {vz:.6g} v.w {vw:.6g}: z with +v.w: {vz * c + vw * s_:.6g}; v.z c only {vz * c:.6g}; -v.w s only {-vw * s_:.6g}; with cos of the x/y pair {vz * c0 - vw * s_:.6g}; with sin of the x/y pair {vz * c - vw * s0:.6g}; both {vz * c0 - vw * s0:.6g}")
        for (t, f) in d: hist_f[f] += 1; hist_tok[t % 128] += 1
        print(f"rep {r}: {len(d)} elements differ")
    print(f"{nb} of {sys.argv[3]} runs differ;  by e = feature % 4: {[int(hist_f[e::4].sum()) for e in range(4)]};  q vs k: {int(hist_f[:256].sum())} {int(hist_f[256:].sum())}")
    print("by tile:", [int(hist_f[32 * i:32 * i + 32].sum()) for i in range(16)])
    print("by in-tile group g (of 8 features) :", [int(hist_f.reshape(16, 4, 8)[:, g, :].sum()) for g in range(4)], " by half (hh):", [int(hist_f.reshape(64, 2, 4)[:, h, :].sum()) for h in range(2)])
    print("by j:", [int(hist_tok[32 * j:32 * j + 32].sum()) for j in range(4)])
