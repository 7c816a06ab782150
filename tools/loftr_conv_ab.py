"""Developer A/B (round 5): LoFTR's 3 x 3 convolutions with the next slice's halo tile prefetched into registers (knob 42 bit 0 = 0, shipped) against
the staging pass of rounds 3-4 (bit 0 = 1), and the overhead term of lf_conv's rows-per-wave cost model (knob 42 bits 8..: x 100).  Same process, interleaved.
   python tools/loftr_conv_ab.py [exact_f32|split_fp16]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd import loftr_synthetic as olf
from gisnav_amd.loftr import LoFTR
from gisnav_amd.engine import PoseEngine
arith = sys.argv[1] if len(sys.argv) > 1 else "exact_f32"
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f32")      # a gn_ctx to reach the process-wide developer knob
dev = torch.device("cuda", 0)
i0, i1 = olf.synthetic_pair(1, 480, 640)
data = {"image0": i0.to(dev), "image1": i1.to(dev)}
ref = None
knobs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 0, (25 << 8), (50 << 8), (100 << 8), 1, 0, (25 << 8), (50 << 8), (100 << 8)]
extra = [tuple(int(x) for x in a.split(":")) for a in sys.argv[3:]]      # further process-wide knobs, which:value (e.g. 44:0 = the 64-row GEMM without its ring)
for w, v in extra:
    eng.lib.gn_debug_set_variant(eng.ctx, w, v)
for knob in knobs:
    eng.lib.gn_debug_set_variant(eng.ctx, 42, knob)
    m2 = LoFTR(state_dict=olf.synthetic_state_dict(0), fine=True, graph=True, arithmetic=arith).to(dev).eval()   # (a new context: the graph is captured with the knob in force)
    for _ in range(3): out = m2(data)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = m2(data)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    res = tuple(out[k].cpu() for k in ("keypoints0", "keypoints1", "confidence"))
    same = "-" if ref is None else str(all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(ref, res)))
    if ref is None: ref = res
    import hashlib
    digest = hashlib.sha256(b"".join(t.numpy().tobytes() for t in res)).hexdigest()[:12]
    print(f"[outputs sha256 {digest}] knob42 {knob:6d} (prefetch {'off' if knob & 1 else 'on'}, overhead {(knob >> 8) * 0.01 if knob >> 8 else (0.75 if knob & 1 else 0.25):.2f}): {dt:.3f} ms per pair, matches {int(res[0].shape[0])}, bitwise equal to the first variant: {same}", flush=True)
    del m2
