"""Developer tool: latency of small batches (1, 2, 4 pairs; the reference's operating point is 1) in the headline precision, timed like bench.py's extras.
   python tools/small_batch.py [batch ...] [--knob WHICH:VALUE ...] [--precision f32] [--table]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
dev = torch.device("cuda", 0)
sd = synthetic_state_dict(0)
args = sys.argv[1:]
knobs = [tuple(int(x) for x in a.split(":")) for i, a in enumerate(args) if i > 0 and args[i - 1] == "--knob"]
prec = next((args[i + 1] for i, a in enumerate(args) if a == "--precision"), "f16x2_f16_attn")
batches = [int(a) for i, a in enumerate(args) if a.isdigit() and (i == 0 or args[i - 1] != "--knob")] or [1, 2, 4]
for b in batches:
    eng = PoseEngine(0, max_batch=b, max_kpts=1024, precision=prec, state_dict=sd)
    for w, v in knobs:
        eng.lib.gn_debug_set_variant(eng.ctx, w, v)
    if "--overlap" in args:
        eng.set_overlap(True)
    inp = eng.stage_inputs([make_pair(i, n_q=1024, n_r=1024) for i in range(b)])
    out = eng.alloc_outputs(b)
    torch.cuda.synchronize()
    elapsed, _ = bench.timed_steps(eng, inp, out, 300, 30, dev)
    print(f"batch {b} knobs {knobs}: {elapsed / 300 * 1e3:.4f} ms per call, {b * 300 / elapsed:.1f} pairs/s, poses ok {int(out['ok'].sum().item())}", flush=True)
    del eng
    if "--table" in args:      # per-kernel HIP-event table of one call (launch gaps excluded) + the stage split
        eng = PoseEngine(0, max_batch=b, max_kpts=1024, precision=prec, state_dict=sd)
        for w, v in knobs:
            eng.lib.gn_debug_set_variant(eng.ctx, w, v)
        for _ in range(3):
            eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        eng.set_kernel_timing(400)
        eng.match(inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
        torch.cuda.synchronize()
        tot = 0.0
        for r in sorted(eng.kernel_table(), key=lambda r: -r["ms"]):
            tot += r["ms"]
            print(f"   {r['name']:44s} x{int(r['launches']):3d}  {r['ms'] * 1e3 / r['launches']:8.2f} us each  {r['ms'] * 1e3:9.1f} us total")
        print(f"   recorded launches: {tot * 1e3:.1f} us")
        eng.set_kernel_timing(0)
        del eng
