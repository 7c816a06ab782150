"""Developer tool: is the FIRST forward of a fresh context bit-identical to the second one?  (tests/test_gpu_parity.py::
test_full_size_bitwise_repeatable_including_first_run, repeated, with the positions of any difference.)
usage: first_run_check.py <precision> <repeats> [sync]      sync = torch.cuda.synchronize() between context creation and the first call"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.synthetic import make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402

prec, reps = sys.argv[1], int(sys.argv[2])
sync = len(sys.argv) > 3
sd = synthetic_state_dict(0)
pairs = [make_pair(i) for i in range(32)]
T = 32 * 2 * 1024
bad = 0
for r in range(reps):
    eng = PoseEngine(0, max_batch=32, max_kpts=1024, precision=prec, state_dict=sd)
    inp = eng.stage_inputs(pairs)
    if sync:
        torch.cuda.synchronize()
    a = (inp["desc_q"], inp["kpt_q"], inp["n_q"], inp["desc_r"], inp["kpt_r"], inp["n_r"])
    eng.match(*a); torch.cuda.synchronize()
    x1 = eng.debug_read("x", T * 256).view(np.uint32).reshape(T, 256).copy()
    eng.match(*a); torch.cuda.synchronize()
    x2 = eng.debug_read("x", T * 256).view(np.uint32).reshape(T, 256).copy()
    eng.match(*a); torch.cuda.synchronize()
    x3 = eng.debug_read("x", T * 256).view(np.uint32).reshape(T, 256).copy()
    d12, d23 = (x1 != x2), (x2 != x3)
    if d12.any() or d23.any():
        bad += 1
        rows = np.nonzero(d12.any(axis=1))[0]
        print(f"rep {r}: first vs second differ in {int(d12.sum())} words, {len(rows)} token rows (first rows {rows[:8]}, slots {sorted(set(rows // 1024))[:10]}); second vs third: {int(d23.sum())} words")
    del eng
print(f"{prec}: {bad} of {reps} fresh contexts had a first run that differed" + (" (with sync)" if sync else ""))
