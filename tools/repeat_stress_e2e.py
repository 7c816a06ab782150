"""Developer tool: bitwise repeatability beyond the matcher -- poses of gn_estimate (gather + PnP) at batch 32 with and without sub-batch streams,
and the SuperPoint extractor in its three arithmetics on 1080p frames: every output identical over many runs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine  # noqa: E402
from gisnav_amd.superpoint import SuperPoint  # noqa: E402
from gisnav_amd.synthetic import K_MATRIX, make_pair  # noqa: E402
from gisnav_amd.weights import synthetic_state_dict  # noqa: E402
from oracle import superpoint as osp  # noqa: E402  (weights only)

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad_total = 0
eng = PoseEngine(0, max_batch=32, max_kpts=1024, precision="f16x2_bf16_attn", state_dict=synthetic_state_dict(0))
inp = eng.stage_inputs([make_pair(i, n_q=1024 - (i % 5) * 17, n_r=1024 - (i % 3) * 29) for i in range(32)])
for nsub in (1, 2):
    eng.set_substreams(nsub)
    ref, bad = None, 0
    for r in range(runs):
        out = eng.estimate(inp, K_MATRIX)
        eng.flush(); torch.cuda.synchronize()
        cur = {k: v.cpu().numpy().copy() for k, v in out.items()}
        if ref is None:
            ref = cur
        elif not all(np.array_equal(ref[k].view(np.uint8), cur[k].view(np.uint8)) for k in ref):
            bad += 1
    print(f"gn_estimate, {nsub} sub-batch stream(s): {bad} of {runs} runs differ from the first")
    bad_total += bad
del eng
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="f16x2_bf16_attn", feature="superpoint")
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((4, 1080, 1920), dtype=np.float32)).cuda()
for arith in ("split_fp16", "fp16", "exact_f32"):
    sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0), arithmetic=arith)
    ref, bad = None, 0
    for r in range(max(8, runs // 4)):
        kpt, score, desc, n = sp.detect_and_describe_device(img)
        torch.cuda.synchronize()
        cur = (kpt.cpu().numpy().copy(), score.cpu().numpy().copy(), desc.cpu().numpy().copy(), n.copy())
        if ref is None:
            ref = cur
        elif not all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(ref, cur)):
            bad += 1
    print(f"SuperPoint {arith}: {bad} of {max(8, runs // 4)} runs differ from the first")
    bad_total += bad
sys.exit(1 if bad_total else 0)
