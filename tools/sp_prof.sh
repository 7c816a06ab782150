#!/bin/bash
# Developer tool: rocprofv3 kernel table of the SuperPoint extractor on 1920x1080 frames (4 per call) in a context of the given precision.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PREC=${1:-f16x2_bf16_attn}
ARITH=${2:-None}      # None (the context's default) | '"fp16"' | '"split_fp16"' | '"exact_f32"'
cat > /tmp/sp_run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="$PREC", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0), arithmetic=$ARITH)
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((4, 1080, 1920), dtype=np.float32)).cuda()
for _ in range(3):
    sp.detect_and_describe_device(img)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    sp.detect_and_describe_device(img)
torch.cuda.synchronize()
print("ms per image:", (time.perf_counter() - t0) / 20 * 1e3)
PY
python /tmp/sp_run.py
rm -rf $O/sp_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sp_prof -- python /tmp/sp_run.py > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/sp_prof/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
rm -rf $O/sp_prof
