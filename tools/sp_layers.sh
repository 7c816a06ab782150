#!/bin/bash
# Developer tool: per-LAUNCH durations of the SuperPoint extractor's kernels (rocprofv3 kernel trace, last call of the run), 4 x 1080p per call.
#   bash tools/sp_layers.sh [precision] [arithmetic]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PREC=${1:-f16x2_f16_attn}
ARITH=${2:-None}
cat > /tmp/sp_run1.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="$PREC", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0), arithmetic=$ARITH)
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((4, 1080, 1920), dtype=np.float32)).cuda()
for _ in range(4):
    sp.detect_and_describe_device(img)
torch.cuda.synchronize()
PY
rm -rf $O/sp_lay
rocprofv3 --kernel-trace --output-format csv -d $O/sp_lay -- python /tmp/sp_run1.py > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/sp_lay/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_sp_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = everything behind the second-to-last k_sp_describe (the first convolution is fused into the second one's kernel since round 6)
ends = [i for i, r in enumerate(rows) if "k_sp_describe" in r["Kernel_Name"]]
last = ends[-2] + 1
tot = 0
for r in rows[last:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = r["Kernel_Name"].replace("gn::(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:40]:40s} grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8s} x{r.get('Grid_Size_Y','')} x{r.get('Grid_Size_Z','')} {d:9.1f} us")
print("sum of kernel time of one call: %.1f us (4 images)" % tot)
PY
rm -rf $O/sp_lay
