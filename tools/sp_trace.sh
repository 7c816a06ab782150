#!/bin/bash
# Developer tool: per-dispatch durations of one SuperPoint call (4 x 1080p) from rocprofv3 --kernel-trace.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PREC=${1:-f16x2_bf16_attn}
cat > /tmp/sp_run1.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="$PREC", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0))
img = torch.from_numpy(np.random.default_rng(0).random((4, 1080, 1920), dtype=np.float32)).cuda()
for _ in range(2):
    sp.detect_and_describe_device(img)
torch.cuda.synchronize()
PY
rm -rf $O/sp_tr
rocprofv3 --kernel-trace --output-format csv -d $O/sp_tr -- python /tmp/sp_run1.py > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/sp_tr/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "k_sp_" in r["Kernel_Name"]]
half = rows[len(rows) // 2:]
for r in half:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 30: print(f"{r['Kernel_Name'][30:80]:52s} grid {r.get('Grid_Size_X','?'):>8s},{r.get('Grid_Size_Y','?'):>6s},{r.get('Grid_Size_Z','?'):>4s}  {d:9.1f} us")
PY
rm -rf $O/sp_tr
