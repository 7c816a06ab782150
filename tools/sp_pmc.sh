#!/bin/bash
# Developer tool: PMC counters of the SuperPoint kernels (own passes, --kernel-trace only).  bash tools/sp_pmc.sh [precision] [arithmetic]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PREC=${1:-f16x2_f16_attn}
ARITH=${2:-None}
cat > /tmp/sp_run2.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from gisnav_amd.engine import PoseEngine
from gisnav_amd.superpoint import SuperPoint
from oracle import superpoint as osp
eng = PoseEngine(0, max_batch=1, max_kpts=128, precision="$PREC", feature="superpoint")
sp = SuperPoint(engine=eng, max_keypoints=1024, state_dict=osp.synthetic_state_dict(0), arithmetic=$ARITH)
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((4, 1080, 1920), dtype=np.float32)).cuda()
for _ in range(2):
    sp.detect_and_describe_device(img)
torch.cuda.synchronize()
PY
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  rm -rf $O/sp_pmc
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/sp_pmc -- python /tmp/sp_run2.py > /dev/null 2>&1
  python $R/tools/pmc_summary.py $O/sp_pmc > /tmp/pmc.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("/tmp/pmc.json"))
for k, v in sorted(d.items()):
    if "k_sp_conv" in k:
        print(k[:28], {c: (round(x["sum"] / x["dispatches"]), x["dispatches"]) for c, x in v.items()})
PY
done
rm -rf $O/sp_pmc
