#!/bin/bash
# developer tool: PMC passes over the attention kernels (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/attn_pmc
for grp in "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/attn_pmc/$tag -- python $R/tools/attn_ablate.py $1 > $R/gpurun_out/attn_pmc/$tag.log 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/attn_pmc > $R/gpurun_out/attn_pmc/summary.json
python - <<PY
import json
d=json.load(open("$R/gpurun_out/attn_pmc/summary.json"))
for k,v in d.items():
    if "attn" in k:
        print(k, {c: round(x["avg"],1) for c,x in v.items()})
PY
