#!/bin/bash
# developer tool: PMC passes over the attention kernels (run on the GPU box through gpurun): attn_pmc.sh <variant> [...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for var in "$@"; do
  O=$R/gpurun_out/attn_pmc_$var
  mkdir -p $O
  for grp in "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_BUSY_sum TD_BUSY_sum" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_IFETCH SQ_INSTS_BRANCH"; do
    tag=$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$tag -- python $R/tools/attn_ablate.py $var > $O/$tag.log 2>&1
  done
  python $R/tools/pmc_summary.py $O > $O/summary.json
  python - <<PY
import json, glob, csv
d=json.load(open("$O/summary.json"))
for k,v in d.items():
    if "attn" in k:
        print("variant $var", k, {c: round(x["avg"],1) for c,x in sorted(v.items())})
f=glob.glob("$O/GRBM_GUI_ACTIVE_SQ_WAVES/**/*kernel_trace.csv",recursive=True)[0]
dur=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "attn" in r["Kernel_Name"]]
print("variant $var avg duration ns", sum(dur)/len(dur))
PY
done
