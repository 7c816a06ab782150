#!/bin/bash
# Developer tool: matrix-pipe busy fraction and sustained clock of LoFTR's kernels (one forward without hipGraph; --kernel-trace + --pmc only).
#   bash tools/loftr_pmc.sh [exact_f32|split_fp16]   -> gpurun_out/loftr_pmc_<arith>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; A=${1:-exact_f32}; mkdir -p $O; rm -rf $O/lfpmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/lfpmc -- python $R/tools/loftr_profile.py $A 3 > /dev/null 2>&1
python - <<PY | tee $O/loftr_pmc_$A.txt
import csv, glob, collections
cc = glob.glob("$O/lfpmc/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("$O/lfpmc/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in csv.DictReader(open(kt))}
per = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
for r in csv.DictReader(open(cc)):
    if "gn::" not in r["Kernel_Name"]: continue
    per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = (r["Kernel_Name"].replace("void ", "").replace("gn::(anonymous namespace)::", "").split("(")[0], r.get("Grid_Size", ""))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for d, c in per.items():
    if d not in dur: continue
    a = agg[names[d]]; a[0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[1] += c.get("GRBM_GUI_ACTIVE", 0.0); a[2] += dur[d]; a[3] += 1
print("kernel (grid)                                                  launches   avg us   matrix pipe busy   clock GHz (GRBM_GUI_ACTIVE / 8 / duration)")
for (name, grid), (b, g, t, n) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:28]:
    print(f"{name[:44]:44s} {grid:>10s} {n:6d} {t / n * 1e6:9.1f} {b / (128.0 * g) if g else 0:12.3f} {g / 8 / t / 1e9 if t else 0:10.2f}")
PY
rm -rf $O/lfpmc
