#!/bin/bash
# Developer tool: the shader clock the chip sustains inside each big kernel of the bench step = GRBM_GUI_ACTIVE (cycles, per XCD) / the kernel's duration
# (MI355X_MICROARCH.md "DVFS give-back": effective clock = GRBM_GUI_ACTIVE / kernel wall time).  One rocprofv3 pass, --kernel-trace + --pmc only.
#   bash tools/kernel_clock.sh   -> gpurun_out/kernel_clock.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/kclk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/kclk -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --no-stream --no-rccl-check --substreams 1 > /dev/null 2>&1
python - <<PY | tee $O/kernel_clock.txt
import csv, glob, collections
cc = glob.glob("$O/kclk/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("$O/kclk/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
seen = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "gn::" not in r["Kernel_Name"]:
        continue
    seen[(r["Dispatch_Id"], r["Kernel_Name"])].append(float(r["Counter_Value"]))
for (d, k), vals in seen.items():
    if d not in dur: continue
    name = k.replace("void ", "").replace("gn::(anonymous namespace)::", "").split("(")[0]
    a = agg[name]; a[0] += sum(vals); a[1] += dur[d]; a[2] += 1
print("kernel                                   dispatches   avg us   GRBM_GUI_ACTIVE per dispatch   /8 XCDs -> GHz (if the counter sums the XCDs)   raw GHz")
for name, (c, t, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{name[:40]:40s} {n:6d} {t / n * 1e6:10.1f} {c / n:18.0f} {c / t / 8 / 1e9:12.3f} {c / t / 1e9:12.3f}")
PY
rm -rf $O/kclk
