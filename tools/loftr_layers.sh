#!/bin/bash
# Developer tool: per-LAUNCH durations of one LoFTR forward (640x480 pair, rocprofv3 kernel trace, no hipGraph).   bash tools/loftr_layers.sh [exact_f32|split_fp16]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARITH=${1:-exact_f32}
rm -rf $O/lf_lay
rocprofv3 --kernel-trace --output-format csv -d $O/lf_lay -- python $R/tools/loftr_profile.py $ARITH 4 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$O/lf_lay/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "gn::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_lf_conv1" in r["Kernel_Name"] and "k_lf_conv1" not in rows[i - 1]["Kernel_Name"]) if any("k_lf_conv1" in r["Kernel_Name"] for r in rows) else 0
# a forward starts with the stem convolution (two images: may be one launch or two); take the last forward
starts = [i for i, r in enumerate(rows) if "k_lf_conv1" in r["Kernel_Name"]]
first = starts[-1] if len(starts) == 4 else starts[-2] if len(starts) == 8 else starts[-1]
tot = 0; agg = collections.OrderedDict()
for r in rows[first:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = r["Kernel_Name"].replace("gn::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    key = (name, r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
for (name, gx, gy, gz), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{name[:44]:44s} grid {gx:>8s} x{gy:>4s} x{gz:>3s}  launches {n:3d}  avg {t / n:8.1f} us  total {t:8.1f} us")
print("kernel time of one forward: %.1f us over %d launches" % (tot, len(rows) - first))
PY
rm -rf $O/lf_lay
