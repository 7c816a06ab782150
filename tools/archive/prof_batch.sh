#!/bin/bash
# Developer tool: rocprofv3 kernel table of bench.py at one batch size (default 1), top rows.  usage: prof_batch.sh [batch] [extra bench args]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
b=${1:-1}; shift
python $R/bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-traffic --no-extras "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
rm -rf $O/prof_b$b
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$b -- python $R/bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-traffic --no-extras "$@" > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/prof_b$b/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f"{r['Name'][:64]:64s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
cp $(ls $O/prof_b$b/*/*kernel_stats.csv | head -1) $O/kernel_stats_b$b.csv; rm -rf $O/prof_b$b
