#!/bin/bash
# rocprofv3 kernel stats of the default bench (run on the GPU box):  bash tools/kstats.sh [bench args]  -> gpurun_out/kstats/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/kstats.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/kstats/k_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f}us pct={float(r['TotalDurationNs'])/tot*100:5.1f}")
PY
