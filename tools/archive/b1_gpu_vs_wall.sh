#!/bin/bash
# Developer: batch-1 wall time per gn_estimate call next to the sum of its kernels' durations (rocprofv3 kernel trace): how much of a call is
# the host enqueueing launches rather than the GPU executing them.  Run on the GPU box: bash tools/b1_gpu_vs_wall.sh  -> gpurun_out/b1_trace/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/b1_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b1_trace -o k -- python $R/tools/small_batch.py 1 "$@" > $R/gpurun_out/b1_trace.log 2>&1
python - <<PY
import csv, glob, re
f = glob.glob('$R/gpurun_out/b1_trace/**/k_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
log = open('$R/gpurun_out/b1_trace.log').read()
print(log.strip().split('\n')[-1])
calls = 330.0      # small_batch.py: 300 timed + 30 warm-up calls (engine creation and staging launch a few kernels more)
print('kernel time per call: %.1f us (sum over %d kernel names / %d calls)' % (tot / calls / 1e3, len(rows), calls))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print('  %-60s calls %6s  avg %8.2f us  total/call %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / calls / 1e3))
PY
