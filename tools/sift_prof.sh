#!/bin/bash
# rocprofv3 kernel stats of the SIFT extractor (run on the GPU box):  bash tools/sift_prof.sh  -> gpurun_out/sift_prof/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/sift_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sift_prof -o sift -- python $R/tools/bench_sift.py --steps 20 $SIFT_BENCH_ARGS > $R/gpurun_out/sift_prof.log 2>&1
find $R/gpurun_out/sift_prof -name "*kernel_stats.csv" -exec head -14 {} \; | cut -d, -f1-6 | cut -c1-150
