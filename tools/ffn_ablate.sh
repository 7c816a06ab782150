#!/bin/bash
# Developer tool: rocprofv3 average duration of k_ffn_fused under its timing-only ablations (knob 12), through bench.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in ${@:-0 1 2 3 4 7}; do
  rm -rf $O/ffn_abl_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ffn_abl_$v -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --debug-variant 12:$v > /dev/null 2>&1
  echo "ablation $v: $(grep k_ffn_fused $(ls $O/ffn_abl_$v/*/*kernel_stats.csv | head -1) | awk -F, "{print \$(NF-5), \$(NF-4)}")"
  rm -rf $O/ffn_abl_$v
done
