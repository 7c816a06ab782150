#!/bin/bash
# Developer tool: step time and k_ffn_fused average duration for the two workgroup shapes (knob 14) at batch 32, 4 and 1.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for b in ${BATCHES:-32 4 1}; do for v in 64 32; do
  rm -rf $O/ffn_shape_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ffn_shape_$v -- python $R/bench.py --batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --debug-variant 14:$v > $O/shape_${b}_$v.log 2>&1
  echo "batch $b shape $v: $(grep k_ffn_fused $(ls $O/ffn_shape_$v/*/*kernel_stats.csv | head -1) | awk -F, "{print \$(NF-5), \$(NF-4), \$(NF-3)}")  ms_per_step=$(grep -o '"ms_per_step": [0-9.]*' $O/shape_${b}_$v.log)"
  rm -rf $O/ffn_shape_$v
done; done
