#!/bin/bash
# on the GPU box: bash tools/slp_ab.sh <runs> tag...   -> gpurun_out/slp/ab_<tag>.log (a reference from the no-SLP build, then tools/slp_diag.py diff for each tag)
mkdir -p gpurun_out/slp; R=$1; shift
GISNAV_AMD_LIB=tools/probes/variants/lib_noslp.so timeout 200 python tools/slp_diag.py save /tmp/slp_ref.npy 2>&1 | grep -v amdgpu.ids
for t in "$@"; do
  GISNAV_AMD_LIB=tools/probes/variants/lib_$t.so timeout 300 python tools/slp_diag.py diff /tmp/slp_ref.npy $R 2>&1 | grep -v amdgpu.ids > gpurun_out/slp/ab_$t.log
  echo "== $t: $(grep 'runs differ' gpurun_out/slp/ab_$t.log)"
done
