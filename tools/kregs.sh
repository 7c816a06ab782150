#!/bin/bash
# Developer tool: register / LDS / scratch usage of the kernels in one object file whose mangled name contains a pattern.
#   bash tools/kregs.sh gisnav_amd/csrc/gn_superpoint.o k_sp_conv
D=$(mktemp -d); cp "$1" $D/x.o; (cd $D && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o >/dev/null 2>&1; /opt/rocm/lib/llvm/bin/llvm-readelf --notes x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 > notes.txt)
python3 - "$D/notes.txt" "$2" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", t, re.S):
    a, l, n, p, s, v = m.groups()
    if sys.argv[2] in n: print(n[:70], 'agpr', a, 'vgpr', v, 'sgpr', s, 'lds', l, 'scratch', p)
PY
echo $D
