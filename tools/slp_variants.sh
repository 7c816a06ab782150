#!/bin/bash
# Developer tool: libraries that differ ONLY in how gn_qkv.hip is compiled, for tools/flake_layers.py A/B runs on the GPU box:
#   bash tools/slp_variants.sh          (here, no GPU needed)  ->  tools/probes/variants/lib_<tag>.so
#   GISNAV_AMD_LIB=tools/probes/variants/lib_<tag>.so python tools/flake_layers.py f16x2_bf16_attn 40     (on the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/gisnav_amd/csrc; V=$R/tools/probes/variants
python -m gisnav_amd.build > /dev/null
OBJS=$(ls $C/*.o | grep -v gn_qkv.o)
build() {  # tag, extra flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-value $2 -c $C/gn_qkv.hip -o $V/gn_qkv_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib_$1.so $OBJS $V/gn_qkv_$1.o
  echo built $V/lib_$1.so
}
build noslp "-fno-slp-vectorize"
build slp ""
build slp_wait0 "-mllvm -amdgpu-waitcnt-forcezero"
build noslp_wait0 "-fno-slp-vectorize -mllvm -amdgpu-waitcnt-forcezero"

# ---- variants made by EDITING the SLP build's device assembly (tools/slp_asm_edit.py) and re-assembling it
L=/opt/rocm/lib/llvm/bin; W=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-value --cuda-device-only -S $C/gn_qkv.hip -o $W/slp.s
asm_variant() {  # name of an edit in tools/slp_asm_edit.py
  python $R/tools/slp_asm_edit.py $W/slp.s $W/$1.s $1
  $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/$1.s -o $W/$1.o
  $L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/$1.out $W/$1.o
  $L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/$1.out -output=$W/$1.hipfb
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-value -Wno-unused-command-line-argument --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/$1.hipfb -c $C/gn_qkv.hip -o $V/gn_qkv_asm_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib_asm_$1.so $OBJS $V/gn_qkv_asm_$1.o
  echo built $V/lib_asm_$1.so
}
for e in base scalar_fma scalar_mul nop_before_pk nop_around_suspect opsel_copy not_in_place; do asm_variant $e; done
rm -rf $W
