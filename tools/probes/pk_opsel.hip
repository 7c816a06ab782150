// Developer probe (not part of the library): does the IGNORED half of a v_pk_*_f32 operand pair influence the result on gfx950?
// With the SLP vectoriser on, hipcc turns the rotary epilogue of k_qkv into v_pk_mul_f32 / v_pk_fma_f32 whose broadcast operand is a
// register pair with only the low register written (op_sel_hi:0 -> both lanes read the low register); the high register holds whatever
// the previous owner of the VGPR left there.  k_qkv built that way is not bitwise repeatable run to run (gisnav_amd/build.py).
//   part 1: one wave, the ignored register set to special bit patterns, result compared with the scalar product
//   part 2: 8 waves per block, 4096 blocks, the ignored register NEVER written in this kernel (leftovers of other waves), repeated
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_opsel.hip -o tools/probes/pk_opsel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

__global__ void k_set(const float* in, unsigned garbage, float* out) {
  const int t = threadIdx.x;
  const float a = in[t], b0 = in[64 + t], b1 = in[128 + t], c0 = in[192 + t], c1 = in[256 + t];
  float r0, r1, f0, f1;
  asm volatile(
      "v_mov_b32 v10, %4\n\t"
      "v_mov_b32 v11, %9\n\t"
      "v_mov_b32 v12, %5\n\tv_mov_b32 v13, %6\n\t"
      "v_mov_b32 v16, %7\n\tv_mov_b32 v17, %8\n\t"
      "s_nop 4\n\t"
      "v_pk_mul_f32 v[14:15], v[10:11], v[12:13] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 v[18:19], v[12:13], v[10:11], v[16:17] op_sel_hi:[1,0,1]\n\t"
      "s_nop 4\n\t"
      "v_mov_b32 %0, v14\n\tv_mov_b32 %1, v15\n\tv_mov_b32 %2, v18\n\tv_mov_b32 %3, v19"
      : "=v"(r0), "=v"(r1), "=v"(f0), "=v"(f1)
      : "v"(a), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(garbage)
      : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
  out[t] = r0; out[64 + t] = r1; out[128 + t] = f0; out[192 + t] = f1;
}

// leaves distinctive bit patterns in as many VGPRs as the compiler will give it
__global__ __launch_bounds__(256) void k_litter(unsigned* sink, unsigned pattern) {
  unsigned v[96];
#pragma unroll
  for (int i = 0; i < 96; ++i) v[i] = pattern ^ (unsigned)(i * 0x01000193u) ^ threadIdx.x;
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 96; ++i) asm volatile("v_mov_b32 %0, %0" : "+v"(v[i]));
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 96; ++i) s += v[i];
  if (s == 0x12345) sink[0] = s;
}

__global__ __launch_bounds__(512) void k_unwritten(const float* in, float* out) {
  const int t = threadIdx.x & 63, g = blockIdx.x * 512 + threadIdx.x;
  const float a = in[t], b0 = in[64 + t], b1 = in[128 + t];
  float r0, r1;
  asm volatile(
      "v_mov_b32 v200, %2\n\t"             // v201: never written here
      "v_mov_b32 v202, %3\n\tv_mov_b32 v203, %4\n\t"
      "s_nop 4\n\t"
      "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel_hi:[0,1]\n\t"
      "s_nop 4\n\t"
      "v_mov_b32 %0, v204\n\tv_mov_b32 %1, v205"
      : "=v"(r0), "=v"(r1)
      : "v"(a), "v"(b0), "v"(b1)
      : "v200", "v201", "v202", "v203", "v204", "v205");
  out[2 * (size_t)g] = r0; out[2 * (size_t)g + 1] = r1;
}

int main() {
  std::vector<float> h(320);
  for (int i = 0; i < 320; ++i) h[i] = 0.37f + 0.0131f * i * ((i & 1) ? -1.f : 1.f);
  float *in, *out; unsigned* sink;
  hipMalloc(&in, 320 * 4); hipMalloc(&out, (size_t)4096 * 512 * 2 * 4); hipMalloc(&sink, 4);
  hipMemcpy(in, h.data(), 320 * 4, hipMemcpyHostToDevice);
  const unsigned pats[] = {0u, 0x7fc00000u, 0x7f800001u, 0xffc00001u, 0x7f800000u, 0xff800000u, 0x00000001u, 0x807fffffu, 0xffffffffu, 0x3f800000u, 0x7f7fffffu};
  int bad = 0;
  for (unsigned p : pats) {
    k_set<<<1, 64>>>(in, p, out);
    float r[256];
    hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost);
    int nb = 0;
    for (int t = 0; t < 64; ++t) {
      const float a = h[t], b0 = h[64 + t], b1 = h[128 + t], c0 = h[192 + t], c1 = h[256 + t];
      const float e[4] = {a * b0, a * b1, __builtin_fmaf(b0, a, c0), __builtin_fmaf(b1, a, c1)};
      for (int k = 0; k < 4; ++k) nb += memcmp(&e[k], &r[64 * k + t], 4) != 0;
    }
    printf("ignored half = %08x: %d of 256 results differ from the scalar value\n", p, nb);
    bad += nb;
  }
  std::vector<float> big((size_t)4096 * 512 * 2);
  long total = 0;
  for (int rep = 0; rep < 20; ++rep) {
    k_litter<<<2048, 256>>>(sink, 0x7f800001u + rep * 0x00400000u);
    k_unwritten<<<4096, 512>>>(in, out);
    hipMemcpy(big.data(), out, big.size() * 4, hipMemcpyDeviceToHost);
    long nb = 0;
    for (size_t g = 0; g < (size_t)4096 * 512; ++g) {
      const int t = g & 63;
      const float e0 = h[t] * h[64 + t], e1 = h[t] * h[128 + t];
      nb += memcmp(&e0, &big[2 * g], 4) != 0; nb += memcmp(&e1, &big[2 * g + 1], 4) != 0;
    }
    total += nb;
  }
  printf("unwritten ignored half, 20 x 2 M waves-lanes after a littering kernel: %ld results differ\n", total);
  printf(bad + total ? "RESULT: the ignored half matters\n" : "RESULT: the ignored half does not matter\n");
  return 0;
}
