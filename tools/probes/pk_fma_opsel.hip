// Developer probe (not part of the library): v_pk_fma_f32 with op_sel:[0,1,0] -- the LOW lane multiplies by the HIGH register of source 1.
// hipcc's SLP vectoriser emits exactly this for the second rotary pair of k_qkv (o.z = v.z * cos' - v.w * sin'), and in that kernel the low lane
// intermittently returns source 2 alone (the product is dropped) in lanes 48..63 (docs/DESIGN_HISTORY.md 12.5; tools/slp_variants.sh: replacing only this
// instruction by two v_fma_f32 makes the kernel bitwise repeatable).  This probe looks for the same thing outside k_qkv: 8 waves per CU, each alternating
// an MFMA phase (LDS-fed, like the projection's k-loop) with an epilogue of packed rotations on freshly loaded table entries, compared lane by lane
// with scalar v_fma_f32 results.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_fma_opsel.hip -o tools/probes/pk_fma_opsel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: op_sel:[0,1,0] (the suspect);  1: op_sel_hi:[1,0,1] (the first rotary pair's form: never seen wrong);  2: suspect form, s_nop 4 around it
__global__ __launch_bounds__(512) void k_probe(const f32x4* __restrict__ table, int entries, int rounds, int mfmas, unsigned* __restrict__ bad_lane, unsigned* __restrict__ bad_kind) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 1024];   // one workgroup per CU, as k_qkv
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 128 * 1024 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + i, 0x38003800u, 0x3a003a00u, 0x3c003c00u);
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.001f * (float)(lane + r + 16 * j);
  unsigned nbad = 0, kinds = 0;
  for (int round = 0; round < rounds; ++round) {
    // ---- MFMA phase (waves drift apart: wave w runs mfmas + 4 w steps)
    for (int ks = 0; ks < mfmas + 4 * wave; ++ks) {
      const f16x8 x = *reinterpret_cast<const f16x8*>(smem + ((ks * 1024 + lane * 16 + wave * 8192) & (128 * 1024 - 16)));
      const f16x8 w = *reinterpret_cast<const f16x8*>(smem + ((ks * 2048 + lane * 16 + 64 * 1024) & (128 * 1024 - 16)));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = acc[j][r] * 1e-3f + 0.25f;   // keep the values tame
    // ---- epilogue: 16 table entries requested back to back, then 16 rotations
    f32x4 rot[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) rot[e] = table[(size_t)((blockIdx.x * 977 + round * 131 + e * 4099) % (entries / 64)) * 64 + lane];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = e >> 2, g = e & 3;
      typedef unsigned long long u64;
      auto pair = [](float lo, float hi) { return (u64)__builtin_bit_cast(unsigned, lo) | ((u64)__builtin_bit_cast(unsigned, hi) << 32); };
      const float az = acc[j][4 * g + 2], aw = acc[j][4 * g + 3];          // (v.z, v.w)
      const u64 a = pair(az, aw), t = pair(-aw, az);
      const u64 rw = pair(rot[e].w, __builtin_bit_cast(float, 0x7fc00001u ^ (unsigned)tid));   // (sin', don't care)
      const u64 rxy = pair(rot[e].x, rot[e].y);
      u64 c, d;
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(c) : "v"(rw), "v"(t));
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(rxy), "v"(c));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(rxy), "v"(c));
      if (MODE == 2) asm volatile("s_nop 4\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]\n\ts_nop 4" : "=v"(d) : "v"(a), "v"(rxy), "v"(c));
      const float m = MODE == 1 ? rot[e].x : rot[e].y;
      const unsigned cl = (unsigned)c, ch = (unsigned)(c >> 32), dl = (unsigned)d, dh = (unsigned)(d >> 32);
      float wl, wh;
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(wl) : "v"(az), "v"(m), "v"(cl));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(wh) : "v"(aw), "v"(m), "v"(ch));
      const bool bl = dl != __builtin_bit_cast(unsigned, wl), bh = dh != __builtin_bit_cast(unsigned, wh);
      if (bl || bh) {
        ++nbad;
        kinds |= (bl ? 1u : 0u) | (bh ? 2u : 0u) | (bl && dl == cl ? 4u : 0u);
      }
      acc[j][4 * g + 2] = __builtin_bit_cast(float, dl); acc[j][4 * g + 3] = __builtin_bit_cast(float, dh);
    }
  }
  if (nbad) { atomicAdd(&bad_lane[lane], nbad); atomicOr(bad_kind, kinds); }
  if (acc[0][0] == 12345.678f) bad_kind[1] = 1;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200, mfmas = argc > 2 ? atoi(argv[2]) : 16, reps = argc > 3 ? atoi(argv[3]) : 20;
  const int entries = 1 << 22;    // 64 MB table: the loads really go to memory
  std::vector<float> h((size_t)entries * 4);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 8388608.f) - 1.f; }
  f32x4* table; unsigned *bad, *kind;
  hipMalloc(&table, h.size() * 4); hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&bad, 256); hipMalloc(&kind, 8);
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(bad, 0, 256); hipMemset(kind, 0, 8);
    for (int r = 0; r < reps; ++r) {
      if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(512), dim3(512), 0, 0, table, entries, rounds, mfmas, bad, kind);
      if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(512), dim3(512), 0, 0, table, entries, rounds, mfmas, bad, kind);
      if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(512), dim3(512), 0, 0, table, entries, rounds, mfmas, bad, kind);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    unsigned hb[64], hk[2];
    hipMemcpy(hb, bad, 256, hipMemcpyDeviceToHost); hipMemcpy(hk, kind, 8, hipMemcpyDeviceToHost);
    unsigned long long tot = 0; unsigned q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) { tot += hb[l]; q[l >> 4] += hb[l]; }
    printf("mode %d (%s): %llu wrong of %.3g packed results;  by lane quarter %u %u %u %u;  kinds: low lane %d, high lane %d, low lane == source 2 %d\n", mode,
           mode == 0 ? "op_sel:[0,1,0]" : mode == 1 ? "op_sel_hi:[1,0,1]" : "op_sel:[0,1,0] between s_nop 4", tot, (double)reps * 512 * 512 * rounds * 16, q[0], q[1], q[2], q[3],
           (int)(hk[0] & 1), (int)((hk[0] >> 1) & 1), (int)((hk[0] >> 2) & 1));
  }
  return 0;
}
