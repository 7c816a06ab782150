// Developer probe (not part of the library): do VALU and MFMA instructions overlap on one gfx950 SIMD?
//   mode 0: MFMA only          mode 1: VALU only (v_fma_f32)      mode 2: same wave, MFMA + 7 independent VALU per MFMA
//   mode 3: 512-thread blocks, waves 0-3 MFMA only, waves 4-7 VALU only (one of each per SIMD)
//   mode 4: VALU only with v_exp_f32 (transcendental rate)       mode 5: as 3 but VALU waves run v_exp_f32
//   mode 6: same wave, MFMA + 2 independent v_exp_f32 per MFMA   mode 7: VALU only, 2 v_exp_f32 per slot (the VALU half of mode 6)
//   mode 8: same wave, MFMA + 2 v_exp_f32 + 3 v_fma_f32 per MFMA
//   mode 9 / 10: MFMA only, the 4 MFMAs of a body on 2 / 1 accumulators (dependent distance 2 / 1 instead of 4)
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/overlap.hip -o tools/probes/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + lane + e); b[e] = (short)(0x3f00 + 2 * lane + e); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 0.001f + i;
  const float c1 = 0.999f, c2 = 0.001f;
  const bool do_mfma = MODE == 0 || MODE == 2 || MODE == 6 || MODE == 8 || MODE == 9 || MODE == 10 || ((MODE == 3 || MODE == 5) && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 2 || MODE == 4 || (MODE >= 6 && MODE <= 8) || ((MODE == 3 || MODE == 5) && wave >= 4);
  const bool use_exp = MODE == 4 || MODE == 5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (do_mfma) { constexpr int NA = MODE == 9 ? 2 : (MODE == 10 ? 1 : 4); acc[j % NA] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j % NA], 0, 0, 0); }
      if (do_valu) {
        if (MODE >= 6) {
#pragma unroll
          for (int i = 0; i < 2; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
          if (MODE == 8) {
#pragma unroll
            for (int i = 2; i < 5; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);
          }
        } else if (use_exp) {
#pragma unroll
          for (int i = 0; i < 7; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 7; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);
        }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float* d, int blocks, int threads, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float* d; hipMalloc(&d, 4096 * 512 * 4);
  const int iters = 20000;
  // one block per CU: 256 threads = 1 wave / SIMD; 512 threads = 2 waves / SIMD
  printf("cycles below are per (4 MFMA [+ 28 VALU]) loop body, assuming 2.4 GHz\n");
  struct { const char* name; float ms; } r[] = {
    {"mode0 MFMA only, 1 wave/SIMD", run<0>(d, 256, 256, iters)},
    {"mode1 VALU fma only, 1 wave/SIMD", run<1>(d, 256, 256, iters)},
    {"mode2 same wave MFMA + 7 VALU each", run<2>(d, 256, 256, iters)},
    {"mode0 MFMA only, 2 waves/SIMD", run<0>(d, 256, 512, iters)},
    {"mode1 VALU only, 2 waves/SIMD", run<1>(d, 256, 512, iters)},
    {"mode3 MFMA waves + VALU waves (1+1 per SIMD)", run<3>(d, 256, 512, iters)},
    {"mode4 VALU exp only, 1 wave/SIMD", run<4>(d, 256, 256, iters)},
    {"mode5 MFMA waves + exp waves (1+1 per SIMD)", run<5>(d, 256, 512, iters)},
    {"mode2 same wave MFMA+VALU, 2 waves/SIMD", run<2>(d, 256, 512, iters)},
    {"mode7 2 exp per slot only, 1 wave/SIMD", run<7>(d, 256, 256, iters)},
    {"mode6 same wave MFMA + 2 exp each, 1 wave/SIMD", run<6>(d, 256, 256, iters)},
    {"mode8 same wave MFMA + 2 exp + 3 fma, 1 wave/SIMD", run<8>(d, 256, 256, iters)},
    {"mode6 same wave MFMA + 2 exp each, 2 waves/SIMD", run<6>(d, 256, 512, iters)},
    {"mode9 MFMA only, 2 accumulators, 1 wave/SIMD", run<9>(d, 256, 256, iters)},
    {"mode10 MFMA only, 1 accumulator, 1 wave/SIMD", run<10>(d, 256, 256, iters)},
    {"mode9 MFMA only, 2 accumulators, 2 waves/SIMD", run<9>(d, 256, 512, iters)},
    {"mode10 MFMA only, 1 accumulator, 2 waves/SIMD", run<10>(d, 256, 512, iters)},
  };
  for (auto& x : r) printf("%-48s %8.3f ms  %7.1f cycles/body\n", x.name, x.ms, x.ms * 1e-3 * 2.4e9 / iters);
  return 0;
}
