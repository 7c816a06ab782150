// Developer probe (not part of the library): what ONE wave per SIMD can issue on gfx950 (256-thread workgroups, one per CU).
// Cycles (s_memtime) per MFMA / per VALU instruction for the instruction mixes k_ffn128 (gn_ffn128.hip) is built from:
//   0  48 MFMAs (32x32x16 f16) per iteration on 16 accumulators, operands in registers
//   1  + 8 ds_read_b128 per iteration (next iteration's B fragments)
//   2  + 8 buffer_load_dwordx4 per iteration (L2-resident 1 MB array, 3 iterations in flight)
//   3  1 + 2 together
//   4  VALU only: 8 independent v_fma_f32 chains (register constants)        5  the same with v_pk_fma_f32 (16 values)
//   6  VALU only: v_fmaak_f32 (32-bit literal, 8-byte encoding)               7  v_exp_f32 only
//   10 + k: MFMA + k independent v_fma_f32 per MFMA (k = 1..8), order pinned
//   20 + k: MFMA + k v_pk_fma_f32 per MFMA
//   8   the k-step of k_ffn128's GEMM 1: 3 x 2 fragment buffers read in pairs one token tile ahead, swizzled k-tile layout, 4-slot weight ring
//   30: mode 0 fully unrolled, executed ONCE per pass over 8 KB of code x 8 (cold instruction cache)
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/stream1w.hip -o tools/probes/stream1w
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define PIN() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned char* w, float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(w), 0, 1 << 20, 0x00020000);
  f16x8 a[4], b[2][4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (_Float16)(0.01f * (lane + e + i)); b[0][i][e] = b[1][i][e] = (_Float16)(0.02f * (lane - e + i)); }
  f32x16 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + lane * 0.001f + i;
  float c1 = 0.999f + 1e-6f * iters, c2 = 0.001f;
  u32x4 ring[3][8];
  for (int q = 0; q < 3; ++q) for (int i = 0; i < 8; ++i) ring[q][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (q * 8 + i) * 1024, 0);
  constexpr bool MF = (MODE <= 3 || MODE >= 10) && MODE != 8 && (MODE < 40 || MODE > 43), DS = MODE == 1 || MODE == 3, VM = MODE == 2 || MODE == 3;
  constexpr int KV = (MODE >= 10 && MODE < 20) ? MODE - 10 : 0, KP = (MODE >= 20 && MODE < 30) ? MODE - 20 : 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 8 || MODE == 9 || (MODE >= 40 && MODE <= 43)) {
  } else if (MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 6; ++rep) {
        if (MODE == 4) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);
        } else if (MODE == 5) {
#pragma unroll
          for (int i = 0; i < 16; i += 2) { f32x2 p = {v[i], v[i + 1]}; p = p * (f32x2){c1, c1} + (f32x2){c2, c2}; v[i] = p[0]; v[i + 1] = p[1]; }
        } else if (MODE == 6) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], v[i + 8], 0.0015096671413630247f);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
        }
        PIN();
      }
    }
  } else {
    for (int it0 = 0; it0 < (MODE == 30 ? 1 : iters); it0 += (MODE == 30 ? 1 : 6)) {
#pragma unroll
      for (int rep = 0; rep < (MODE == 30 ? 64 : 6); ++rep) {      // six iterations per trip: the buffer / ring indices stay compile-time constants
        const int it = rep, cb = rep & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int m = 12 * j + 4 * p + i;
              if (MF) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(VM ? __builtin_bit_cast(f16x8, ring[it % 3][(2 * i + (p == 0)) & 7]) : a[i], b[DS ? cb : 0][j], acc[i][j], 0, 0, 0);
              if (DS && m % 6 == 1) b[cb ^ 1][(m / 6) & 3] = *reinterpret_cast<const f16x8*>(lds + (((m / 6) * 1024 + lane * 16 + (it0 + it) * 64) & 65520));
              if (VM && m % 6 == 4) ring[(it + 2) % 3][m / 6] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (((it0 + it + 3) * 8 + m / 6) * 1024) & ((1 << 20) - 1), 0);
#pragma unroll
              for (int k = 0; k < KV; ++k) v[k] = __builtin_fmaf(v[k], c1, c2);
#pragma unroll
              for (int k = 0; k < KP; ++k) { f32x2 pp = {v[2 * k], v[2 * k + 1]}; pp = pp * (f32x2){c1, c1} + (f32x2){c2, c2}; v[2 * k] = pp[0]; v[2 * k + 1] = pp[1]; }
              PIN();
            }
      }
    }
  }
  if (MODE == 8 || MODE == 9 || (MODE >= 40 && MODE <= 43)) {
    // 42: as 40, every workgroup starts at a different k-step of the (cyclic) weight stream; 43: as 40 with two waves per slice (half the traffic)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // (provably uniform: the offsets below stay in SGPRs)
    const int wofs = (MODE == 43 ? (wv >> 1) * 262144 : (MODE == 40 || MODE == 41 || MODE == 42) ? wv * 262144 : 0) + (MODE == 42 ? (blockIdx.x * 37 % 128) * 8192 : 0);      // 40: every wave streams its own 256 KB slice (4 x the L2 -> CU traffic)
    constexpr int RD = MODE == 41 ? 7 : 4;      // 41: six k-steps (48 KB per wave) in flight instead of three
    f16x8 bq[3][2], fa[RD][4][2];
    const int ql = lane & 31, hh = lane >> 5;
    unsigned bo[2][4];
    for (int jp = 0; jp < 2; ++jp) for (int c = 0; c < 4; ++c) { const int r = 32 * jp + ql; bo[jp][c] = r * 128 + (((2 * c + hh) ^ ((r ^ (r >> 3)) & 7)) * 16); }
    for (int q = 0; q < RD; ++q) for (int i = 0; i < 4; ++i) for (int pl = 0; pl < 2; ++pl) fa[q][i][pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, wofs + ((q * 8 + 2 * i + pl) * 1024), 0));
    for (int pl = 0; pl < 2; ++pl) bq[0][pl] = *reinterpret_cast<const f16x8*>(lds + bo[0][pl]);
    constexpr int TRIP = MODE == 41 ? 84 : 12;
    for (int it0 = 0; it0 < (MODE == 9 ? 1 : iters); it0 += TRIP) {
#pragma unroll
      for (int n2 = 0; n2 < (MODE == 9 ? 60 : TRIP); ++n2) {      // 9: 60 k-steps of straight-line code, executed once
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int g = 4 * n2 + j, g1 = g + 1, j1 = g1 & 3, ks1 = (g1 >> 2) & 1, t1 = ((g1 >> 3) & 3) * 16384;
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) bq[g1 % 3][pl] = *reinterpret_cast<const f16x8*>(lds + bo[j1 & 1][2 * ks1 + pl] + t1 + (j1 >> 1) * 8192);
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int mj = 4 * p + i;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[n2 % RD][i][p == 0 ? 1 : 0], bq[g % 3][p == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
              if (mj == 5 || mj == 10) {
                const int u = 2 * j + (mj == 10 ? 1 : 0);
                fa[(n2 + RD - 1) % RD][u >> 1][u & 1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (wofs + ((it0 + n2 + RD - 1) * 8 + u) * 1024) & ((1 << 20) - 1), 0));
              }
              PIN();
            }
        }
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int q = 0; q < 3; ++q) for (int i = 0; i < 8; ++i) s += (float)ring[q][i].x;
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
void run(const unsigned char* w, float* out, long long* cyc, int iters, const char* what, double per_iter) {
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(1024);
  hipMemcpy(h.data(), cyc, 1024 * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double n = (MODE == 30 ? 64.0 : (double)iters) * per_iter;
  printf("mode %2d  %-58s median %9lld cycles  = %7.2f per unit\n", MODE, what, h[512], h[512] / n);
}

int main() {
  unsigned char* w; float* out; long long* cyc;
  hipMalloc(&w, 1 << 20); hipMemset(w, 0x11, 1 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  const int it = 66;
  run<0>(w, out, cyc, it, "48 MFMA / iteration (per MFMA)", 48);
  run<1>(w, out, cyc, it, "+ 8 ds_read_b128 (per MFMA)", 48);
  run<2>(w, out, cyc, it, "+ 8 buffer_load_dwordx4 (per MFMA)", 48);
  run<3>(w, out, cyc, it, "+ both (per MFMA)", 48);
  run<8>(w, out, cyc, 60, "k_ffn128 GEMM 1 k-step replica (per MFMA)", 48);
  run<9>(w, out, cyc, 60, "the same, 60 k-steps straight-line, executed once (per MFMA)", 48);
  run<40>(w, out, cyc, 60, "the same (hot loop), each wave its own weight slice (per MFMA)", 48);
  run<42>(w, out, cyc, 60, "as 40, workgroups de-phased along the weight stream (per MFMA)", 48);
  run<43>(w, out, cyc, 60, "as 40, two waves per slice = half the traffic (per MFMA)", 48);
  if (0) run<41>(w, out, cyc, 84, "the same, six k-steps of weights in flight (per MFMA)", 48);
  run<30>(w, out, cyc, it, "64 x 48 MFMA straight-line, executed once (per MFMA)", 48);
  run<4>(w, out, cyc, it, "VALU only, v_fma_f32 (per instruction)", 48);
  run<5>(w, out, cyc, it, "VALU only, v_pk_fma_f32 (per instruction)", 48);
  run<6>(w, out, cyc, it, "VALU only, v_fmaak_f32 literal (per instruction)", 48);
  run<7>(w, out, cyc, it, "VALU only, v_exp_f32 (per instruction)", 48);
  run<11>(w, out, cyc, it, "MFMA + 1 v_fma (per MFMA)", 48);
  run<12>(w, out, cyc, it, "MFMA + 2 v_fma (per MFMA)", 48);
  run<13>(w, out, cyc, it, "MFMA + 3 v_fma (per MFMA)", 48);
  run<14>(w, out, cyc, it, "MFMA + 4 v_fma (per MFMA)", 48);
  run<15>(w, out, cyc, it, "MFMA + 5 v_fma (per MFMA)", 48);
  run<16>(w, out, cyc, it, "MFMA + 6 v_fma (per MFMA)", 48);
  run<18>(w, out, cyc, it, "MFMA + 8 v_fma (per MFMA)", 48);
  run<21>(w, out, cyc, it, "MFMA + 1 v_pk_fma (per MFMA)", 48);
  run<22>(w, out, cyc, it, "MFMA + 2 v_pk_fma (per MFMA)", 48);
  run<24>(w, out, cyc, it, "MFMA + 4 v_pk_fma (per MFMA)", 48);
  return 0;
}
