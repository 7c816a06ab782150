// Developer probe (not part of the library): the "two 64-token halves" form of the block tail priced BEFORE building it.
// 512-thread workgroups, one per CU: two waves per SIMD, 256 registers each (128 accumulators + 128 others).  Wave w = (slice wv = w & 3, token half th = w >> 2):
// the two waves of a SIMD own the SAME 128 hidden units (same weight fragments) and different 64 tokens.  Questions:
//   1  GEMM-1 k-step replica, 24 MFMAs per wave and k-step, 8 weight fragments through a register ring of RA k-steps, pair shares the slice:
//      does the second wave's request for the same lines hit in the vector L1 (no second L2 -> CU transfer), i.e. does the pair run at the
//      matrix-pipe floor of 2 x 24 x 32 cycles per k-step?          2  the same, every wave its own slice (twice the L2 -> CU traffic)
//   3  one wave per SIMD, 48 MFMAs per k-step (k_ffn128's shape) for reference
//   4  VALU-only phase (8 independent fma per stage, 22 stages: a GELU fragment), two waves per SIMD      5  the same, one wave per SIMD
//   6  the GELU-inside-GEMM-2 mix: 24 MFMAs + 22 stages x 8 VALU per wave, two waves per SIMD             7  the same mix, one wave per SIMD (48 MFMAs + 2 x 22 stages)
//   8  OUT OF PHASE: the th = 0 wave of every SIMD runs MFMA-only k-steps (24 MFMAs, register operands) while the th = 1 wave runs VALU-only stages;
//      9 / 10: the same two instruction streams alone (the other wave of the SIMD idle) -- do an MFMA wave and a VALU wave share a SIMD for free?
//   11 TWO INDEPENDENT 4-wave WORKGROUPS PER CU (<= 256 registers, launch_bounds(256, 2)), each alternating an MFMA phase (48 x 24 MFMAs, ~37 k cycles: the
//      two GEMMs of a 64-token tile) and a VALU phase (1200 stages x 8 fma, ~27 k cycles: statistics, GELU, epilogue arithmetic), no synchronisation between
//      them: wall time of the launch at 1 and 2 workgroups per CU -- do they drift out of phase and overlap?
//   12 / 13  k_attn_pw's mix per 64 keys (40 MFMAs, 160 fma-class + 32 exp2 instructions) on one wave per SIMD / split over two waves per SIMD (20 MFMAs +
//      80 + 16 each): what a partner wave could buy the attention kernel
// cycles are s_memtime per workgroup (median over the grid), reported per k-step / per stage.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/stream2w.hip -o tools/probes/stream2w
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define PIN() __builtin_amdgcn_sched_barrier(0)

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void probe(const unsigned char* w, float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += NT) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(w), 0, 1 << 21, 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool TWO = NT == 512;
  constexpr int NJ = TWO ? 2 : 4;
  f32x16 acc[4][NJ];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 0.001f + i;
  const float c1 = 0.999f + 1e-6f * iters, c2 = 0.001f;
  const int ql = lane & 31, hh = lane >> 5;
  long long t0 = 0, t1 = 0;
  if (MODE <= 3) {
    // weight slice: 256 KB per slice (32 k-steps x 8 fragments x 1 KB), cyclic
    const int slice = (MODE == 2) ? wv : (wv & 3);
    const int wofs = slice * 262144;
    constexpr int RA = TWO ? 3 : 4;
    f16x8 fa[RA][4][2], bq[3][2];
    unsigned bo[2][4];
    for (int jp = 0; jp < 2; ++jp) for (int c = 0; c < 4; ++c) { const int r = 32 * jp + ql; bo[jp][c] = r * 128 + (((2 * c + hh) ^ ((r ^ (r >> 3)) & 7)) * 16) + (TWO ? (wv >> 2) * 8192 : 0); }
    for (int q = 0; q < RA - 1; ++q) for (int i = 0; i < 4; ++i) for (int pl = 0; pl < 2; ++pl)
      fa[q][i][pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, wofs + ((q * 8 + 2 * i + pl) * 1024), 0));
    for (int pl = 0; pl < 2; ++pl) bq[0][pl] = *reinterpret_cast<const f16x8*>(lds + bo[0][pl]);
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    constexpr int TRIP = 12;
    for (int it0 = 0; it0 < iters; it0 += TRIP) {
#pragma unroll
      for (int n2 = 0; n2 < TRIP; ++n2) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int g = NJ * n2 + j, g1 = g + 1, j1 = g1 % NJ, ks1 = (g1 / NJ) & 1, tl = ((g1 / NJ / 2) & 3) * 16384;
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) bq[g1 % 3][pl] = *reinterpret_cast<const f16x8*>(lds + bo[j1 & 1][2 * ks1 + pl] + tl + (TWO ? 0 : (j1 >> 1) * 8192));
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int mj = 4 * p + i;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[n2 % RA][i][p == 0 ? 1 : 0], bq[g % 3][p == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
              // 8 refills per k-step: one wave per SIMD: gaps 5, 10 of each of the 4 j-steps; two waves: gaps 1, 4, 7, 10 of each of the 2 j-steps
              const bool ld = TWO ? (mj == 1 || mj == 4 || mj == 7 || mj == 10) : (mj == 5 || mj == 10);
              if (ld) {
                const int u = TWO ? 4 * j + (mj - 1) / 3 : 2 * j + (mj == 10 ? 1 : 0);
                fa[(n2 + RA - 1) % RA][u >> 1][u & 1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (wofs + (((it0 + n2 + RA - 1) & 31) * 8 + u) * 1024), 0));
              }
              PIN();
            }
        }
        if (MODE != 3 || true) __syncthreads();       // the token-ring barrier of the real k-loop
        PIN();
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else if (MODE == 4 || MODE == 5) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int st = 0; st < 22; ++st) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);
        PIN();
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else if (MODE == 12 || MODE == 13) {
    f16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.01f * (lane + e + i));
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)(0.02f * (lane - e + i));
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < (TWO ? 20 : 40); ++m) {
        acc[m & 3][(m >> 2) % NJ] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[m & 1], acc[m & 3][(m >> 2) % NJ], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);                       // 4 fma-class instructions per gap
        if ((m % 5) != 4) v[4 + (m & 3)] = __builtin_amdgcn_exp2f(v[4 + (m & 3)] * 1e-3f);    // 4 of 5 gaps: one exp2 (+ its scaling multiply)
        PIN();
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else if (MODE >= 8 && MODE <= 10) {
    f16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.01f * (lane + e + i));
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)(0.02f * (lane - e + i));
    const bool mf = (wv >> 2) == 0;
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    if (mf && MODE != 10) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) { acc[m & 3][(m >> 2) % NJ] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[m & 1], acc[m & 3][(m >> 2) % NJ], 0, 0, 0); PIN(); }
      }
    } else if (!mf && MODE != 9) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 35; ++st) {       // 35 stages x ~22 cycles ~ 24 MFMAs x 32 cycles: the two streams take about as long alone
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);
          PIN();
        }
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else {
    f16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.01f * (lane + e + i));
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)(0.02f * (lane - e + i));
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      // one (quarter, k-step) of GEMM 2: TWO: 24 MFMAs + one fragment's 22 stages; else 48 MFMAs + two fragments' stages (22 of every 24 gaps)
#pragma unroll
      for (int m = 0; m < (TWO ? 24 : 48); ++m) {
        acc[m & 3][(m >> 2) % NJ] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[m & 1], acc[m & 3][(m >> 2) % NJ], 0, 0, 0);
        if (m % 24 < 22) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);
        }
        PIN();
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * NT + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

__global__ __launch_bounds__(256, 2) void probe_phases(float* out, long long* cyc, int tiles, int skew) {
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 0.001f + i;
  const float c1 = 0.999f + 1e-6f * tiles, c2 = 0.001f;
  f16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.01f * (lane + e + i));
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)(0.02f * (lane - e + i));
  const long long t0 = __builtin_amdgcn_s_memtime();
  // (skew: odd workgroups start with half a VALU phase, so that co-resident workgroups do not start in lock-step)
  if (skew && (blockIdx.x & 1)) {
    for (int st = 0; st < 600; ++st) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);
      PIN();
    }
  }
  for (int t = 0; t < tiles; ++t) {
    for (int ks = 0; ks < 48; ++ks) {
#pragma unroll
      for (int m = 0; m < 24; ++m) { acc[m & 3][(m >> 2) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[m & 1], acc[m & 3][(m >> 2) & 1], 0, 0, 0); PIN(); }
    }
    for (int st = 0; st < 1200; ++st) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);
      PIN();
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[(blockIdx.x % 256) * 256 + tid] = s;
  if (lane == 0) cyc[(blockIdx.x % 256) * 8 + (tid >> 6)] = t1 - t0;
}

template <int MODE, int NT>
void run(const unsigned char* w, float* out, long long* cyc, int iters, const char* what, double units, double floor_) {
  hipMemset(cyc, 0, 256 * 8 * 8);
  hipLaunchKernelGGL((probe<MODE, NT>), dim3(256), dim3(NT), 0, 0, w, out, cyc, iters);
  hipLaunchKernelGGL((probe<MODE, NT>), dim3(256), dim3(NT), 0, 0, w, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(2048), u;
  hipMemcpy(h.data(), cyc, 2048 * sizeof(long long), hipMemcpyDeviceToHost);
  for (long long x : h) if (x > 0) u.push_back(x);
  std::sort(u.begin(), u.end());
  double med = (double)u[u.size() / 2];
  if (MODE >= 8 && MODE <= 10) {      // the two kinds of waves separately: slots 0..3 of a workgroup = MFMA waves, 4..7 = VALU waves
    std::vector<long long> m, vv;
    for (int b = 0; b < 256; ++b) for (int k = 0; k < 8; ++k) (k < 4 ? m : vv).push_back(h[b * 8 + k]);
    std::sort(m.begin(), m.end()); std::sort(vv.begin(), vv.end());
    printf("mode %d: MFMA waves median %lld cycles (%.1f per 24 MFMAs), VALU waves median %lld cycles (%.1f per 35 stages)\n", MODE, m[512], m[512] / units, vv[512], vv[512] / units);
    return;
  }
  printf("mode %d (%d threads)  %-78s %9.0f cycles = %8.1f per unit (floor %.0f: %.2f x)\n", MODE, NT, what, med, med / units, floor_, med / units / floor_);
}

int main() {
  unsigned char* w; float* out; long long* cyc;
  hipMalloc(&w, 1 << 21); hipMemset(w, 0x11, 1 << 21); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int it = 96;
  run<3, 256>(w, out, cyc, it, "one wave / SIMD, 48 MFMAs per k-step, own slice (k_ffn128's GEMM 1), per k-step", it, 48 * 32.0);
  run<1, 512>(w, out, cyc, it, "two waves / SIMD, 24 MFMAs each, the pair SHARES its weight slice, per k-step", it, 48 * 32.0);
  run<2, 512>(w, out, cyc, it, "two waves / SIMD, 24 MFMAs each, every wave its own slice (2 x traffic), per k-step", it, 48 * 32.0);
  run<5, 256>(w, out, cyc, 200, "VALU only, one wave / SIMD: 22 stages x 8 fma, per stage", 200 * 22, 16.0);
  run<4, 512>(w, out, cyc, 200, "VALU only, two waves / SIMD: 22 stages x 8 fma each, per stage (of one wave)", 200 * 22, 16.0);
  run<7, 256>(w, out, cyc, 100, "GEMM 2 + GELU mix, one wave / SIMD: 48 MFMAs + 44 stages x 8 fma, per 48 MFMAs", 100, 48 * 32.0);
  run<6, 512>(w, out, cyc, 100, "GEMM 2 + GELU mix, two waves / SIMD: 24 MFMAs + 22 stages x 8 fma each, per 48 MFMAs", 100, 48 * 32.0);
  run<12, 256>(w, out, cyc, 200, "attention mix, one wave / SIMD: 40 MFMAs + 160 fma + 32 (mul, exp2), per 64 keys", 200, 40 * 32.0);
  run<13, 512>(w, out, cyc, 200, "attention mix, two waves / SIMD: 20 MFMAs + 80 fma + 16 (mul, exp2) each, per 64 keys", 200, 40 * 32.0);
  for (int wgs = 256; wgs <= 512; wgs += 256)
    for (int skew = 0; skew < 2; ++skew) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(probe_phases, dim3(wgs), dim3(256), 0, 0, out, cyc, 8, skew);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(probe_phases, dim3(wgs), dim3(256), 0, 0, out, cyc, 8, skew);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
      printf("mode 11: %d workgroups (%d per CU), skew %d: 8 tiles of (37 k MFMA + 27 k VALU cycles) each: %.1f us per launch = %.2f us per 64-token tile-equivalent per CU\n",
             wgs, wgs / 256, skew, ms * 1e3, ms * 1e3 / (8.0 * wgs / 256));
    }
  run<9, 512>(w, out, cyc, 200, "MFMA waves alone", 200, 768.0);
  run<10, 512>(w, out, cyc, 200, "VALU waves alone", 200, 768.0);
  run<8, 512>(w, out, cyc, 200, "MFMA wave + VALU wave on every SIMD", 200, 768.0);
  return 0;
}
