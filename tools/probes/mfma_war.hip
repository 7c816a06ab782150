// Probe: is a VGPR that an MFMA reads as SrcA safe to overwrite right behind the MFMA?  (gfx950, v_mfma_f32_32x32x16_f16)
// Two MFMAs with the same A / B fragments are issued back to back (the second waits for the matrix pipe), then NV VALU instructions later the
// A fragment's registers are overwritten with zeros.  Both accumulators must hold A.B; a mismatch means the queued MFMA read the overwritten
// registers.  The compiler never sees this situation for MFMAs it generates (it knows their operands) -- an MFMA written in inline assembly does.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_war mfma_war.hip ; run: ./mfma_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define BODY(DSTC, GAP)                                                                                                  \
  asm volatile("v_mov_b32 v100, %[a0]\n\tv_mov_b32 v101, %[a1]\n\tv_mov_b32 v102, %[a2]\n\tv_mov_b32 v103, %[a3]\n\t"    \
               "s_nop 7\n\t"                                                                                             \
               "v_mfma_f32_32x32x16_f16 %[c0], v[100:103], %[b], %[c0]\n\t"                                              \
               "v_mfma_f32_32x32x16_f16 %[c1], v[100:103], %[b], %[c1]\n\t" GAP                                          \
               "v_mov_b32 v103, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v100, 0\n\t"                    \
               "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"                    \
               : [c0] "+" DSTC(c0), [c1] "+" DSTC(c1)                                                                    \
               : [a0] "v"(a.x), [a1] "v"(a.y), [a2] "v"(a.z), [a3] "v"(a.w), [b] "v"(b)                                  \
               : "v100", "v101", "v102", "v103");

template <int MODE> __global__ void k(const u32x4* A, const s16x8* B, float* out, int iters) {
  const int lane = threadIdx.x;
  const u32x4 a = A[lane];
  const s16x8 b = B[lane];
  f32x16 c0, c1;
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { BODY("v", "") }
    else if (MODE == 1) { BODY("v", "v_mov_b32 v104, 0\n\t") }
    else if (MODE == 2) { BODY("v", "v_mov_b32 v104, 0\n\tv_mov_b32 v104, 0\n\t") }
    else if (MODE == 4) { BODY("v", "v_mov_b32 v104, 0\n\tv_mov_b32 v104, 0\n\tv_mov_b32 v104, 0\n\tv_mov_b32 v104, 0\n\t") }
    else if (MODE == 8) { BODY("v", "s_nop 7\n\t") }
    else if (MODE == 16) { BODY("v", "s_nop 7\n\ts_nop 7\n\t") }
    else if (MODE == 100) { BODY("a", "") }
    else if (MODE == 104) { BODY("a", "v_mov_b32 v104, 0\n\tv_mov_b32 v104, 0\n\tv_mov_b32 v104, 0\n\tv_mov_b32 v104, 0\n\t") }
  }
  for (int r = 0; r < 16; ++r) { out[(r * 2 + 0) * 64 + lane] = c0[r]; out[(r * 2 + 1) * 64 + lane] = c1[r]; }
}

template <int MODE> void run(const u32x4* A, const s16x8* B, float* out, const char* what) {
  std::vector<float> h(32 * 64);
  int bad_launch = 0; double worst = 0;
  for (int rep = 0; rep < 200; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, A, B, out, 64);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    bool bad = false;
    for (int r = 0; r < 16; ++r) for (int l = 0; l < 64; ++l) { const double d = fabs((double)h[(r * 2) * 64 + l] - h[(r * 2 + 1) * 64 + l]); if (d != 0) { bad = true; if (d > worst) worst = d; } }
    bad_launch += bad;
  }
  printf("%-70s: %d of 200 launches with c1 != c0 (max |diff| %.3g)\n", what, bad_launch, worst);
}

int main() {
  std::vector<unsigned> a(64 * 4); std::vector<short> b(64 * 8);
  for (size_t i = 0; i < a.size(); ++i) a[i] = 0x3c003c00u;            // fp16 (1, 1)
  for (size_t i = 0; i < b.size(); ++i) b[i] = (short)0x3c00;
  u32x4* A; s16x8* B; float* out;
  hipMalloc(&A, a.size() * 4); hipMalloc(&B, b.size() * 2); hipMalloc(&out, 32 * 64 * 4);
  hipMemcpy(A, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, b.data(), b.size() * 2, hipMemcpyHostToDevice);
  run<0>(A, B, out, "VGPR accumulators, A overwritten right behind the second MFMA");
  run<1>(A, B, out, "VGPR accumulators, 1 VALU instruction in between");
  run<2>(A, B, out, "VGPR accumulators, 2 VALU instructions in between");
  run<4>(A, B, out, "VGPR accumulators, 4 VALU instructions in between");
  run<8>(A, B, out, "VGPR accumulators, s_nop 7 in between");
  run<16>(A, B, out, "VGPR accumulators, 2 x s_nop 7 in between");
  run<100>(A, B, out, "AGPR accumulators, A overwritten right behind the second MFMA");
  run<104>(A, B, out, "AGPR accumulators, 4 VALU instructions in between");
  return 0;
}
