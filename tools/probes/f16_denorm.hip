// Developer probe: does v_mfma_f32_32x32x16_f16 honour subnormal fp16 inputs on gfx950? and f32->f16 convert codegen
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, float av, float bv) {
  f16x8 a, b;
  f32x2 t; t[0] = av; t[1] = av;
  f16x2 ah = __builtin_convertvector(t, f16x2);
  for (int e = 0; e < 8; ++e) { a[e] = ah[e & 1]; b[e] = (_Float16)bv; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)ah[0]; }
}
int main() {
  float* d; (void)hipMalloc(&d, 64);
  float h[2];
  const float avs[] = {1.0f, 6.2e-5f, 3.0e-5f, 9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24 */};
  for (float av : avs) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, av, 1024.0f);
    (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a=%.6e  cvt->f16->f32=%.6e  mfma sum over k=16 of a*1024 = %.6e (expected %.6e)\n", av, h[1], h[0], 16.0 * h[1] * 1024.0);
  }
  return 0;
}
