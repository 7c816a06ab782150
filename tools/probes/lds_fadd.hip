// Does the LDS float atomic (ds_add_f32) add exactly like v_add_f32 (round-to-nearest-even, subnormals kept)?
// The SIFT descriptor kernel relies on it for bit-identical histograms.   hipcc --offload-arch=gfx950 -O2 lds_fadd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cstdint>
__global__ void k(const float* a, const float* b, float* lds_out, float* valu_out, int n) {
  __shared__ float acc[256];
  const int t = threadIdx.x;
  for (int i = t; i < n; i += 256) {
    acc[t] = a[i];
    __hip_atomic_fetch_add(&acc[t], b[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_out[i] = acc[t];
    valu_out[i] = a[i] + b[i];
  }
}
int main() {
  const int n = 1 << 20;
  float *ha = (float*)malloc(n * 4), *hb = (float*)malloc(n * 4), *h1 = (float*)malloc(n * 4), *h2 = (float*)malloc(n * 4);
  srand(7);
  for (int i = 0; i < n; ++i) {
    uint32_t ua, ub;
    const int mode = i & 3;
    if (mode == 0) { ua = (rand() & 0x7fffff); ub = (rand() & 0x7fffff); }                       // subnormal + subnormal
    else if (mode == 1) { ua = 0x3f800000u | (rand() & 0x7fffff); ub = ((uint32_t)(100 + rand() % 28) << 23) | (rand() & 0x7fffff); }   // ties / sticky bits
    else if (mode == 2) { ua = ((uint32_t)(1 + rand() % 3) << 23) | (rand() & 0x7fffff); ub = (rand() & 0x7fffff); }   // tiny normal + subnormal
    else { ua = ((uint32_t)(90 + rand() % 60) << 23) | (rand() & 0x7fffff); ub = ((uint32_t)(90 + rand() % 60) << 23) | (rand() & 0x7fffff); }
    memcpy(&ha[i], &ua, 4); memcpy(&hb[i], &ub, 4);
  }
  float *da, *db, *d1, *d2;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
  hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, da, db, d1, d2, n);
  hipMemcpy(h1, d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 4, hipMemcpyDeviceToHost);
  int bad_lds[4] = {0, 0, 0, 0}, bad_valu[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const float ref = ha[i] + hb[i];                    // host IEEE add
    if (memcmp(&ref, &h1[i], 4)) bad_lds[i & 3]++;
    if (memcmp(&ref, &h2[i], 4)) bad_valu[i & 3]++;
  }
  printf("lds_fadd mismatches vs host IEEE add  [subnormal, ties, tiny+subnormal, normal]: %d %d %d %d\n", bad_lds[0], bad_lds[1], bad_lds[2], bad_lds[3]);
  printf("v_add_f32 mismatches vs host IEEE add [subnormal, ties, tiny+subnormal, normal]: %d %d %d %d\n", bad_valu[0], bad_valu[1], bad_valu[2], bad_valu[3]);
  return 0;
}
