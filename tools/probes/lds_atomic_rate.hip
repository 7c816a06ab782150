// Issue rate of back-to-back LDS float atomics from one wave (the SIFT descriptor committer's inner loop).
//   hipcc --offload-arch=gfx950 -O2 lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(long long* out, float* sink) {
  __shared__ float h[512];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) h[i] = 0.f;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)h;
  unsigned addr = base + 4u * (unsigned)((lane * 7) & 255);
  float v = 1.0f + lane;
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    if (MODE == 0) {          // 8 x full-wave atomics, distinct addresses
      asm volatile("ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n"
                   "ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n" ::"v"(addr), "v"(v) : "memory");
    } else if (MODE == 1) {   // 8 x 8-lane atomics, exec rewritten between them
      unsigned long long sv;
      asm volatile("s_mov_b64 %0, exec\n s_mov_b32 exec_hi, 0\n"
                   "s_mov_b32 exec_lo, 0xff\n ds_add_f32 %1, %2\n s_mov_b32 exec_lo, 0xff00\n ds_add_f32 %1, %2\n"
                   "s_mov_b32 exec_lo, 0xff0000\n ds_add_f32 %1, %2\n s_mov_b32 exec_lo, 0xff000000\n ds_add_f32 %1, %2\n"
                   "s_mov_b32 exec_lo, 0\n s_mov_b32 exec_hi, 0xff\n ds_add_f32 %1, %2\n s_mov_b32 exec_hi, 0xff00\n ds_add_f32 %1, %2\n"
                   "s_mov_b32 exec_hi, 0xff0000\n ds_add_f32 %1, %2\n s_mov_b32 exec_hi, 0xff000000\n ds_add_f32 %1, %2\n"
                   "s_mov_b64 exec, %0\n" : "=&s"(sv) : "v"(addr), "v"(v) : "memory");
    } else if (MODE == 2) {   // 8 x plain ds_write_b32 (no atomic) for comparison
      asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n"
                   "ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n ds_write_b32 %0, %1\n" ::"v"(addr), "v"(v) : "memory");
    } else if (MODE == 3) {   // 8 x full-wave atomics + a dependent read each iteration (the committer's pattern)
      float r;
      asm volatile("ds_add_f32 %1, %2\n ds_add_f32 %1, %2\n ds_add_f32 %1, %2\n ds_add_f32 %1, %2\n"
                   "ds_add_f32 %1, %2\n ds_add_f32 %1, %2\n ds_add_f32 %1, %2\n ds_add_f32 %1, %2\n"
                   "ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=v"(r) : "v"(addr), "v"(v) : "memory");
      v += r * 1e-30f;
    } else if (MODE == 4) {   // 8 x 8-lane atomics without touching exec: lanes >= 8 hit a far address... all lanes active, 8 distinct addresses x 8 lanes each (conflicts)
      const unsigned a2 = base + 4u * (unsigned)(lane & 7);
      asm volatile("ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n"
                   "ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n ds_add_f32 %0, %1\n" ::"v"(a2), "v"(v) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = clock64();
  if (lane == 0) out[0] = t1 - t0;
  sink[lane] = h[lane] + v;
}

template <int LANES>
__global__ void k2(long long* out, float* sink) {
  __shared__ float h[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2048; i += blockDim.x) h[i] = 0.f;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)h;
  unsigned addr = base + 4u * (unsigned)(wave * 256 + ((lane * 7) & 255));
  float v = 1.0f + lane;
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    unsigned long long sv;
    if (LANES == 8)
      asm volatile("s_mov_b64 %0, exec\n s_mov_b32 exec_hi, 0\n"
                   "s_mov_b32 exec_lo, 0xff\n ds_add_f32 %1, %2\n s_mov_b32 exec_lo, 0xff00\n ds_add_f32 %1, %2\n"
                   "s_mov_b32 exec_lo, 0xff0000\n ds_add_f32 %1, %2\n s_mov_b32 exec_lo, 0xff000000\n ds_add_f32 %1, %2\n"
                   "s_mov_b32 exec_lo, 0\n s_mov_b32 exec_hi, 0xff\n ds_add_f32 %1, %2\n s_mov_b32 exec_hi, 0xff00\n ds_add_f32 %1, %2\n"
                   "s_mov_b32 exec_hi, 0xff0000\n ds_add_f32 %1, %2\n s_mov_b32 exec_hi, 0xff000000\n ds_add_f32 %1, %2\n"
                   "s_mov_b64 exec, %0\n" : "=&s"(sv) : "v"(addr), "v"(v) : "memory");
    else
      asm volatile("s_mov_b64 %0, exec\n s_mov_b32 exec_hi, 0\n"
                   "s_mov_b32 exec_lo, 0xf\n ds_add_f32 %1, %2\n s_mov_b32 exec_lo, 0xf00\n ds_add_f32 %1, %2\n"
                   "s_mov_b32 exec_lo, 0xf0000\n ds_add_f32 %1, %2\n s_mov_b32 exec_lo, 0xf000000\n ds_add_f32 %1, %2\n"
                   "s_mov_b32 exec_lo, 0\n s_mov_b32 exec_hi, 0xf\n ds_add_f32 %1, %2\n s_mov_b32 exec_hi, 0xf00\n ds_add_f32 %1, %2\n"
                   "s_mov_b32 exec_hi, 0xf0000\n ds_add_f32 %1, %2\n s_mov_b32 exec_hi, 0xf000000\n ds_add_f32 %1, %2\n"
                   "s_mov_b64 exec, %0\n" : "=&s"(sv) : "v"(addr), "v"(v) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = clock64();
  if (lane == 0) out[wave] = t1 - t0;
  sink[tid] = h[tid] + v;
}
int main() {
  long long* d; float* s; hipMalloc(&d, 8); hipMalloc(&s, 256);
  const char* names[] = {"full-wave ds_add_f32", "8-lane ds_add_f32 + exec writes", "ds_write_b32", "ds_add_f32 x8 + dependent read", "full-wave, 8-way same-address"};
  for (int m = 0; m < 5; ++m) {
    long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, s);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, s);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, s);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, s);
      if (m == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, s);
      hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    }
    printf("%-36s %8.1f clock64 ticks per instruction (2048 instructions)\n", names[m], (double)h / 2048.0);
  }
  long long* d8; float* s8; hipMalloc(&d8, 64); hipMalloc(&s8, 4096);
  for (int W = 1; W <= 8; W *= 2)
    for (int L = 8; L >= 4; L -= 4) {
      long long h[8] = {0};
      for (int rep = 0; rep < 2; ++rep) {
        if (L == 8) hipLaunchKernelGGL(k2<8>, dim3(1), dim3(64 * W), 0, 0, d8, s8); else hipLaunchKernelGGL(k2<4>, dim3(1), dim3(64 * W), 0, 0, d8, s8);
        hipMemcpy(h, d8, 8 * W, hipMemcpyDeviceToHost);
      }
      printf("%d wave(s) concurrently, %d-lane ds_add_f32: %.1f ticks per instruction per wave\n", W, L, (double)h[0] / 2048.0);
    }
  // tick calibration: clock64 vs wall time of a long kernel is not needed; report s_memtime rate via hipDeviceAttributeWallClockRate
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0); int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, shader clock %d kHz\n", rate, clk);
  return 0;
}
