// Developer probe (not part of the library): semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 that the SuperPoint convolution's
// halo staging relies on:  (a) LDS destination = M0 + instruction offset + lane * 16, also for M0 above 64 KB;  (b) a lane whose offset is
// outside the descriptor's range writes ZEROS to its LDS slot (it does not skip it);  (c) an SGPR soffset is added to the address but not
// to the range check.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/buf_lds.hip -o tools/probes/buf_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ __launch_bounds__(64) void probe(const float* src, float* out, int nbytes, int m0off) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[81920];
  const int lane = threadIdx.x;
  for (int i = lane; i < 81920 / 4; i += 64) reinterpret_cast<float*>(lds)[i] = -7.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds + (unsigned)m0off;
  // lanes 0..31: in range (reversed order: lane i reads chunk 31 - i); lanes 32..47: offset beyond the range; lanes 48..63: 0xffffff00
  unsigned voff = lane < 32 ? (31 - lane) * 16 : (lane < 48 ? (unsigned)nbytes + (lane - 32) * 16 : 0xffffff00u);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(rs), "s"(lds0) : "memory");
  // second instruction: soffset 1024 (bytes), instruction offset 16 -> LDS destination + 16 + 1024?  (the offset field applies to both sides)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen offset:16 lds\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(rs), "s"(lds0 + 2048u), "s"(1024) : "memory");
  // third instruction: EXEC = lanes 0..7 only, LDS + 4096: do the masked lanes leave their slots alone?  (d)
  asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, 0xff\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_mov_b64 exec, -1\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(rs), "s"(lds0 + 4096u) : "memory");
  __syncthreads();
  for (int i = lane; i < 8192 / 4; i += 64) out[i] = reinterpret_cast<float*>(lds + m0off)[i];
}
int main() {
  const int n = 4096;   // floats in the source; the descriptor covers the first 512 bytes + 1024 ... see nbytes
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 8192 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int m0off : {0, 70000 / 16 * 16}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, 2048, m0off);
    std::vector<float> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    printf("m0 offset %d: first instruction, first float of each lane's 16-byte slot:\n", m0off);
    for (int l = 0; l < 64; ++l) printf("%g ", r[l * 4]);
    printf("\nsecond instruction (m0 + 2048, soffset 1024, offset:16): floats at LDS +2048.. (slot of lane l at +2048 + 16 l [+16?]):\n");
    for (int l = 0; l < 68; ++l) printf("%g ", r[512 + l * 4]);
    printf("\nthird instruction (EXEC = lanes 0..7), LDS + 4096, first float of each lane's slot (masked lanes should keep -7):\n");
    for (int l = 0; l < 64; ++l) printf("%g ", r[1024 + l * 4]);
    printf("\n");
  }
  return 0;
}
