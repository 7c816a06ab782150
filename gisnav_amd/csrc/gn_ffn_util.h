// Helpers shared by the fused block-tail kernels (gn_ffn.hip: 64 / 32 tokens per workgroup; gn_ffn128.hip: 128 tokens per workgroup).
#pragma once
#include "gn_common.h"

namespace gn {
namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int YP = 260;                 // float pitch of the output tile staged for the row-wise epilogue (aliases the hidden tile)

__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }

// Two-wide f32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 work on an aligned register pair at the rate of a single
// v_fma_f32: the VALU phases between the GEMMs are issue-bound with two waves per SIMD).  Written out by hand: this file is built
// without the SLP vectoriser.  Per element the operations and their order are those of gelu_erf (gn_common.h): same bits.
__device__ __forceinline__ f32x2v pair(const f32x16& v, int r) { return (f32x2v){v[r], v[r + 1]}; }
__device__ __forceinline__ void set_pair(f32x16& v, int r, f32x2v p) { v[r] = p[0]; v[r + 1] = p[1]; }
__device__ __forceinline__ f32x2v splat2(float c) { return (f32x2v){c, c}; }
__device__ __forceinline__ f32x2v gelu_erf2(f32x2v y) {
  const f32x2v ys = y * splat2(0.70710678118654752440f);
  const f32x2v t = {fminf(fabsf(ys[0]), 4.0f), fminf(fabsf(ys[1]), 4.0f)};
  f32x2v q = splat2(4.6081331674940884e-05f);
  q = q * t + splat2(-0.00045161080197431147f);
  q = q * t + splat2(0.0015096671413630247f);
  q = q * t + splat2(0.0007409505778923631f);
  q = q * t + splat2(-0.028223754838109016f);
  q = q * t + splat2(0.1484677642583847f);
  q = q * t + splat2(0.918419361114502f);
  q = q * t + splat2(1.6279083490371704f);
  const f32x2v qt = q * t;
  const f32x2v ex = {__builtin_amdgcn_exp2f(-qt[0]), __builtin_amdgcn_exp2f(-qt[1])};
  const f32x2v e = splat2(1.0f) - ex;
  const f32x2v cs = {copysignf(e[0], y[0]), copysignf(e[1], y[1])};
  return (splat2(0.5f) * y) * (splat2(1.0f) + cs);
}

// 8 f32 -> 8 fp16 high terms and the 8 fp16 residual terms (round to nearest), as two 16-byte fragments
__device__ __forceinline__ void split8(const float* v, uint4& h, uint4& m) {
  unsigned int hw[4], mw[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x2v x = {v[2 * e], v[2 * e + 1]};
    const f16x2v hv = __builtin_convertvector(x, f16x2v);
    const f32x2v r = x - __builtin_convertvector(hv, f32x2v);
    const f16x2v mv = __builtin_convertvector(r, f16x2v);
    hw[e] = __builtin_bit_cast(unsigned int, hv);
    mw[e] = __builtin_bit_cast(unsigned int, mv);
  }
  h = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  m = make_uint4(mw[0], mw[1], mw[2], mw[3]);
}
}  // namespace
}  // namespace gn
