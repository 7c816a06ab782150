// Input preparation and the row-wise pieces of the LightGlue blocks (HBM-bound, one pass each).
//
//   k_extent        keypoint extent (max_x, max_y) per (pair, side): kornia's LightGlueMatcher passes
//                   image_size = keypoints.max(dim=1) when hw is None (call site pose_node.py:285-287)
//   k_prep          RootSIFT (pose_node.py:278-284), LAF -> (centre, scale, orientation)
//                   (pose_node.py:267-276 + kornia get_laf_*), normalize_keypoints, and the learnable
//                   Fourier positional encoding cos/sin(Wr [x^,y^,scale,ori]) cached for all 9 layers
//   k_ln_gelu       LayerNorm(512, eps 1e-5, affine) + exact-erf GELU of the FFN hidden (ffn.1, ffn.2)
//   k_matchability  logsigmoid(Linear(256 -> 1)) per keypoint (MatchAssignment.matchability)
#include "gn_common.h"

namespace gn {

namespace {
constexpr float kPi = 3.14159265358979323846f;

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ inline void read_kpt(const float* kp, int fmt, float& x, float& y, float& scale, float& ori) {
  float a00, a01, a10, a11;
  if (fmt == GN_KPT_LAF) {
    a00 = kp[0]; a01 = kp[1]; x = kp[2]; a10 = kp[3]; a11 = kp[4]; y = kp[5];
  } else {
    x = kp[0]; y = kp[1];
    const bool rec = fmt == GN_KPT_RECORD;     // KEYPOINT_DTYPE record: x, y, z, size, angle, descriptor[128]
    const float size = kp[rec ? 3 : 2];
    const float ang = kp[rec ? 4 : 3] * kPi / 180.0f;  // kornia deg2rad
    const float c = cosf(ang), s = sinf(ang);
    a00 = size * c; a01 = size * s; a10 = size * (-s); a11 = size * c;
  }
  scale = sqrtf(fabsf(a00 * a11 - a10 * a01));             // get_laf_scale
  const float deg = 180.0f * atan2f(a01, a00) / kPi;       // get_laf_orientation (rad2deg)
  ori = deg * kPi / 180.0f;                                // deg2rad in LightGlueMatcher.forward
  if (ori < 0.f) ori += 2.0f * kPi;
}

__global__ __launch_bounds__(256) void k_extent(PrepArgs a) {
  const int bs = blockIdx.x, b = bs >> 1, side = bs & 1;
  const int n = min(side ? a.n_r[b] : a.n_q[b], a.npad);   // keypoints beyond the padded size of this call are ignored (gn_set_active_kpts)
  const int stride = side ? a.stride_r : a.stride_q;
  const int fmt = a.kpt_format & 0xff;
  const int w = fmt == GN_KPT_LAF ? 6 : fmt == GN_KPT_RECORD ? kRecordFloats : 4;
  const int xo = fmt == GN_KPT_LAF ? 2 : 0, yo = fmt == GN_KPT_LAF ? 5 : 1;
  const float* kp = (side ? a.kpt_r : a.kpt_q) + (size_t)b * stride * w;
  float mx = -INFINITY, my = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) {
    mx = fmaxf(mx, kp[(size_t)i * w + xo]);
    my = fmaxf(my, kp[(size_t)i * w + yo]);
  }
  mx = wave_max(mx); my = wave_max(my);
  __shared__ float sx[4], sy[4];
  if ((threadIdx.x & 63) == 0) { sx[threadIdx.x >> 6] = mx; sy[threadIdx.x >> 6] = my; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float* given = side ? a.size_r : a.size_q;     // hw passed to the matcher: image size instead of the keypoint extent
    a.extent[bs * 2 + 0] = given[0] > 0.f ? given[0] : fmaxf(fmaxf(sx[0], sx[1]), fmaxf(sx[2], sx[3]));
    a.extent[bs * 2 + 1] = given[1] > 0.f ? given[1] : fmaxf(fmaxf(sy[0], sy[1]), fmaxf(sy[2], sy[3]));
    a.nvalid[bs] = n;
  }
}

// one wave per token slot
__global__ __launch_bounds__(256) void k_prep(PrepArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bs = blockIdx.y, b = bs >> 1, side = bs & 1;
  const int i = blockIdx.x * 4 + wave;
  const int n = side ? a.n_r[b] : a.n_q[b];
  const int stride = side ? a.stride_r : a.stride_q;
  const size_t tok = (size_t)bs * a.npad + i;
  float* dout = a.desc + tok * kInDim;
  if (i >= n) {  // padding slot: finite, inert values
    dout[lane] = 0.f; dout[lane + 64] = 0.f;
    if (lane < kFreq) { a.cos_t[tok * kFreq + lane] = 1.f; a.sin_t[tok * kFreq + lane] = 0.f; }
    return;
  }
  const int fmt_ = a.kpt_format & 0xff;
  // GN_KPT_RECORD: the descriptor is the tail of the keypoint's own 532-byte wire record (4-byte aligned only: scalar loads)
  const float* din = fmt_ == GN_KPT_RECORD ? (side ? a.kpt_r : a.kpt_q) + ((size_t)b * stride + i) * kRecordFloats + 5
                                           : (side ? a.desc_r : a.desc_q) + ((size_t)b * stride + i) * kInDim;
  const float d0 = din[lane], d1 = din[lane + 64];
  if (a.kpt_format & GN_DESC_ROOTSIFT) {
    dout[lane] = d0; dout[lane + 64] = d1;
  } else {
    const float l1 = wave_sum(fabsf(d0) + fabsf(d1));
    const float den = fmaxf(l1, 1e-12f);  // F.normalize(p=1, eps=1e-12)
    dout[lane] = sqrtf(d0 / den);
    dout[lane + 64] = sqrtf(d1 / den);
  }

  if (lane < kFreq) {
    const int fmt = a.kpt_format & 0xff;
    const int w = fmt == GN_KPT_LAF ? 6 : fmt == GN_KPT_RECORD ? kRecordFloats : 4;
    const float* kp = (side ? a.kpt_r : a.kpt_q) + ((size_t)b * stride + i) * w;
    float x, y, scale, ori;
    read_kpt(kp, fmt, x, y, scale, ori);
    const float sx = a.extent[bs * 2], sy = a.extent[bs * 2 + 1];
    const float sc = fmaxf(sx, sy) / 2.0f;
    const float xn = (x - sx / 2.0f) / sc, yn = (y - sy / 2.0f) / sc;  // normalize_keypoints
    const float* wr = a.wr + lane * 4;
    const float e = wr[0] * xn + wr[1] * yn + wr[2] * scale + wr[3] * ori;
    a.cos_t[tok * kFreq + lane] = cosf(e);
    a.sin_t[tok * kFreq + lane] = sinf(e);
  }
}

// LightGlue(features = "superpoint" / any 256-d extractor): descriptors are used as they are (input_proj is the identity when
// input_dim == descriptor_dim), the positional encoding sees only the normalised (x, y).  One wave per token slot: the 256
// descriptor values go straight into the residual stream (f32 and / or hm16 rows), cos / sin(Wr [x^, y^]) into the tables.
__global__ __launch_bounds__(256) void k_prep_sp(PrepArgs a) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bs = blockIdx.y, b = bs >> 1, side = bs & 1;
  const int i = blockIdx.x * 4 + wave;
  const int n = side ? a.n_r[b] : a.n_q[b];
  const int stride = side ? a.stride_r : a.stride_q;
  const size_t tok = (size_t)bs * a.npad + i;
  f32x4v d = {0.f, 0.f, 0.f, 0.f};
  const bool valid = i < n;
  if (valid) d = *reinterpret_cast<const f32x4v*>((side ? a.desc_r : a.desc_q) + ((size_t)b * stride + i) * kDim + lane * 4);
  if (a.x != nullptr) *reinterpret_cast<f32x4v*>(a.x + tok * kDim + lane * 4) = d;
  if (a.xp != nullptr) {
    const f16x4 h = __builtin_convertvector(d, f16x4);
    const f16x4 m = __builtin_convertvector(d - __builtin_convertvector(h, f32x4v), f16x4);
    uint16_t* q = a.xp + hm16_off(tok, kDim, lane * 4);
    *reinterpret_cast<f16x4*>(q) = h;
    *reinterpret_cast<f16x4*>(q + 16) = m;
  }
  if (lane < kFreq) {
    if (!valid) { a.cos_t[tok * kFreq + lane] = 1.f; a.sin_t[tok * kFreq + lane] = 0.f; return; }
    const int fmt = a.kpt_format & 0xff;
    const int w = fmt == GN_KPT_LAF ? 6 : 4;
    const float* kp = (side ? a.kpt_r : a.kpt_q) + ((size_t)b * stride + i) * w;
    const float x = kp[fmt == GN_KPT_LAF ? 2 : 0], y = kp[fmt == GN_KPT_LAF ? 5 : 1];
    const float sx = a.extent[bs * 2], sy = a.extent[bs * 2 + 1];
    const float sc = fmaxf(sx, sy) / 2.0f;
    const float xn = (x - sx / 2.0f) / sc, yn = (y - sy / 2.0f) / sc;  // normalize_keypoints
    const float* wr = a.wr + lane * 2;
    const float e = wr[0] * xn + wr[1] * yn;
    a.cos_t[tok * kFreq + lane] = cosf(e);
    a.sin_t[tok * kFreq + lane] = sinf(e);
  }
}

// one wave per 512-wide row, in place
__global__ __launch_bounds__(256) void k_ln_gelu(float* h, const float* gamma, const float* beta, int rows, uint16_t* hp, unsigned int* ovf) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* p = h + (size_t)row * 512;
  float4 v0 = *reinterpret_cast<const float4*>(p + lane * 4);
  float4 v1 = *reinterpret_cast<const float4*>(p + 256 + lane * 4);
  const float mean = wave_sum(v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w) * (1.0f / 512.0f);
  v0.x -= mean; v0.y -= mean; v0.z -= mean; v0.w -= mean;
  v1.x -= mean; v1.y -= mean; v1.z -= mean; v1.w -= mean;
  const float var = wave_sum(v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w +
                             v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w) * (1.0f / 512.0f);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  const float4 g0 = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float4 g1 = *reinterpret_cast<const float4*>(gamma + 256 + lane * 4);
  const float4 b0 = *reinterpret_cast<const float4*>(beta + lane * 4);
  const float4 b1 = *reinterpret_cast<const float4*>(beta + 256 + lane * 4);
  auto f = [rstd](float x, float g, float b) { return gelu_erf(x * rstd * g + b); };
  v0.x = f(v0.x, g0.x, b0.x); v0.y = f(v0.y, g0.y, b0.y); v0.z = f(v0.z, g0.z, b0.z); v0.w = f(v0.w, g0.w, b0.w);
  v1.x = f(v1.x, g1.x, b1.x); v1.y = f(v1.y, g1.y, b1.y); v1.z = f(v1.z, g1.z, b1.z); v1.w = f(v1.w, g1.w, b1.w);
  if (hp != nullptr) {   // hm16 rows for the f16x2 GEMM: x = xh + xm, both round-to-nearest
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    uint16_t* q0 = hp + hm16_off((size_t)row, 512, lane * 4), *q1 = hp + hm16_off((size_t)row, 512, 256 + lane * 4);
    const f32x4v a0 = {v0.x, v0.y, v0.z, v0.w}, a1 = {v1.x, v1.y, v1.z, v1.w};
    const f16x4 h0 = __builtin_convertvector(a0, f16x4), h1 = __builtin_convertvector(a1, f16x4);
    const f16x4 m0 = __builtin_convertvector(a0 - __builtin_convertvector(h0, f32x4v), f16x4);
    const f16x4 m1 = __builtin_convertvector(a1 - __builtin_convertvector(h1, f32x4v), f16x4);
    *reinterpret_cast<f16x4*>(q0) = h0; *reinterpret_cast<f16x4*>(q1) = h1;
    *reinterpret_cast<f16x4*>(q0 + 16) = m0; *reinterpret_cast<f16x4*>(q1 + 16) = m1;
    float amax = 0.f;
    ovf_track(amax, v0.x, v0.y); ovf_track(amax, v0.z, v0.w); ovf_track(amax, v1.x, v1.y); ovf_track(amax, v1.z, v1.w);
    ovf_commit(ovf, amax);
    return;
  }
  *reinterpret_cast<float4*>(p + lane * 4) = v0;
  *reinterpret_cast<float4*>(p + 256 + lane * 4) = v1;
}

// one wave per 256-wide row
__global__ __launch_bounds__(256) void k_matchability(const float* x, const float* w, const float* b, float* ls, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * kDim + lane * 4);
  const float4 ww = *reinterpret_cast<const float4*>(w + lane * 4);
  const float z = wave_sum(v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w) + b[0];
  // logsigmoid(z) = min(z, 0) - log1p(exp(-|z|))
  if (lane == 0) ls[row] = fminf(z, 0.f) - log1pf(expf(-fabsf(z)));
}

__global__ void k_cast_bf16(const float* in, uint16_t* out, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    unsigned int u = __float_as_uint(in[i]);
    u += 0x7FFFu + ((u >> 16) & 1u);
    out[i] = (uint16_t)(u >> 16);
  }
}
// x = h + m + l exactly, each term bf16 (8 + 8 + 8 mantissa bits); used once per weight matrix at load time
__global__ void k_split3_bf16(const float* in, uint16_t* planes, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float x = in[i];
    const unsigned int u = __float_as_uint(x);
    const float r = x - __uint_as_float(u & 0xFFFF0000u);      // truncation split, same rule as split8 in gn_gemm.hip
    const unsigned int v = __float_as_uint(r);
    const float r2 = r - __uint_as_float(v & 0xFFFF0000u);
    planes[i] = (uint16_t)(u >> 16);
    planes[n + i] = (uint16_t)(v >> 16);
    planes[2 * n + i] = (uint16_t)(__float_as_uint(r2) >> 16);
  }
}
// x * scale = h + m + e with h, m fp16 (round to nearest), |e| <= 2^-22 |x * scale|; planes of the f16x2 GEMM
__global__ void k_split2_f16(const float* in, uint16_t* planes, long long n, float scale) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float x = in[i] * scale;
    const _Float16 h = (_Float16)x;
    const _Float16 m = (_Float16)(x - (float)h);
    planes[i] = __builtin_bit_cast(uint16_t, h);
    planes[n + i] = __builtin_bit_cast(uint16_t, m);
  }
}
// the same split written in the hm16 row format consumed by k_gemm_p2 (4 columns per thread)
__global__ void k_split_hm16(const float* in, uint16_t* out, long long rows, int cols, float scale) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const long long n4 = rows * cols / 4;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < n4; i += step) {
    const long long e = i * 4, row = e / cols;
    const int col = (int)(e - row * cols);
    const f32x4v x = *reinterpret_cast<const f32x4v*>(in + e) * scale;
    const f16x4 h = __builtin_convertvector(x, f16x4);
    const f16x4 m = __builtin_convertvector(x - __builtin_convertvector(h, f32x4v), f16x4);
    uint16_t* q = out + hm16_off((size_t)row, cols, col);
    *reinterpret_cast<f16x4*>(q) = h;
    *reinterpret_cast<f16x4*>(q + 16) = m;
  }
}
// rotary tables re-laid-out for k_qkv (gn_qkv.hip): rot4[fg][t] = (cos[t][2 fg], cos[t][2 fg + 1], sin[t][2 fg], sin[t][2 fg + 1]),
// fg = 0..15, rows `stride` tokens apart -- the lanes of a wave are consecutive tokens, so one 16-byte load per lane is a fully coalesced wave load
__global__ void k_rot_table(const float* cos_t, const float* sin_t, f32x4* rot4, int T, long long stride) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
#pragma unroll
  for (int fg = 0; fg < kFreq / 2; ++fg) {
    const float2 c = *reinterpret_cast<const float2*>(cos_t + (size_t)t * kFreq + 2 * fg);
    const float2 s_ = *reinterpret_cast<const float2*>(sin_t + (size_t)t * kFreq + 2 * fg);
    const f32x4 v = {c.x, c.y, s_.x, s_.y};
    rot4[(size_t)fg * stride + t] = v;
  }
}
}  // namespace

void launch_rot_table(const float* cos_t, const float* sin_t, float* rot4, int T, long long stride, hipStream_t s) {
  hipLaunchKernelGGL(k_rot_table, dim3((T + 255) / 256), dim3(256), 0, s, cos_t, sin_t, reinterpret_cast<f32x4*>(rot4), T, stride);
}

void launch_split_hm16(const float* in, uint16_t* out, long long rows, int cols, float scale, hipStream_t s) {
  hipLaunchKernelGGL(k_split_hm16, dim3(2048), dim3(256), 0, s, in, out, rows, cols, scale);
}

void launch_split2_f16(const float* in, uint16_t* planes, long long n, float scale, hipStream_t s) {
  hipLaunchKernelGGL(k_split2_f16, dim3(2048), dim3(256), 0, s, in, planes, n, scale);
}

void launch_split3_bf16(const float* in, uint16_t* planes, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_split3_bf16, dim3(512), dim3(256), 0, s, in, planes, n);
}

void launch_prep(const PrepArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_extent, dim3(a.B * 2), dim3(256), 0, s, a);
  if (a.feature == 1) hipLaunchKernelGGL(k_prep_sp, dim3(a.npad / 4, a.B * 2), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_prep, dim3(a.npad / 4, a.B * 2), dim3(256), 0, s, a);
}
void launch_ln_gelu(float* h, const float* gamma, const float* beta, int rows, hipStream_t s, uint16_t* hp, unsigned int* ovf) {
  hipLaunchKernelGGL(k_ln_gelu, dim3((rows + 3) / 4), dim3(256), 0, s, h, gamma, beta, rows, hp, ovf);
}
void launch_matchability(const float* x, const float* w, const float* b, float* ls, int rows, hipStream_t s) {
  hipLaunchKernelGGL(k_matchability, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, ls, rows);
}
void launch_cast_bf16(const float* in, uint16_t* out, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_cast_bf16, dim3(2048), dim3(256), 0, s, in, out, n);
}

namespace {
// one workgroup; any number of slots: the per-slot tile / item counts are prefix-summed 256 slots at a time (block-wide scan, running base), so
// nothing is sized by BS (the fixed 1025-entry arrays of round 4 were overrun by contexts of more than 512 pairs per call)
__global__ __launch_bounds__(256) void k_tile_lists(const int32_t* nvalid, int BS, int npad, int* lists, unsigned long long* feedback) {
  __shared__ int sc_t[256], sc_a[256], base[2];
  const int gx = npad / 256, tps = npad / 128, tid = threadIdx.x;
  const bool items = gx > 0 && npad % 256 == 0;
  int* const tl = lists + kTileListBase;
  int* const al = tl + BS * tps;
  if (tid == 0) { base[0] = 0; base[1] = 0; }
  __syncthreads();
  for (int s0 = 0; s0 < BS; s0 += 256) {
    const int s = s0 + tid;
    int nt = 0, nb = 0;
    if (s < BS) {
      const int n = min(max(nvalid[s], 0), npad);
      nt = (n + 127) / 128;
      nb = items ? (n + 255) / 256 : 0;
    }
    sc_t[tid] = nt; sc_a[tid] = 4 * nb;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int vt = tid >= d ? sc_t[tid - d] : 0, va = tid >= d ? sc_a[tid - d] : 0;
      __syncthreads();
      sc_t[tid] += vt; sc_a[tid] += va;
      __syncthreads();
    }
    const int ot = base[0] + sc_t[tid] - nt, oa = base[1] + sc_a[tid] - 4 * nb;
    for (int i = 0; i < nt; ++i) tl[ot + i] = s * tps + i;
    for (int h = 0; h < 4; ++h)
      for (int i = 0; i < nb; ++i) al[oa + h * nb + i] = (s * 4 + h) * gx + i;
    __syncthreads();
    if (tid == 255) { base[0] += sc_t[255]; base[1] += sc_a[255]; }
    __syncthreads();
  }
  if (tid == 0) {
    lists[0] = base[0]; lists[1] = base[1];
    // (valid tiles, all tiles) of this call as ONE 8-byte store into pinned host memory: the library reads it -- whenever it has arrived, no
    // synchronisation -- to choose the block tail's form for the NEXT call of the same group (a hint only: both forms give the same bits)
    if (feedback) __hip_atomic_store(feedback, ((unsigned long long)(unsigned)(BS * tps) << 32) | (unsigned)base[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace
void launch_tile_lists(const int32_t* nvalid, int BS, int npad, int* lists, unsigned long long* feedback, hipStream_t s) {
  hipLaunchKernelGGL(k_tile_lists, dim3(1), dim3(256), 0, s, nvalid, BS, npad, lists, feedback);
}

}  // namespace gn
