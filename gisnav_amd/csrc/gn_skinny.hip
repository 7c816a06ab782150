// Small-grid GEMMs of the f16x2 precision mode (round 5): ONE or TWO pairs per call -- the reference's own operating point
// (one pair per message at <= 5 Hz: ros/gisnav/gisnav/core/pose_node.py:178-184, docker/gscam/gscam_params.yaml:8).
//
// At 2048 tokens the bulk kernels' shapes (128-token workgroups that stream a whole weight matrix each) put 16 workgroups on 256 CUs, and
// the small-grid fall-backs of rounds 2-4 (k_gemm_p2's 128 x 128 LDS tiles for the projections: 64-96 workgroups, 12 us; k_ffn_fused's
// 32-token workgroups for the block tail: 64 workgroups x 1.8 MB of weights each, 22 us) are bound by what ONE CU can pull out of L2
// (~45 B/clk).  The lever named there -- split the WEIGHT stream across CUs -- is this file: a workgroup owns 32 (16) tokens x
// 128 features (4 waves, one 32 x 32 MFMA tile each over the whole K), i.e. 256-384 workgroups at one pair, each streaming 128-256 KB of
// weights.  The token rows of a workgroup are staged ONCE in LDS, verbatim (hm16 rows are already in k-step order: 16-byte chunk
// 4 ks + 2 term + half-wave), with a 16-byte row pad that spreads the 32 rows of a fragment read over all banks; the weight fragments
// (build_weight_fragments order, natural k: 1 KB per instruction and wave) go straight from L2 into a register ring of RD k-steps.
// (First form of this file, measured and dropped: token fragments read straight from L2 as well -- 32 cache lines per instruction and the
// same 64 KB of rows fetched by all four waves through a 32 KB L1: 17 us per tail GEMM against 22 us for the whole k_ffn_fused tail.)
//
//   k_skinny_qkv   : attention input projections (kornia SelfBlock Wqkv + rotary / CrossBlock to_qk, to_v) -> fp16 / bf16 q | k rows and
//                    V^T panels: k_qkv's arithmetic (gn_qkv.hip: the same partial products in the same order per accumulator, the same
//                    epilogue expressions) -> the same bits as the bulk kernel on the same rows;
//   k_skinny_h     : block tail, first GEMM on the composed weight: h = [W1_x | W1_m Wo] [x | ctx] + (b1 + W1_m bo)   (f32 rows);
//   k_skinny_out   : block tail, LayerNorm + GELU of the workgroup's 16 token rows (k_ln_gelu's arithmetic, row for row; computed by both
//                    workgroups that share the rows) -> hm16 tile in LDS -> second GEMM + bias + residual: x += ffn.3(g)   (hm16 rows in
//                    place, optional f32 copy).
// Two launches instead of k_ffn_fused's one: the row statistics of LayerNorm need all 512 hidden features of a token, which no longer
// live in one workgroup (an in-launch hand-off between workgroups costs what a launch boundary costs at this size).
#include "gn_common.h"
#include <type_traits>

namespace gn {

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kPad = 16;      // bytes behind every staged row: consecutive rows start 4 banks apart

// one 32 x 32 tile over K: acc += sum_ks products(ks).  Product order per k-step: W_m X_h, (W_h X_m,) W_h X_h -- small terms first, as in
// k_qkv / k_ffn128.  TOKROWS: tokens are the A operand (rows = tokens, columns = features) instead of the weights.
// wsrc: this lane's 16 bytes of the tile's first weight fragment; xs: this lane's token row in the LDS tile (+ 16 bytes for the upper half-wave);
// pre(): requests the workgroup's token rows (loads return in order: they must be in front of the weight ring's); stage(): fills the LDS tile
// from them and ends with the workgroup barrier -- called once the first RD k-steps of weights have been requested.  RD = K / 16: the whole
// weight slice of the wave is in flight from the first instruction on (every launch's weights come from the Infinity Cache -- a call's 50 MB of
// weights cycle through 4 MB of L2 per XCD -- so a short ring pays one miss latency per RD k-steps: k_skinny_h 12.7 -> 8 us with RD 12 -> 32).
template <int K, int NP, bool TOKROWS, int RD, typename Pre, typename Stage>
__device__ __forceinline__ f32x16 skinny_tile(const unsigned char* __restrict__ wsrc, const unsigned char* xs, Pre&& pre, Stage&& stage) {
  constexpr int KS = K / 16;
  constexpr int PRE = RD < 31 ? RD : 31;     // k-steps requested in front of stage() (s_waitcnt counts to 63)
  f16x8 fw[RD][2];
  pre();
#pragma unroll
  for (int q = 0; q < PRE; ++q)
#pragma unroll
    for (int term = 0; term < 2; ++term) fw[q][term] = *reinterpret_cast<const f16x8*>(wsrc + (size_t)(q * 2 + term) * 1024);
  stage(std::integral_constant<int, 2 * PRE>{});
#pragma unroll
  for (int q = PRE; q < RD; ++q)
#pragma unroll
    for (int term = 0; term < 2; ++term) fw[q][term] = *reinterpret_cast<const f16x8*>(wsrc + (size_t)(q * 2 + term) * 1024);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int s = ks % RD;
    const f16x8 xh = *reinterpret_cast<const f16x8*>(xs + ks * 64);
    f16x8 xm = xh;
    if (NP == 3) xm = *reinterpret_cast<const f16x8*>(xs + ks * 64 + 32);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      if (NP == 2 && p == 1) continue;
      const f16x8 w = fw[s][p == 0 ? 1 : 0], x = p == 1 ? xm : xh;
      acc = TOKROWS ? __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, acc, 0, 0, 0);
    }
    if (ks + RD < KS) {
#pragma unroll
      for (int term = 0; term < 2; ++term) fw[s][term] = *reinterpret_cast<const f16x8*>(wsrc + (size_t)((ks + RD) * 2 + term) * 1024);
    }
  }
  return acc;
}

// token rows -> LDS without passing through registers (the weight slice needs the 256 architectural ones): LDS-DMA, one 1 KB row per instruction
// and wave, LDS row stride STRIDE, byte column col0.  Written in assembly (the compiler treats the builtin as a FLAT access and then waits
// vmcnt(0) -- for the whole weight ring -- in front of the first fragment read; as in gn_attention_pw.hip); the rows are the OLDEST loads of
// the wave, so the compiler's own counted waits for the weight registers stay conservative, and stage_wait<N>() = "everything but the N loads
// issued since" is the explicit wait for them.
typedef __attribute__((address_space(3))) void* lptr_t;
template <int PER, int STRIDE>
__device__ __forceinline__ void rows_dma(const unsigned char* tile, int col0, const unsigned char* __restrict__ src, int wave, int lane) {
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)tile;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const unsigned char* sbase = src + (size_t)(wave * PER + i) * 1024;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((unsigned)(lane * 16)), "s"(sbase), "s"(lds0 + (unsigned)((wave * PER + i) * STRIDE + col0)) : "memory");
  }
}
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
template <int PER>
__device__ __forceinline__ void rows_load(u32x4s (&v)[PER], const unsigned char* __restrict__ src, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < PER; ++i) v[i] = *reinterpret_cast<const u32x4s*>(src + (size_t)(wave * PER + i) * 1024 + lane * 16);
}
template <int PER, int STRIDE>
__device__ __forceinline__ void rows_store(const u32x4s (&v)[PER], unsigned char* tile, int col0, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < PER; ++i) *reinterpret_cast<u32x4s*>(tile + (wave * PER + i) * STRIDE + col0 + lane * 16) = v[i];
}
template <int N> __device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- attention input projections ---------------------------------------------------------------------------------------------------
template <bool CROSS, bool F16, int NP>
__global__ __launch_bounds__(256) void k_skinny_qkv(QkvArgs a) {
  constexpr int NQK = CROSS ? kDim : 2 * kDim;
  constexpr int STRIDE = 1024 + kPad;
  __shared__ __attribute__((aligned(16))) unsigned char tile[32 * STRIDE];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int bm = blockIdx.x * 32;
  const int tile_f = blockIdx.y * 4 + wave;               // 32-feature tile of [q | k | v] (or [qk | v])
  const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(a.wf) + (size_t)tile_f * 16 * 2 * 1024 + lane * 16;
  const unsigned char* const xs = tile + ql * STRIDE + hh * 16;
  auto pre = [&]() __attribute__((always_inline)) { rows_dma<8, STRIDE>(tile, 0, reinterpret_cast<const unsigned char*>(a.xp) + (size_t)bm * 1024, wave, lane); };
  auto stage = [&](auto n) __attribute__((always_inline)) {
    stage_wait<decltype(n)::value>();
    __syncthreads();
  };
  const float ascale = a.acc_scale;
  float amax = 0.f;
  if (32 * tile_f < NQK) {
    // register r <-> feature 32 tile + (r & 3) + 8 (r >> 2) + 4 hh, token bm + ql (k_qkv's q / k epilogue, expression for expression)
    const f32x16 acc = skinny_tile<256, NP, false, 16>(wsrc, xs, pre, stage);
    uint2 pk[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + 32 * tile_f + 8 * g + 4 * hh);
      f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      v.x = v.x * ascale; v.y = v.y * ascale; v.z = v.z * ascale; v.w = v.w * ascale;
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      if (CROSS) {
        v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
      } else {
        const int fg = ((32 * tile_f + 8 * g + 4 * hh) & 63) >> 2;
        const f32x4 rot = *reinterpret_cast<const f32x4*>(a.rot4 + ((size_t)fg * a.rot_stride + (size_t)(bm + ql)) * 4);
        f32x4 o;
        o.x = v.x * rot.x + (-v.y) * rot.z;
        o.y = v.y * rot.x + v.x * rot.z;
        o.z = v.z * rot.y + (-v.w) * rot.w;
        o.w = v.w * rot.y + v.z * rot.w;
        v = o;
        if (tile_f < 8) { v.x *= a.qscale; v.y *= a.qscale; v.z *= a.qscale; v.w *= a.qscale; }
      }
      pk[g].x = pack16<F16>(v.x, v.y);
      pk[g].y = pack16<F16>(v.z, v.w);
      if (F16) { ovf_track(amax, v.x, v.y); ovf_track(amax, v.z, v.w); }
    }
    const size_t row = (size_t)(bm + ql);
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {     // the half-waves trade every other group of 4 features: a lane stores 8 consecutive features (16 bytes)
      const uint2 give = hh ? pk[2 * gp] : pk[2 * gp + 1];
      uint2 got;
      got.x = __shfl_xor(give.x, 32); got.y = __shfl_xor(give.y, 32);
      const uint2 own = hh ? pk[2 * gp + 1] : pk[2 * gp];
      const uint4 out = hh ? make_uint4(got.x, got.y, own.x, own.y) : make_uint4(own.x, own.y, got.x, got.y);
      *reinterpret_cast<uint4*>(a.qkb + row * a.ldyb + 32 * tile_f + 16 * gp + 8 * hh) = out;
    }
  } else {
    // register r <-> token bm + (r & 3) + 8 (r >> 2) + 4 hh, feature d = 32 (tile - NQK / 32) + ql of the V panel; registers 8 m .. 8 m + 7 are the
    // keys 16 m + 4 hh + {0..3, 8..11}: group 2 m + hh of the permuted V^T layout
    const f32x16 acc = skinny_tile<256, NP, true, 16>(wsrc, xs, pre, stage);
    const int slot = bm / a.npad, i0 = bm - slot * a.npad;
    const int d = 32 * tile_f - NQK + ql;
    const float bias = a.bias[NQK + d];
    uint16_t* const dst = a.vt + (((size_t)slot * kHeads + (d >> 6)) * kHeadDim + (d & 63)) * a.npad + i0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      unsigned int w4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = acc[8 * m + 2 * e] * ascale + bias;
        const float hi = acc[8 * m + 2 * e + 1] * ascale + bias;
        w4[e] = pack16<F16>(lo, hi);
        if (F16) ovf_track(amax, lo, hi);
      }
      *reinterpret_cast<uint4*>(dst + 8 * (2 * m + hh)) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
  }
  if (F16) ovf_commit(a.ovf, amax);
}

// ---- block tail, first GEMM (composed ffn.0 over [x | ctx]) -> f32 hidden rows -------------------------------------------------------
template <int RD, bool DMA>
__global__ __launch_bounds__(256) void k_skinny_h(SkinnyTailArgs a) {
  constexpr int STRIDE = 2048 + kPad;
  __shared__ __attribute__((aligned(16))) unsigned char tile[32 * STRIDE];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int bm = blockIdx.x * 32;
  const int tile_f = blockIdx.y * 4 + wave;               // of 16
  const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(a.w1) + (size_t)tile_f * 32 * 2 * 1024 + lane * 16;
  u32x4s rx[8], rc[8];
  auto pre = [&]() __attribute__((always_inline)) {
    if (DMA) {
      rows_dma<8, STRIDE>(tile, 0, reinterpret_cast<const unsigned char*>(a.xp) + (size_t)bm * 1024, wave, lane);
      rows_dma<8, STRIDE>(tile, 1024, reinterpret_cast<const unsigned char*>(a.cp) + (size_t)bm * 1024, wave, lane);
    } else {
      rows_load<8>(rx, reinterpret_cast<const unsigned char*>(a.xp) + (size_t)bm * 1024, wave, lane);
      rows_load<8>(rc, reinterpret_cast<const unsigned char*>(a.cp) + (size_t)bm * 1024, wave, lane);
    }
  };
  auto stage = [&](auto n) __attribute__((always_inline)) {
    if (DMA) stage_wait<decltype(n)::value>();
    else { rows_store<8, STRIDE>(rx, tile, 0, wave, lane); rows_store<8, STRIDE>(rc, tile, 1024, wave, lane); }
    __syncthreads();
  };
  const f32x16 acc = skinny_tile<512, 3, false, RD>(wsrc, tile + ql * STRIDE + hh * 16, pre, stage);
  float* const hrow = a.h + (size_t)(bm + ql) * (2 * kDim) + 32 * tile_f + 4 * hh;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b1 + 32 * tile_f + 8 * g + 4 * hh);
    f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    v.x = v.x * a.w1_scale + b4.x; v.y = v.y * a.w1_scale + b4.y; v.z = v.z * a.w1_scale + b4.z; v.w = v.w * a.w1_scale + b4.w;
    *reinterpret_cast<f32x4*>(hrow + 8 * g) = v;
  }
}

// ---- block tail: LayerNorm + GELU of 16 rows -> LDS, second GEMM + bias + residual -> hm16 rows (in place) -----------------------------
__device__ inline float skinny_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int RD>
__global__ __launch_bounds__(256) void k_skinny_out(SkinnyTailArgs a) {
  constexpr int STRIDE = 2048 + kPad;
  __shared__ __attribute__((aligned(16))) unsigned char tile[16 * STRIDE];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int bm = blockIdx.x * 16;
  const int tile_f = blockIdx.y * 4 + wave;               // of 8
  const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(a.w2) + (size_t)tile_f * 32 * 2 * 1024 + lane * 16;
  float amax = 0.f;
  // k_ln_gelu (gn_prep.hip), row for row: a wave owns rows wave, wave + 4, ..; a lane the columns 4 lane .. + 3 and 256 + 4 lane .. + 3
  f32x4 v0[4], v1[4];
  f16x4 rxh[4], rxm[4];     // the residual rows of the epilogue: requested with the hidden rows, in front of the weights
  auto pre = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* p = a.h + (size_t)(bm + wave + 4 * i) * 512;
      v0[i] = *reinterpret_cast<const f32x4*>(p + lane * 4);
      v1[i] = *reinterpret_cast<const f32x4*>(p + 256 + lane * 4);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint16_t* const xp = a.xp + hm16_off((size_t)(bm + (ql & 15)), kDim, 32 * tile_f + 8 * g + 4 * hh);
      rxh[g] = *reinterpret_cast<const f16x4*>(xp);
      rxm[g] = *reinterpret_cast<const f16x4*>(xp + 16);
    }
  };
  auto stage = [&](auto) __attribute__((always_inline)) {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln_g + lane * 4), g1 = *reinterpret_cast<const f32x4*>(a.ln_g + 256 + lane * 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.ln_b + lane * 4), b1 = *reinterpret_cast<const f32x4*>(a.ln_b + 256 + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 x0 = v0[i], x1 = v1[i];
      const float mean = skinny_wave_sum(x0.x + x0.y + x0.z + x0.w + x1.x + x1.y + x1.z + x1.w) * (1.0f / 512.0f);
      x0.x -= mean; x0.y -= mean; x0.z -= mean; x0.w -= mean;
      x1.x -= mean; x1.y -= mean; x1.z -= mean; x1.w -= mean;
      const float var = skinny_wave_sum(x0.x * x0.x + x0.y * x0.y + x0.z * x0.z + x0.w * x0.w +
                                        x1.x * x1.x + x1.y * x1.y + x1.z * x1.z + x1.w * x1.w) * (1.0f / 512.0f);
      const float rstd = 1.0f / sqrtf(var + 1e-5f);
      f32x4 y0, y1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y0[e] = gelu_erf(x0[e] * rstd * g0[e] + b0[e]);
        y1[e] = gelu_erf(x1[e] * rstd * g1[e] + b1[e]);
      }
      ovf_track(amax, y0.x, y0.y); ovf_track(amax, y0.z, y0.w); ovf_track(amax, y1.x, y1.y); ovf_track(amax, y1.z, y1.w);
      const f16x4 h0 = __builtin_convertvector(y0, f16x4), h1 = __builtin_convertvector(y1, f16x4);
      const f16x4 m0 = __builtin_convertvector(y0 - __builtin_convertvector(h0, f32x4), f16x4);
      const f16x4 m1 = __builtin_convertvector(y1 - __builtin_convertvector(h1, f32x4), f16x4);
      unsigned char* const q = tile + (wave + 4 * i) * STRIDE + (lane >> 2) * 64 + (lane & 3) * 8;     // hm16_off(., 512, 4 lane) in bytes
      *reinterpret_cast<f16x4*>(q) = h0; *reinterpret_cast<f16x4*>(q + 32) = m0;
      *reinterpret_cast<f16x4*>(q + 1024) = h1; *reinterpret_cast<f16x4*>(q + 1024 + 32) = m1;
    }
    __syncthreads();
  };
  // the 32 x 32 tile's token columns 16 .. 31 repeat the rows 0 .. 15 (their results are dropped): 16 tokens per workgroup keep the
  // LayerNorm / GELU work per CU at half of a 32-token tile's
  const f32x16 acc = skinny_tile<512, 3, false, RD>(wsrc, tile + (ql & 15) * STRIDE + hh * 16, pre, stage);
  if (ql < 16) {
    const size_t row = (size_t)(bm + ql);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = 32 * tile_f + 8 * g + 4 * hh;            // 4 consecutive features of this token: half a 16-byte chunk of the hm16 row
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b2 + f);
      uint16_t* const pp = a.xp_out + hm16_off(row, kDim, f);
      const f16x4 xh = rxh[g], xm = rxm[g];
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = acc[4 * g + e] * a.w2_scale + b4[e];
        v[e] = (float)xh[e] + ((float)xm[e] + y);            // + x = + x_m + x_h, the small term first (k_ffn128's epilogue)
      }
      ovf_track(amax, v.x, v.y); ovf_track(amax, v.z, v.w);
      const f16x4 hv = __builtin_convertvector(v, f16x4);
      const f16x4 mv = __builtin_convertvector(v - __builtin_convertvector(hv, f32x4), f16x4);
      *reinterpret_cast<f16x4*>(pp) = hv;
      *reinterpret_cast<f16x4*>(pp + 16) = mv;
      if (a.y != nullptr) *reinterpret_cast<f32x4*>(a.y + row * kDim + f) = v;
    }
  }
  ovf_commit(a.ovf, amax);
}
}  // namespace

void launch_skinny_qkv(const QkvArgs& a, bool cross, hipStream_t s) {
  const dim3 grid(a.T / 32, cross ? 4 : 6), block(256);
  const int sel = (a.half_fmt ? 4 : 0) | (cross ? 2 : 0) | (a.products == 2 ? 1 : 0);
  switch (sel) {
    case 7: hipLaunchKernelGGL((k_skinny_qkv<true, true, 2>), grid, block, 0, s, a); g_last_kernel = "k_skinny_qkv<true, true, 2>"; break;
    case 6: hipLaunchKernelGGL((k_skinny_qkv<true, true, 3>), grid, block, 0, s, a); g_last_kernel = "k_skinny_qkv<true, true, 3>"; break;
    case 5: hipLaunchKernelGGL((k_skinny_qkv<false, true, 2>), grid, block, 0, s, a); g_last_kernel = "k_skinny_qkv<false, true, 2>"; break;
    case 4: hipLaunchKernelGGL((k_skinny_qkv<false, true, 3>), grid, block, 0, s, a); g_last_kernel = "k_skinny_qkv<false, true, 3>"; break;
    case 3: hipLaunchKernelGGL((k_skinny_qkv<true, false, 2>), grid, block, 0, s, a); g_last_kernel = "k_skinny_qkv<true, false, 2>"; break;
    case 2: hipLaunchKernelGGL((k_skinny_qkv<true, false, 3>), grid, block, 0, s, a); g_last_kernel = "k_skinny_qkv<true, false, 3>"; break;
    case 1: hipLaunchKernelGGL((k_skinny_qkv<false, false, 2>), grid, block, 0, s, a); g_last_kernel = "k_skinny_qkv<false, false, 2>"; break;
    default: hipLaunchKernelGGL((k_skinny_qkv<false, false, 3>), grid, block, 0, s, a); g_last_kernel = "k_skinny_qkv<false, false, 3>"; break;
  }
}

// variant (knob 33 >> 4; 0 = shipped): bit 0 = token rows through registers instead of LDS-DMA, bits 1-2 = weight ring depth 32 / 28 / 24 / 12 (h) and 32 / 24 / 16 / 12 (out); measured on one box, batch 1: 0.941 / 0.966 / 0.976 / 1.043 ms per call
void launch_skinny_h(const SkinnyTailArgs& a, hipStream_t s, int variant) {
  const dim3 grid(a.T / 32, 4), block(256);
  switch (variant & 7) {
    case 1: hipLaunchKernelGGL((k_skinny_h<28, false>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_skinny_h<28, true>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_skinny_h<32, false>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((k_skinny_h<24, true>), grid, block, 0, s, a); break;
    case 5: hipLaunchKernelGGL((k_skinny_h<24, false>), grid, block, 0, s, a); break;
    case 6: hipLaunchKernelGGL((k_skinny_h<12, true>), grid, block, 0, s, a); break;
    case 7: hipLaunchKernelGGL((k_skinny_h<12, false>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((k_skinny_h<32, true>), grid, block, 0, s, a); break;
  }
  g_last_kernel = "k_skinny_h";
}

void launch_skinny_out(const SkinnyTailArgs& a, hipStream_t s, int variant) {
  const dim3 grid(a.T / 16, 2), block(256);
  switch ((variant >> 1) & 3) {
    case 1: hipLaunchKernelGGL((k_skinny_out<24>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_skinny_out<16>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_skinny_out<12>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((k_skinny_out<32>), grid, block, 0, s, a); break;
  }
  g_last_kernel = "k_skinny_out";
}

}  // namespace gn
