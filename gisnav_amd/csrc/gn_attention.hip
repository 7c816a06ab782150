// Fused softmax(Q K^T) V for LightGlue's self- and cross-attention blocks (4 heads x 64) on gfx950.
//
// Stands in for kornia's `Attention.forward` (F.scaled_dot_product_attention / einsum-softmax-einsum)
// as used by SelfBlock and CrossBlock, reached from ros/gisnav/gisnav/core/pose_node.py:285-287.
// Cross attention is issued as two launches (q=qk0,k=qk1,v=v1 and q=qk1,k=qk0,v=v0) exactly like
// kornia's flash path; the second softmax is over the columns of the same sim matrix because
// (qk1 qk0^T) = (qk0 qk1^T)^T term by term.
//
// Wave-level plan (64-wide wavefront, no cross-lane traffic inside the tile loop):
//   * a wave owns 32 query rows; the block (4 waves) shares 64-key K/V tiles staged in LDS;
//   * S^T = K Q^T is computed with the operands swapped (A = K rows, B = Q rows), so in the MFMA
//     C layout (col = lane & 31) every lane holds the scores of ONE query for 32 of the tile's 64 keys:
//     the online-softmax max / sum are in-lane reductions plus a single lane<->lane+32 exchange;
//   * O^T = V^T P^T reuses those score registers directly as the B operand (no shuffle, no LDS
//     round trip for P), and every O register of a lane belongs to that same query, so the running
//     rescale by exp(m_old - m_new) is lane-local as well.
// f32 variant: v_mfma_f32_32x32x2_f32 (exact f32).  bf16 variant: v_mfma_f32_32x32x16_bf16 with f32
// scores, softmax statistics and accumulators (GN_PREC_BF16_ATTN).
#include <type_traits>

#include "gn_common.h"

namespace gn {

namespace {
constexpr int KT = 64;        // keys per LDS tile
constexpr int KLS = 68;       // K tile row stride (floats): conflict-free ds_read_b128 across 16 rows
constexpr int QB = 128;       // query rows per block

__global__ __launch_bounds__(256) void k_attn_f32(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[KT * KLS + KT * 64];
  float* Ks = smem;
  float* Vs = smem + KT * KLS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, bs = blockIdx.z;
  const int kvs = a.cross ? (bs ^ 1) : bs;
  const int nkv = a.nvalid[kvs];
  const int q0 = blockIdx.x * QB + wave * 32;

  // Q fragment: lane (query ql, half hh) keeps Q[q][8j + 4hh + s], j = 0..7, s = 0..3.
  float qf[8][4];
  {
    const float* qp = a.q + ((size_t)bs * a.npad + q0 + ql) * a.ldq + h * 64 + 4 * hh;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(qp + 8 * j);
      qf[j][0] = t.x * a.qscale; qf[j][1] = t.y * a.qscale;
      qf[j][2] = t.z * a.qscale; qf[j][3] = t.w * a.qscale;
    }
  }

  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const float* kbase = a.k + (size_t)kvs * a.npad * a.ldk + h * 64;
  const float* vbase = a.v + (size_t)kvs * a.npad * a.ldv + h * 64;
  const int ntiles = (nkv + KT - 1) / KT;
  const int lr = tid >> 4, lc = (tid & 15) * 4;

  for (int t = 0; t < ntiles; ++t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = lr + 16 * p;
      const size_t grow = (size_t)(t * KT + r);
      *reinterpret_cast<float4*>(&Ks[r * KLS + lc]) = *reinterpret_cast<const float4*>(kbase + grow * a.ldk + lc);
      *reinterpret_cast<float4*>(&Vs[r * 64 + lc]) = *reinterpret_cast<const float4*>(vbase + grow * a.ldv + lc);
    }
    __syncthreads();

    // S^T[key][query] for the tile's two 32-key halves
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 kf = *reinterpret_cast<const float4*>(&Ks[(kt * 32 + ql) * KLS + 8 * j + 4 * hh]);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[j][0], st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[j][1], st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[j][2], st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[j][3], st[kt], 0, 0, 0);
      }
    }
    if (t * KT + KT > nkv) {  // ragged tail: keys >= nkv never contribute
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KT + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= nkv) st[kt][r] = -INFINITY;
        }
    }
    float mloc = st[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[kt][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = expf(m_run - m_new);
    l_run *= alpha;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = expf(st[kt][r] - m_new);
        st[kt][r] = p;
        l_run += p;
      }
    m_run = m_new;

    // O^T[d][query] += V^T[d][key] P^T[key][query]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float v0 = Vs[key * 64 + ql];
        const float v1 = Vs[key * 64 + 32 + ql];
        o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, st[kt][r], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, st[kt][r], o[1], 0, 0, 0);
      }
    __syncthreads();
  }

  const float l = l_run + __shfl_xor(l_run, 32);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  float* op = a.out + ((size_t)bs * a.npad + q0 + ql) * a.ldo + h * 64 + 4 * hh;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 w;
      w.x = o[d][4 * g + 0] * inv; w.y = o[d][4 * g + 1] * inv;
      w.z = o[d][4 * g + 2] * inv; w.w = o[d][4 * g + 3] * inv;
      *reinterpret_cast<float4*>(op + d * 32 + 8 * g) = w;
    }
}

// k_attn_f32_ks (round 5): the exact-f32 attention of ONE or TWO pairs -- BASELINE configs[1] as SURVEY.md reads it, the reference's own operating
// point in the guaranteed arithmetic.  k_attn_f32 puts 64 workgroups on 256 CUs there, each walking 16 key tiles whose loads nothing hides
// (101 us per launch, 47 % of the call).  Here a workgroup owns 32 queries and its four waves split the KEYS (wave w: tiles w, w + 4, ..): 256
// workgroups at one pair; a wave stages its tile in its OWN LDS region (no workgroup barrier inside the loop) and requests the next tile into
// registers before it computes on the current one; the four partial results meet through LDS by the log-sum-exp identity.  The same MFMA
// instruction and per-tile arithmetic as k_attn_f32; the order in which a query's keys enter its sums differs (rounding level).
template <int NW>     // waves per workgroup = key shares (4, or 8: two waves per SIMD, one's softmax under the other's MFMAs)
__global__ __launch_bounds__(64 * NW) void k_attn_f32_ks(AttnArgs a) {
  constexpr int KT2 = 32;                                  // keys per tile here: the next tile waits in 64 registers (with 64-key tiles the kernel spilled)
  constexpr int WREG = KT2 * KLS + KT2 * 64;               // floats of one wave's K | V region
  __shared__ __attribute__((aligned(16))) float smem[NW * 32 * 64 + 2 * NW * 64 > NW * WREG ? NW * 32 * 64 + 2 * NW * 64 : NW * WREG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, bs = blockIdx.z;
  const int kvs = a.cross ? (bs ^ 1) : bs;
  const int nkv = a.nvalid[kvs];
  const int q0 = blockIdx.x * 32;
  float* Ks = smem + wave * WREG;
  float* Vs = Ks + KT2 * KLS;

  float qf[8][4];
  {
    const float* qp = a.q + ((size_t)bs * a.npad + q0 + ql) * a.ldq + h * 64 + 4 * hh;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(qp + 8 * j);
      qf[j][0] = t.x * a.qscale; qf[j][1] = t.y * a.qscale;
      qf[j][2] = t.z * a.qscale; qf[j][3] = t.w * a.qscale;
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const float* kbase = a.k + (size_t)kvs * a.npad * a.ldk + h * 64;
  const float* vbase = a.v + (size_t)kvs * a.npad * a.ldv + h * 64;
  const int ntiles = (nkv + KT2 - 1) / KT2;
  // the wave's share of a tile: lane -> (row lr + 4 p, 16-byte column lc), p = 0 .. 7
  const int lr = lane >> 4, lc = (lane & 15) * 4;
  float4 pk[8], pv[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) { pk[p] = make_float4(0.f, 0.f, 0.f, 0.f); pv[p] = pk[p]; }     // (defined on every path: otherwise the two arrays live in scratch memory)
  auto fetch = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const size_t grow = (size_t)(t * KT2 + lr + 4 * p);
      pk[p] = *reinterpret_cast<const float4*>(kbase + grow * a.ldk + lc);
      pv[p] = *reinterpret_cast<const float4*>(vbase + grow * a.ldv + lc);
    }
  };
  if (wave < ntiles) fetch(wave);
  for (int t = wave; t < ntiles; t += NW) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      *reinterpret_cast<float4*>(&Ks[(lr + 4 * p) * KLS + lc]) = pk[p];
      *reinterpret_cast<float4*>(&Vs[(lr + 4 * p) * 64 + lc]) = pv[p];
    }
    if (t + NW < ntiles) fetch(t + NW);
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 kf = *reinterpret_cast<const float4*>(&Ks[ql * KLS + 8 * j + 4 * hh]);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[j][0], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[j][1], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[j][2], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[j][3], st, 0, 0, 0);
    }
    if (t * KT2 + KT2 > nkv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * KT2 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= nkv) st[r] = -INFINITY;
      }
    }
    float mloc = st[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = expf(m_run - m_new);
    l_run *= alpha;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = expf(st[r] - m_new);
      st[r] = p;
      l_run += p;
    }
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float v0 = Vs[key * 64 + ql];
      const float v1 = Vs[key * 64 + 32 + ql];
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, st[r], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, st[r], o[1], 0, 0, 0);
    }
  }
  // merge: every wave publishes (m, l, O) of its key share; wave w then finishes 32 / NW of the 32 output registers (d2 = w / (NW / 2), groups of 4)
  const float l_wave = l_run + __shfl_xor(l_run, 32);
  __syncthreads();                                      // every wave is done with its K | V region
  float* Om = smem;                                     // [NW waves][32 registers][64 lanes]
  float* Mm = smem + NW * 32 * 64;                      // [NW][64]
  float* Lm = Mm + NW * 64;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) Om[(wave * 32 + d * 16 + r) * 64 + lane] = o[d][r];
  Mm[wave * 64 + lane] = m_run;
  Lm[wave * 64 + lane] = l_wave;
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int w = 0; w < NW; ++w) M = fmaxf(M, Mm[w * 64 + lane]);
  float sc[NW], L = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float mw = Mm[w * 64 + lane];
    sc[w] = mw == -INFINITY ? 0.f : expf(mw - M);
    L += Lm[w * 64 + lane] * sc[w];
  }
  const float inv = L > 0.f ? 1.0f / L : 0.f;
  constexpr int G = 8 / NW;                             // groups of 4 registers per wave: 2 (NW = 4) or 1 (8)
  const int d2 = wave / (NW / 2), g0 = (wave % (NW / 2)) * G;
  float* op = a.out + ((size_t)bs * a.npad + q0 + ql) * a.ldo + h * 64 + 4 * hh + d2 * 32;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += Om[(w * 32 + d2 * 16 + 4 * (g0 + g) + e) * 64 + lane] * sc[w];
      acc[e] = v * inv;
    }
    *reinterpret_cast<float4*>(op + 8 * (g0 + g)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 variant.  K tile is staged as bf16 [64 keys][64 d] (row stride 72 halves = 144 B), V tile is
// staged TRANSPOSED as bf16 [64 d][64 keys] (row stride 72) so that the A operand of O^T = V^T P^T
// (8 consecutive keys for one d) is contiguous.  Each lane's 16 f32 scores per 32-key half are
// packed to bf16 in the order the MFMA wants: step u of a half uses registers 8u..8u+7, i.e. keys
// {8u*2.. } as laid out by the C layout (keys (r&3) + 8(r>>2) + 4hh).  Any consistent permutation of
// k between A and B is legal, so V^T is read at exactly those key positions.
constexpr int HLS = 72;  // halves per LDS row

typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kRing = 64 * 64;         // k_attn_bf16_v5: shorts per ring stage (one 64 x 64 bf16 tile)
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ inline unsigned short f2bf(float x) {
  // round-to-nearest-even f32 -> bf16 (inputs are finite here)
  unsigned int u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(256) void k_attn_bf16(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * KT * HLS];
  unsigned short* Ks = smem;           // [key][d]
  unsigned short* Vt = smem + KT * HLS;  // [d][key]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, bs = blockIdx.z;
  const int kvs = a.cross ? (bs ^ 1) : bs;
  const int nkv = a.nvalid[kvs];
  const int q0 = blockIdx.x * QB + wave * 32;

  // Q fragment as MFMA B operand: lane (query ql, half hh) holds Q[q][16c + 8hh + 0..7], c = 0..3.
  bf16x8 qf[4];
  {
    const float* qp = a.q + ((size_t)bs * a.npad + q0 + ql) * a.ldq + h * 64 + 8 * hh;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 t0 = *reinterpret_cast<const float4*>(qp + 16 * c);
      const float4 t1 = *reinterpret_cast<const float4*>(qp + 16 * c + 4);
      qf[c][0] = (short)f2bf(t0.x * a.qscale); qf[c][1] = (short)f2bf(t0.y * a.qscale);
      qf[c][2] = (short)f2bf(t0.z * a.qscale); qf[c][3] = (short)f2bf(t0.w * a.qscale);
      qf[c][4] = (short)f2bf(t1.x * a.qscale); qf[c][5] = (short)f2bf(t1.y * a.qscale);
      qf[c][6] = (short)f2bf(t1.z * a.qscale); qf[c][7] = (short)f2bf(t1.w * a.qscale);
    }
  }

  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const float* kbase = a.k + (size_t)kvs * a.npad * a.ldk + h * 64;
  const float* vbase = a.v + (size_t)kvs * a.npad * a.ldv + h * 64;
  const int ntiles = (nkv + KT - 1) / KT;
  const int lr = tid >> 4, lc = (tid & 15) * 4;

  for (int t = 0; t < ntiles; ++t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = lr + 16 * p;
      const size_t grow = (size_t)(t * KT + r);
      const float4 kv = *reinterpret_cast<const float4*>(kbase + grow * a.ldk + lc);
      const float4 vv = *reinterpret_cast<const float4*>(vbase + grow * a.ldv + lc);
      ushort4 kb;
      kb.x = f2bf(kv.x); kb.y = f2bf(kv.y); kb.z = f2bf(kv.z); kb.w = f2bf(kv.w);
      *reinterpret_cast<ushort4*>(&Ks[r * HLS + lc]) = kb;
      Vt[(lc + 0) * HLS + r] = f2bf(vv.x);
      Vt[(lc + 1) * HLS + r] = f2bf(vv.y);
      Vt[(lc + 2) * HLS + r] = f2bf(vv.z);
      Vt[(lc + 3) * HLS + r] = f2bf(vv.w);
    }
    __syncthreads();

    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(kt * 32 + ql) * HLS + 16 * c + 8 * hh]);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[c], st[kt], 0, 0, 0);
      }
    }
    if (t * KT + KT > nkv) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KT + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= nkv) st[kt][r] = -INFINITY;
        }
    }
    float mloc = st[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[kt][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = expf(m_run - m_new);
    l_run *= alpha;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

    // P -> bf16, packed 8 per MFMA step; the row sum uses the ROUNDED probabilities so that the
    // normaliser matches what is multiplied into V.
    bf16x8 pf[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float p = expf(st[kt][8 * u + e] - m_new);
          const unsigned short pb = f2bf(p);
          pf[kt][u][e] = (short)pb;
          l_run += __uint_as_float(((unsigned int)pb) << 16);
        }
    m_run = m_new;

    // O^T[d][q] += sum over the 8 keys of step (kt,u): keys kt*32 + 16u + {0..3, 8..11} + 4hh
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kb = kt * 32 + 16 * u + 4 * hh;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const unsigned short* vp = &Vt[(d * 32 + ql) * HLS + kb];
          const ushort4 lo = *reinterpret_cast<const ushort4*>(vp);
          const ushort4 hi = *reinterpret_cast<const ushort4*>(vp + 8);
          bf16x8 vf;
          vf[0] = (short)lo.x; vf[1] = (short)lo.y; vf[2] = (short)lo.z; vf[3] = (short)lo.w;
          vf[4] = (short)hi.x; vf[5] = (short)hi.y; vf[6] = (short)hi.z; vf[7] = (short)hi.w;
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][u], o[d], 0, 0, 0);
        }
      }
    __syncthreads();
  }

  const float l = l_run + __shfl_xor(l_run, 32);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  float* op = a.out + ((size_t)bs * a.npad + q0 + ql) * a.ldo + h * 64 + 4 * hh;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 w;
      w.x = o[d][4 * g + 0] * inv; w.y = o[d][4 * g + 1] * inv;
      w.z = o[d][4 * g + 2] * inv; w.w = o[d][4 * g + 3] * inv;
      *reinterpret_cast<float4*>(op + d * 32 + 8 * g) = w;
    }
}

// ------------------------------------------------------------------------------------------------
// The production bf16 kernel below (k_attn_bf16_v5) consumes what the projection GEMM's bf16 epilogue already laid
// out -- Q and K as bf16 rows, V as bf16 V^T panels per (slot, head) with keys permuted inside 16-groups -- so the
// tile loop does no conversion and no transposition.  (Generations v2-v4 -- register-staged K/V, hardware bf16
// convert + matrix-pipe denominator, LDS-DMA ring -- were folded into it and retired.)
// ------------------------------------------------------------------------------------------------
// bf16 variant 5: software-pipelined across key tiles.  Measured on gfx950 (tools/probes/overlap.hip): VALU work
// overlaps MFMA execution only when both sit in the SAME wave's instruction stream; a softmax-phase wave and an
// MFMA-phase wave sharing a SIMD do not overlap.  So each iteration issues the QK^T MFMAs of tile t+1 inside the
// exp / convert work of tile t.  K runs one tile ahead of V^T in two 3-deep LDS-DMA rings (prefetch distance two
// tiles for both), one barrier per tile.  The running maximum is updated lazily: the output is rescaled only when
// some query's maximum grew by more than 2^8 (probabilities then stay <= 256, exact in f32 / harmless in bf16),
// which removes the per-tile rescale after the first tiles.
// SPLIT (small grids: one to three pairs are 64 .. 192 workgroups on 256 CUs): gridDim.x = query blocks x a.nsplit; workgroup (qblk, s)
// attends keys [s, s + 1) * ceil(nkv / 64) / nsplit * 64 only and leaves its unnormalised output, reference and denominator in a.part;
// the workgroup that arrives last at the (slot, head, query block)'s ticket merges the nsplit partial results and writes the rows.
// one 32x32x16 matrix instruction on 16-bit operands held as 8 shorts: bf16 (F16 = false) or fp16 (true), f32 accumulate
template <bool F16> __device__ __forceinline__ f32x16 mfma16(const bf16x8& x, const bf16x8& y, const f32x16& c) {
  if constexpr (F16) {
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, x), __builtin_bit_cast(f16x8_t, y), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
  }
}
template <typename T> __device__ __forceinline__ void st_dev(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T ld_dev(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// F16 (GN_PREC_F16X2_F16_ATTN): q, k, V^T and the probabilities are fp16 instead of bf16 -- what the reference's own CUDA path computes (kornia
// casts q, k, v to half for SDPA, SURVEY.md:314): 11 significant bits instead of 8, at the same matrix-pipe rate.  fp16's range is what differs:
// probabilities above 65504 become inf, so the optimistic reference falls back to the exact running maximum (whose lazy update keeps every
// probability <= 2^8) as soon as a score exceeds the reference by ~16 in log2 units (bf16: ~100); probabilities below 2^-24 flush to zero (they
// are below 2^-16 of the row's largest even under the lazy maximum), subnormals are honoured by the matrix pipe.
template <int ABL, int NW, int ND = 3, int NQ = 1, bool SPLIT = false, bool F16 = false>   // NW waves share one K / V^T tile stream; ND = ring depth (3 or 4 tiles; 4 measured no faster);
                                                     // NQ = query tiles of 32 per wave (2: every K / V^T fragment read from LDS feeds two MFMAs)
__global__ __launch_bounds__(64 * NW, NQ == 1 ? 2 : 1) void k_attn16_v5(AttnArgs a) {   // NQ = 1: two waves per SIMD (two workgroups per CU at NW = 4): at most 256 registers; NQ = 2: one wave per SIMD with the whole register file
  constexpr int IPW = 8 / NW;                 // LDS-DMA instructions per wave per 8 KB tile
  __shared__ __attribute__((aligned(1024))) unsigned short smem[2 * ND * kRing + 8 + ((ABL & 512) ? 24576 : 0)];   // ABL 512 (experiment): +48 KB so that only one workgroup fits a CU   // K ring [ND][64 keys][64], V^T ring [ND][64 dims][64 keys], overflow flag

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, ql = lane & 31;
  int qblk, h, bs;
  {
    const int gx = gridDim.x, nwg = gx * gridDim.y * gridDim.z;
    const int L = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    qblk = v % gx;
    const int g = v / gx;
    h = g % kHeads; bs = g / kHeads;
  }
  int split = 0, key_off = 0;
  const int nqb = SPLIT ? (int)gridDim.x / a.nsplit : (int)gridDim.x;
  if (SPLIT) { split = qblk / nqb; qblk -= split * nqb; }
  const int kvs = a.cross ? (bs ^ 1) : bs;
  int nkv = a.nvalid[kvs];
  if (SPLIT) {   // this workgroup's key tiles [t0, t1): from here on the kernel sees them as a key sequence of its own
    const int nt = (nkv + KT - 1) / KT, t0 = split * nt / a.nsplit, t1 = (split + 1) * nt / a.nsplit;
    key_off = t0 * KT;
    nkv = (nkv < t1 * KT ? nkv : t1 * KT) - key_off;
    if (nkv < 0) nkv = 0;
  }
  const int q0 = qblk * (NW * 32 * NQ) + wave * 32 * NQ;
  const int ntiles = (nkv + KT - 1) / KT;

  bf16x8 qf[NQ][4];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const unsigned short* qp = a.qb + ((size_t)bs * a.npad + q0 + 32 * qi + ql) * a.ldqb + h * 64 + 8 * hh;
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[qi][c] = *reinterpret_cast<const bf16x8*>(qp + 16 * c);
  }

  f32x16 o[NQ][2], ol[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[qi][0][r] = 0.f; o[qi][1][r] = 0.f; ol[qi][r] = 0.f; }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (short)(F16 ? 0x3c00 : 0x3f80);
  float m_run[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) m_run[qi] = -INFINITY;

  // LDS-DMA addressing as in variant 4 (source-side swizzle f(row) = (row >> 1) & 7)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned short* ksrc[IPW]; const unsigned short* vsrc[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int r = (IPW * wave + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r ^ (r >> 3)) & 7);
    ksrc[j] = a.kb + ((size_t)kvs * a.npad + key_off + r) * a.ldkb + h * 64 + c * 8;
    vsrc[j] = a.vt + (((size_t)kvs * kHeads + h) * kHeadDim + r) * a.npad + key_off + c * 8;
    if (ABL & 2) {   // timing probe: the same bytes fetched as contiguous 8 KB tiles (wrong data)
      ksrc[j] = a.kb + ((size_t)kvs * kHeads + h) * a.npad * 64 + (IPW * wave + j) * 512 + lane * 8;
      vsrc[j] = a.vt + ((size_t)kvs * kHeads + h) * a.npad * 64 + (IPW * wave + j) * 512 + lane * 8;
    }
  }
  const size_t kstep = (ABL & 2) ? (size_t)KT * 64 : (size_t)KT * a.ldkb;
  const int vstep = (ABL & 2) ? KT * 64 : KT;
#define GN_DMA_K(stage, t)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < IPW; ++j)                                                                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(ksrc[j] + (size_t)(t) * kstep),                                           \
                                     (lptr_t)(smem + (stage) * kRing + (IPW * wave_u + j) * 512), 16, 0, 0);
#define GN_DMA_V(stage, t)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < IPW; ++j)                                                                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(vsrc[j] + (t) * vstep),                                                   \
                                     (lptr_t)(smem + (ND + (stage)) * kRing + (IPW * wave_u + j) * 512), 16, 0, 0);
  int ro[2], fsw[2];     // fragment row offsets (shorts) and swizzle of rows ql, 32 + ql
#pragma unroll
  for (int i2 = 0; i2 < 2; ++i2) {
    const int row = i2 * 32 + ql;
    fsw[i2] = (row ^ (row >> 3)) & 7;
    ro[i2] = row * 64;
  }

  auto qk_tile = [&](f32x16 (&S)[NQ][2], int stage) __attribute__((always_inline)) {
    const unsigned short* Ks = smem + stage * kRing;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[qi][kt][r] = 0.f;
    // the two 32-key halves alternate: consecutive MFMAs never depend on each other
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[ro[kt] + (((2 * c + hh) ^ fsw[kt]) << 3)]);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) S[qi][kt] = mfma16<F16>(kf, qf[qi][c], S[qi][kt]);
      }
  };

  // Optimistic softmax reference.  The running maximum is searched in the first kSearchTiles key tiles only; after that the
  // reference stays where it is and the tile loop has no maximum search, no cross-half shuffle and no ballot in it (they were 8 %
  // of the launch).  Probabilities may then exceed 1: bf16 keeps the f32 exponent range and every sum is f32, so nothing is lost
  // until a score exceeds the reference by ~100 in log2 units.  A workgroup whose denominators left the safe range (or are not
  // finite) runs its tiles again with the maximum searched in every tile -- the exact algorithm, never a wrong result.
  constexpr int kSearchTiles = 2;
  int* const ovf_flag = reinterpret_cast<int*>(smem + 2 * ND * kRing);
  if (tid == 0) *ovf_flag = 0;
  f32x16 sa[NQ][2], sb[NQ][2];
#pragma unroll 1
  for (int attempt = 0; attempt < 2; ++attempt) {
  const bool safe = attempt != 0 || (ABL & 128);
  int s0 = 0, s1 = 1, s2 = 2, s3 = 3;  // ring stages of tiles t, t+1, t+2 (, t+3) modulo ND
  if (ntiles > 0) {
    GN_DMA_K(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ntiles > 1) GN_DMA_K(1, 1);
    GN_DMA_V(0, 0);
    if (ntiles > 2) GN_DMA_K(2, 2);
    if (ntiles > 1) GN_DMA_V(1, 1);
    if (ND == 4) { if (ntiles > 3) GN_DMA_K(3, 3); if (ntiles > 2) GN_DMA_V(2, 2); }
    qk_tile(sa, 0);
  }

  // FAST: a tile in the steady state of the optimistic pass -- both rings are refilled unconditionally, no key of the tile is masked and
  // the reference is not searched: the body has no scalar branch in it (the generic body tests six conditions per tile)
  auto tile = [&](f32x16 (&ST)[NQ][2], f32x16 (&SN)[NQ][2], int t, auto fast_tag) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_tag)::value;
    // K(t+1) and V^T(t) have landed once everything but the ND - 2 newest DMA groups ({K(t+2), V^T(t+1)}, ...) is complete;
    // lgkmcnt(0): this wave's fragment reads of the stages refilled below (issued just in front of the barrier, consumed by MFMAs behind
    // it) have RETURNED before any wave may start the refill -- without it an LDS-DMA that hits in cache can overtake such a read
    // the barrier publishes all waves' shares and proves the stages refilled below are no longer being read
    constexpr int G = 2 * (8 / NW);                       // DMA instructions per wave per group (K tile + V^T tile)
    if (ABL & 32) {
      // timing probe: no workgroup barrier per tile (races on the ring: wrong data)
    } else if (FAST || t + ND - 1 < ntiles) {
      if (G * (ND - 2) == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else if (G * (ND - 2) == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (!(ABL & 1)) {
      if (FAST || t + ND < ntiles) GN_DMA_K(s0, t + ND);                          // K(t) was consumed one iteration ago
      if (FAST || t + ND - 1 < ntiles) GN_DMA_V(ND == 4 ? s3 : s2, t + ND - 1);   // the stage V^T(t-1) was read from
    }
    if (!FAST && t * KT + KT > nkv) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KT + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= nkv) ST[qi][kt][r] = -INFINITY;
          }
    }
    if (!FAST && (safe || t < kSearchTiles)) {   // wave-uniform
      float mloc[NQ];
      bool grow = false;
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        if (ABL & 64) { mloc[qi] = 0.f; if (t == 0) m_run[qi] = 0.f; continue; }   // timing probe: no maximum search (wrong for large scores)
        float m = fmaxf(ST[qi][0][0], ST[qi][1][0]);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 1; r < 16; r += 2) m = fmaxf(fmaxf(m, ST[qi][kt][r]), ST[qi][kt][(r + 1) & 15]);   // v_max3_f32
        m = fmaxf(m, __shfl_xor(m, 32));
        mloc[qi] = m;
        grow = grow || (m - m_run[qi]) * kLog2e > 8.0f;
      }
      // lazy maximum: keep the stale reference unless some query's maximum grew by more than 2^8
      if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
          const float m_new = fmaxf(m_run[qi], mloc[qi]);
          const float alpha = __builtin_amdgcn_exp2f((m_run[qi] - m_new) * kLog2e);
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[qi][0][r] *= alpha; o[qi][1][r] *= alpha; ol[qi][r] *= alpha; }
          m_run[qi] = m_new;
        }
      }
    }
    float mneg[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) mneg[qi] = -m_run[qi] * kLog2e;

    // scores of the next tile on the matrix pipe while this tile's probabilities are computed on the VALU
    if (ABL & 4) __builtin_amdgcn_s_setprio(1);
    qk_tile(SN, s1);
    if (ABL & 4) __builtin_amdgcn_s_setprio(0);

    bf16x8 pf[NQ][2][2];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          u32x4 pw;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f32x2 x2;
            if (ABL & 8) {   // developer variant 46: one v_pk_fma_f32 for the pair -- measured 5 % SLOWER (114.3 vs 109.1 us on one box)
              x2 = __builtin_elementwise_fma((f32x2){ST[qi][kt][8 * u + 2 * e], ST[qi][kt][8 * u + 2 * e + 1]}, (f32x2){kLog2e, kLog2e}, (f32x2){mneg[qi], mneg[qi]});
            } else {
              x2[0] = __builtin_fmaf(ST[qi][kt][8 * u + 2 * e], kLog2e, mneg[qi]);
              x2[1] = __builtin_fmaf(ST[qi][kt][8 * u + 2 * e + 1], kLog2e, mneg[qi]);
            }
            f32x2 p;
            if (ABL & 16) {   // timing probe: no transcendental (wrong data)
              p = x2;
            } else {
              p[0] = __builtin_amdgcn_exp2f(x2[0]);
              p[1] = __builtin_amdgcn_exp2f(x2[1]);
            }
            pw[e] = pack16<F16>(p[0], p[1]);   // v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (RNE)
          }
          pf[qi][kt][u] = __builtin_bit_cast(bf16x8, pw);
        }
    const unsigned short* Vs = smem + (ND + s0) * kRing;
    if (ABL & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) ol[qi] = mfma16<F16>(ones, pf[qi][kt][u], ol[qi]);   // softmax denominator (rounded p)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vs[ro[d] + (((4 * kt + 2 * u + hh) ^ fsw[d]) << 3)]);
#pragma unroll
          for (int qi = 0; qi < NQ; ++qi) o[qi][d] = mfma16<F16>(vf, pf[qi][kt][u], o[qi][d]);
        }
      }
    if (ABL & 4) __builtin_amdgcn_s_setprio(0);
    const int s_ = s0; s0 = s1; s1 = s2;
    if (ND == 4) { s2 = s3; s3 = s_; } else { s2 = s_; }
  };

  {
    int t = 0;
    if (!safe) {
      for (; t < kSearchTiles && t < ntiles; t += 2) {
        tile(sa, sb, t, std::false_type{});
        if (t + 1 < ntiles) tile(sb, sa, t + 1, std::false_type{});
      }
#pragma unroll 1
      for (; (ABL & 256) && t + 1 + ND < ntiles; t += 2) {   // tiles t and t + 1 are both in the steady state
        tile(sa, sb, t, std::true_type{});
        tile(sb, sa, t + 1, std::true_type{});
      }
    }
#pragma unroll 1
    for (; t < ntiles; t += 2) {
      tile(sa, sb, t, std::false_type{});
      if (t + 1 < ntiles) tile(sb, sa, t + 1, std::false_type{});
    }
  }
  if (safe) break;
  // did the optimistic reference hold for every query of the workgroup?  (the K / V^T rings are shared: all waves repeat or none)
  bool bad = false;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) bad = bad || !(ol[qi][0] < 1e30f);
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) *ovf_flag = 1;
  __syncthreads();
  if (*ovf_flag == 0) break;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[qi][0][r] = 0.f; o[qi][1][r] = 0.f; ol[qi][r] = 0.f; }
    m_run[qi] = -INFINITY;
  }
  }   // attempt
#undef GN_DMA_K
#undef GN_DMA_V

  if (SPLIT) {
    static_assert(!SPLIT || NQ == 1, "split keys: one query tile per wave");
    // partial result of this wave: [34][64] floats = registers o[0][0..15], o[1][0..15], reference, denominator of lane l at [.][l]
    const int S = a.nsplit;
    const size_t entry = (((size_t)bs * kHeads + h) * nqb + qblk) * S;
    float* const mine = a.part + ((entry + split) * NW + wave) * (34 * 64) + lane;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) st_dev(mine + (16 * d + r) * 64, o[0][d][r]);
    st_dev(mine + 32 * 64, m_run[0]);
    st_dev(mine + 33 * 64, ol[0][0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* const last = reinterpret_cast<int*>(smem + 2 * ND * kRing) + 1;
    if (tid == 0) {
      const unsigned int tk = atomicAdd(a.tickets + entry / S, 1u);
      *last = tk == (unsigned)(S - 1);
      if (tk == (unsigned)(S - 1)) a.tickets[entry / S] = 0;   // ready for the next launch
    }
    __syncthreads();
    if (!*last) return;
    float mm[4], ll[4], M = -INFINITY;     // a.nsplit <= 4
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mm[i] = -INFINITY; ll[i] = 0.f;
      if (i < S) {
        const float* p = a.part + ((entry + i) * NW + wave) * (34 * 64) + lane;
        mm[i] = ld_dev(p + 32 * 64); ll[i] = ld_dev(p + 33 * 64);
        M = fmaxf(M, ll[i] > 0.f ? mm[i] : -INFINITY);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][0][r] = 0.f; o[0][1][r] = 0.f; }
    float L = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < S && ll[i] > 0.f) {                             // (a split without keys has l = 0)
        const float* p = a.part + ((entry + i) * NW + wave) * (34 * 64) + lane;
        const float al = __builtin_amdgcn_exp2f((mm[i] - M) * kLog2e);
        L += ll[i] * al;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[0][d][r] += ld_dev(p + (16 * d + r) * 64) * al;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ol[0][r] = L;
  }

#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
  const float l = ol[qi][0];
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  if (a.outp != nullptr) {   // hm16 rows (x = xh + xm) for the f16x2 out-projection GEMM
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const size_t row = (size_t)bs * a.npad + q0 + 32 * qi + ql;
    float amax = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4v w = {o[qi][d][4 * g + 0] * inv, o[qi][d][4 * g + 1] * inv, o[qi][d][4 * g + 2] * inv, o[qi][d][4 * g + 3] * inv};
        ovf_track(amax, w.x, w.y); ovf_track(amax, w.z, w.w);
        const f16x4 hv = __builtin_convertvector(w, f16x4);
        const f16x4 mv = __builtin_convertvector(w - __builtin_convertvector(hv, f32x4v), f16x4);
        uint16_t* pp = a.outp + hm16_off(row, a.ldo, h * 64 + d * 32 + 8 * g + 4 * hh);
        *reinterpret_cast<f16x4*>(pp) = hv;
        *reinterpret_cast<f16x4*>(pp + 16) = mv;
      }
    ovf_commit(a.ovf, amax);
    continue;
  }
  float* op = a.out + ((size_t)bs * a.npad + q0 + 32 * qi + ql) * a.ldo + h * 64 + 4 * hh;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 w;
      w.x = o[qi][d][4 * g + 0] * inv; w.y = o[qi][d][4 * g + 1] * inv;
      w.z = o[qi][d][4 * g + 2] * inv; w.w = o[qi][d][4 * g + 3] * inv;
      *reinterpret_cast<float4*>(op + d * 32 + 8 * g) = w;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// k_attn_ks (round 5): attention of ONE or TWO pairs -- the reference's own operating point (one pair per message, pose_node.py:178-184).
// k_attn16_v5 gives a wave 32 queries and ALL keys: at one pair that is 64 workgroups on a quarter of the chip, each wave walking 16 key tiles one
// after the other (21 us per launch, of which ~4 us are MFMAs).  Here the four waves of a workgroup share the SAME 32 queries and take every fourth
// 64-key tile each; the partial results (running reference, denominator, 64 x 32 output tile per wave) are merged through LDS, in a fixed order.
// 8 x the workgroups (npad / 32 per (slot, head)), a quarter of the serial tile chain per wave.  No LDS ring and no barrier in the tile loop: the
// K rows and V^T panels of a (slot, head) are 256 KB and stay in L2, every wave reads its own fragments straight into registers (16-byte loads:
// a K row's 64 dims and a V^T row's 64 keys are one 128-byte line each), one tile ahead.  Same operands, operand rounding and per-tile arithmetic
// as k_attn16_v5's exact path (maximum searched in every tile, lazy reference, denominators from the rounded probabilities on the matrix pipe);
// the partial sums are combined in another order, so context rows differ from k_attn16_v5's in the last bits (as between any two attention
// kernels of this library; correspondences: tests/test_gpu_round5.py).
// NW = 8 (the shipped form): eight waves, two per SIMD, every eighth tile each (two tiles at 1024 keys: 128 registers of fragments) -- one
// wave's probability arithmetic runs under the other's MFMAs; NW = 4: one wave per SIMD, four tiles each.
template <bool F16, int NW>
__global__ __launch_bounds__(64 * NW) void k_attn_ks(AttnArgs a) {
  constexpr int TP = 16 / NW;      // tiles of a wave per pass, all requested up front
  __shared__ __attribute__((aligned(16))) float part[NW][34][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, bs = blockIdx.z;
  const int kvs = a.cross ? (bs ^ 1) : bs;
  const int nkv = a.nvalid[kvs];
  const int q0 = blockIdx.x * 32;
  const int ntiles = (nkv + KT - 1) / KT;

  bf16x8 qf[4];
  {
    const unsigned short* qp = a.qb + ((size_t)bs * a.npad + q0 + ql) * a.ldqb + h * 64 + 8 * hh;
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const bf16x8*>(qp + 16 * c);
  }
  f32x16 o[2], ol;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; ol[r] = 0.f; }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (short)(F16 ? 0x3c00 : 0x3f80);
  float m_run = -INFINITY;

  // fragments of key tile t: K rows 32 kt + ql, dims 16 c + 8 hh ..; V^T rows (dims) 32 d + ql, keys 8 (4 kt + 2 u + hh) ..
  const unsigned short* const kbase = a.kb + ((size_t)kvs * a.npad + ql) * a.ldkb + h * 64 + 8 * hh;
  const unsigned short* const vbase = a.vt + (((size_t)kvs * kHeads + h) * kHeadDim + ql) * a.npad + 8 * hh;
  // ALL of a wave's tiles are requested before the first one is used (up to four per pass: 64 KB of fragments, 256 registers of a lone wave's 512).
  // A (slot, head)'s K rows and V^T panels were written by the projection launch just before, from other XCDs: every first touch is an L2 miss,
  // and with one tile of look-ahead the four tiles of a wave cost four miss latencies one behind the other (12.0 -> 9.x us per launch).
  bf16x8 kf[TP][2][4], vf[TP][2][2][2];     // [tile of the pass][kt][c], [tile of the pass][kt][u][d]
  auto load_tile = [&](int buf, int t) __attribute__((always_inline)) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int c = 0; c < 4; ++c) kf[buf][kt][c] = *reinterpret_cast<const bf16x8*>(kbase + (size_t)(t * KT + 32 * kt) * a.ldkb + 16 * c);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int d = 0; d < 2; ++d) vf[buf][kt][u][d] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)(32 * d) * a.npad + t * KT + 8 * (4 * kt + 2 * u));
  };
  auto tile = [&](int buf, int t) __attribute__((always_inline)) {
    f32x16 S[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) S[kt] = mfma16<F16>(kf[buf][kt][c], qf[c], S[kt]);
    if (t * KT + KT > nkv) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KT + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= nkv) S[kt][r] = -INFINITY;
        }
    }
    float m = fmaxf(S[0][0], S[1][0]);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 1; r < 16; r += 2) m = fmaxf(fmaxf(m, S[kt][r]), S[kt][(r + 1) & 15]);
    m = fmaxf(m, __shfl_xor(m, 32));
    // lazy maximum: keep the stale reference unless some query's maximum grew by more than 2^8 (k_attn16_v5's exact path)
    if (__builtin_amdgcn_ballot_w64((m - m_run) * kLog2e > 8.0f) != 0) {
      const float m_new = fmaxf(m_run, m);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; ol[r] *= alpha; }
      m_run = m_new;
    }
    const float mneg = -m_run * kLog2e;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4 pw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][8 * u + 2 * e], kLog2e, mneg));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][8 * u + 2 * e + 1], kLog2e, mneg));
          pw[e] = pack16<F16>(p0, p1);
        }
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
        ol = mfma16<F16>(ones, pf, ol);
#pragma unroll
        for (int d = 0; d < 2; ++d) o[d] = mfma16<F16>(vf[buf][kt][u][d], pf, o[d]);
      }
  };
  // wave w takes tiles w, w + NW, ...: passes of up to TP tiles, all requested up front
#pragma unroll 1
  for (int t0 = wave; t0 < ntiles; t0 += 16) {
#pragma unroll
    for (int i = 0; i < TP; ++i)
      if (t0 + NW * i < ntiles) load_tile(i, t0 + NW * i);
#pragma unroll
    for (int i = 0; i < TP; ++i)
      if (t0 + NW * i < ntiles) tile(i, t0 + NW * i);
  }
  // ---- merge the four waves' partial results: [34][64] floats per wave = o[0][0..15], o[1][0..15], reference, denominator of lane l at [.][l]
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][16 * d + r][lane] = o[d][r];
  part[wave][32][lane] = m_run;
  part[wave][33][lane] = ol[0];
  __syncthreads();
  float al[NW], M = -INFINITY, L = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) M = fmaxf(M, part[i][33][lane] > 0.f ? part[i][32][lane] : -INFINITY);
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const float li = part[i][33][lane];
    al[i] = li > 0.f ? __builtin_amdgcn_exp2f((part[i][32][lane] - M) * kLog2e) : 0.f;     // (a wave without keys has l = 0)
    L += li * al[i];
  }
  const float inv = L > 0.f ? 1.0f / L : 0.f;
  // the NW waves share the 2 x 4 groups of four output registers: wave w finishes groups GP w .. GP w + GP - 1 (half d = group >> 2, g = group & 3)
  constexpr int GP = 8 / NW;
  float amax = 0.f;
#pragma unroll
  for (int gg = 0; gg < GP; ++gg) {
    const int d = (GP * wave + gg) >> 2, g = (GP * wave + gg) & 3;
    float w4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) acc += part[i][16 * d + 4 * g + e][lane] * al[i];
      w4[e] = acc * inv;
    }
    const size_t row = (size_t)bs * a.npad + q0 + ql;
    if (a.outp != nullptr) {
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      const f32x4v w = {w4[0], w4[1], w4[2], w4[3]};
      ovf_track(amax, w.x, w.y); ovf_track(amax, w.z, w.w);
      const f16x4 hv = __builtin_convertvector(w, f16x4);
      const f16x4 mv = __builtin_convertvector(w - __builtin_convertvector(hv, f32x4v), f16x4);
      uint16_t* pp = a.outp + hm16_off(row, a.ldo, h * 64 + d * 32 + 8 * g + 4 * hh);
      *reinterpret_cast<f16x4*>(pp) = hv;
      *reinterpret_cast<f16x4*>(pp + 16) = mv;
    } else {
      *reinterpret_cast<float4*>(a.out + row * a.ldo + h * 64 + 4 * hh + d * 32 + 8 * g) = make_float4(w4[0], w4[1], w4[2], w4[3]);
    }
  }
  if (a.outp != nullptr) ovf_commit(a.ovf, amax);
}
}  // namespace

int g_attn_f32_ks = 0;       // developer knob 43: bits 0-1: 0 = k_attn_f32_ks for one or two pairs per call, 1 = never, 2 = always; bit 2: its eight-wave form
void launch_attention_f32(const AttnArgs& a, hipStream_t s) {
  // (the choice depends on the number of pairs only, like k_attn_ks': padding does not change the kernel family)
  if ((g_attn_f32_ks & 3) == 2 || ((g_attn_f32_ks & 3) == 0 && a.BS <= 4)) {
    // (knob 43 bit 2: eight waves per workgroup instead of four -- measured: 34.9 against 35.8 us at one pair, 62.6 against 58.0 at two: not shipped)
    if (g_attn_f32_ks & 4) hipLaunchKernelGGL(k_attn_f32_ks<8>, dim3(a.npad / 32, kHeads, a.BS), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(k_attn_f32_ks<4>, dim3(a.npad / 32, kHeads, a.BS), dim3(256), 0, s, a);
    g_last_kernel = "k_attn_f32_ks(";
    return;
  }
  dim3 grid(a.npad / QB, kHeads, a.BS), block(256);
  hipLaunchKernelGGL(k_attn_f32, grid, block, 0, s, a);
  g_last_kernel = "k_attn_f32(";
}

void launch_attention_bf16(const AttnArgs& a, hipStream_t s) {
  dim3 grid(a.npad / QB, kHeads, a.BS), block(256);
  hipLaunchKernelGGL(k_attn_bf16, grid, block, 0, s, a);
  g_last_kernel = "k_attn_bf16(";
}

}  // namespace gn

namespace gn {
namespace {
// developer / test entry (gn_debug_attention): f32 q | k | v rows -> what k_attn_bf16_v5 reads (bf16 q * qscale and k rows, V^T panels
// with the keys permuted inside 16-groups), i.e. what the projection epilogues of the matcher write
__global__ void k_pack_attn_bf16(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float qscale,
                                 uint16_t* qb, uint16_t* kb, int ldb, uint16_t* vt, long long ntok, int npad, int f16) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ntok * 32) return;
  const long long tok = idx >> 5;
  const int g8 = (int)(idx & 31), bs = (int)(tok / npad), i = (int)(tok % npad);
  const int r = i & 15, sp = (i & ~15) + ((r & 3) | ((r & 4) << 1) | ((r & 8) >> 1));   // key r of a 16-group is stored at 0..3, 8..11, 4..7, 12..15
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int d = 8 * g8 + e;
    const float qf = q[tok * ldq + d] * qscale, kf = k[tok * ldk + d], vf = v[tok * ldv + d];
    qb[tok * ldb + d] = (unsigned short)(pack16_rt(qf, 0.f, f16) & 0xffffu);
    kb[tok * ldb + d] = (unsigned short)(pack16_rt(kf, 0.f, f16) & 0xffffu);
    vt[(((size_t)bs * kHeads + (d >> 6)) * kHeadDim + (d & 63)) * npad + sp] = (unsigned short)(pack16_rt(vf, 0.f, f16) & 0xffffu);
  }
}
}  // namespace
void launch_pack_attn_bf16(const AttnArgs& a, uint16_t* qkb, uint16_t* vtb, hipStream_t s) {
  const long long ntok = (long long)a.BS * a.npad;
  hipLaunchKernelGGL(k_pack_attn_bf16, dim3((unsigned)((ntok * 32 + 255) / 256)), dim3(256), 0, s, a.q, a.ldq, a.k, a.ldk, a.v, a.ldv, a.qscale,
                     qkb, qkb + kDim, 2 * kDim, vtb, ntok, a.npad, a.half_fmt);
}
thread_local long long* g_attn_stamps = nullptr;
thread_local int g_attn_variant = 4;  // developer knob: 4 = k_attn_bf16_v5 (4 waves per block, default), 48 = 8 waves per block, 43 = 4-deep rings, 41 / 42 = timing-only ablations
void launch_attention_bf16_v2(const AttnArgs& a, hipStream_t s) {
  // fp16 mode, grids of at least one 256-query workgroup per CU: k_attn_pw (one wave per SIMD, pinned instruction stream; 31 vs 36 us at 8 pairs x
  // 1024 keypoints, 112 vs 125 us at 32, 224 vs 244 us at 64; below that k_attn16_v5's two workgroups per CU and its key splits win: 27 vs 24 us at 4
  // pairs).  Knob 1: 4 = this choice, 5 = k_attn16_v5 always, 70 = k_attn_pw whenever npad % 256 == 0, 71.. / 1000.. = its timing variants.
  const bool pw_auto = g_attn_variant == 4 && a.half_fmt && a.npad % 256 == 0 && (long long)(a.npad / 256) * kHeads * a.BS >= 256;
  if ((pw_auto || (g_attn_variant >= 70 && g_attn_variant <= 73) || g_attn_variant >= 1000) && !(a.nsplit > 1 && a.part != nullptr) &&
      launch_attention_pw(a, pw_auto ? 0 : g_attn_variant >= 1000 ? g_attn_variant - 900 : g_attn_variant - 70, s)) return;
  // ONE pair (k_attn16_v5's grid leaves three quarters of the CUs idle; measured 14.4 vs 21.1 us per launch at 1024 keypoints; at two pairs the
  // two kernels tie): the four waves of a workgroup split the KEYS of 32 queries (k_attn_ks).  The choice depends on the number of pairs only,
  // never on the padded length: gn_set_active_kpts must not change a result bit (test_active_kpts_padding_does_not_change_results).
  // Knob 1: 80 = always, 81 = always in the four-wave form, 5 = never.
  {
    const bool ks_auto = g_attn_variant == 4 && a.BS <= 2;
    if ((ks_auto || g_attn_variant == 80 || g_attn_variant == 81) && !(a.nsplit > 1 && a.part != nullptr) && a.npad % 64 == 0) {
      const dim3 grid(a.npad / 32, kHeads, a.BS);
      if (g_attn_variant == 81) {
        if (a.half_fmt) { hipLaunchKernelGGL((k_attn_ks<true, 4>), grid, dim3(256), 0, s, a); g_last_kernel = "k_attn_ks<true, 4>"; }
        else { hipLaunchKernelGGL((k_attn_ks<false, 4>), grid, dim3(256), 0, s, a); g_last_kernel = "k_attn_ks<false, 4>"; }
      } else {
        if (a.half_fmt) { hipLaunchKernelGGL((k_attn_ks<true, 8>), grid, dim3(512), 0, s, a); g_last_kernel = "k_attn_ks<true, 8>"; }
        else { hipLaunchKernelGGL((k_attn_ks<false, 8>), grid, dim3(512), 0, s, a); g_last_kernel = "k_attn_ks<false, 8>"; }
      }
      return;
    }
  }
  g_last_kernel = "k_attn16_v5<0, 4, 3, 1, false, false>";   // the name rocprofv3 prints (profiles up to r02m: "k_attn_bf16_v5<0, 4, 3, 1, false>")
  if (a.half_fmt) {   // fp16 operands (GN_PREC_F16X2_F16_ATTN)
    if (a.nsplit > 1 && a.part != nullptr && a.tickets != nullptr) {
      hipLaunchKernelGGL((k_attn16_v5<0, 4, 3, 1, true, true>), dim3(a.npad / 128 * a.nsplit, kHeads, a.BS), dim3(256), 0, s, a);
      g_last_kernel = "k_attn16_v5<0, 4, 3, 1, true, true>";
      return;
    }
    // fp16 probabilities overflow at 65504: the optimistic reference (searched in the first two key tiles only) has to fall back whenever a later
    // score exceeds it by ~11, which the bench's scenes do often enough that the ALWAYS-exact running maximum is the faster kernel here
    // (measured on one box: 4549 vs 4524 pairs/s; bf16, whose range lets the optimistic pass through, 4590).  It is also exactly what the
    // reference's fp16 SDPA computes: probabilities relative to the running maximum.  Knob 1 = 60 selects the optimistic variant.
    if (g_attn_variant == 60) { hipLaunchKernelGGL((k_attn16_v5<0, 4, 3, 1, false, true>), dim3(a.npad / 128, kHeads, a.BS), dim3(256), 0, s, a); g_last_kernel = "k_attn16_v5<0, 4, 3, 1, false, true>"; }
    else { hipLaunchKernelGGL((k_attn16_v5<128, 4, 3, 1, false, true>), dim3(a.npad / 128, kHeads, a.BS), dim3(256), 0, s, a); g_last_kernel = "k_attn16_v5<128, 4, 3, 1, false, true>"; }
    return;
  }
  if (g_attn_variant == 48 && a.npad % 256 == 0) {   // experiment: 8 waves share each K / V^T tile (half the L2 -> LDS traffic per query); measured 6 % SLOWER
    dim3 grid(a.npad / 256, kHeads, a.BS), block(512);
    switch (g_attn_variant) {
      default: hipLaunchKernelGGL((k_attn16_v5<0, 8>), grid, block, 0, s, a); break;
    }
    return;
  }
  if (a.nsplit > 1 && a.part != nullptr && a.tickets != nullptr && (g_attn_variant == 4 || g_attn_variant == 59)) {
    hipLaunchKernelGGL((k_attn16_v5<0, 4, 3, 1, true>), dim3(a.npad / 128 * a.nsplit, kHeads, a.BS), dim3(256), 0, s, a);
    g_last_kernel = "k_attn16_v5<0, 4, 3, 1, true, false>";
    return;
  }
  dim3 grid(a.npad / 128, kHeads, a.BS), block(256);
  switch (g_attn_variant) {
    case 43: hipLaunchKernelGGL((k_attn16_v5<0, 4, 4>), grid, block, 0, s, a); break;  // 4-deep rings
    case 41: hipLaunchKernelGGL((k_attn16_v5<1, 4>), grid, block, 0, s, a); break;   // timing-only ablations
    case 42: hipLaunchKernelGGL((k_attn16_v5<2, 4>), grid, block, 0, s, a); break;
    case 44: hipLaunchKernelGGL((k_attn16_v5<4, 4>), grid, block, 0, s, a); break;
    case 46: hipLaunchKernelGGL((k_attn16_v5<8, 4>), grid, block, 0, s, a); break;    // experiment: packed fma in front of the exponentials (5 % slower)
    case 51: hipLaunchKernelGGL((k_attn16_v5<16, 4>), grid, block, 0, s, a); break;    // timing probes (wrong results): no exponentials
    case 52: hipLaunchKernelGGL((k_attn16_v5<32, 4>), grid, block, 0, s, a); break;    //   no per-tile barrier
    case 53: hipLaunchKernelGGL((k_attn16_v5<64, 4>), grid, block, 0, s, a); break;    //   no maximum search
    case 54: hipLaunchKernelGGL((k_attn16_v5<112, 4>), grid, block, 0, s, a); break;   //   none of the three
    case 55: hipLaunchKernelGGL((k_attn16_v5<33, 4>), grid, block, 0, s, a); break;    //   no barrier, no DMA
    case 58: hipLaunchKernelGGL((k_attn16_v5<512, 4>), grid, block, 0, s, a); break;   // experiment: one workgroup (one wave per SIMD) per CU
    case 57: hipLaunchKernelGGL((k_attn16_v5<256, 4>), grid, block, 0, s, a); break;   // steady-state tiles through a branch-free body in a loop of their own: <= 1 % faster, not the default (it is what exposed the ring race fixed by lgkmcnt(0) in front of the barriers)
    case 56: hipLaunchKernelGGL((k_attn16_v5<128, 4>), grid, block, 0, s, a); break;   // the maximum searched in every tile (the exact path a workgroup falls back to)
    case 45: if (a.npad % 256 == 0) { hipLaunchKernelGGL((k_attn16_v5<0, 4, 3, 2>), dim3(a.npad / 256, kHeads, a.BS), block, 0, s, a); break; }   // experiment: two query tiles per wave (every K / V^T fragment feeds two MFMAs, one wave per SIMD): bit-identical output, 27 % SLOWER (144 vs 113 us)
             hipLaunchKernelGGL((k_attn16_v5<0, 4>), grid, block, 0, s, a); break;   // experiment: s_setprio(1) around the MFMA clusters (measured 3 % SLOWER: 107 vs 104 us)
    default: hipLaunchKernelGGL((k_attn16_v5<0, 4>), grid, block, 0, s, a); break;
  }
}
}  // namespace gn
