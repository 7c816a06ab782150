// f32 projection / FFN / similarity GEMMs of the LightGlue matcher on the gfx950 f32 MFMA
// (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 157 TF peak).
//
//   Y[M][N] = A[M][K] * W[N][K]^T (+ bias) (+ epilogue)          "NT" GEMM, K contiguous in both
//
// Stands in for the nn.Linear calls inside kornia's LightGlue (input_proj, Wqkv, out_proj, to_qk,
// to_v, to_out, ffn.0, ffn.3, final_proj) and the `einsum("bmd,bnd->bmn")` similarity of
// MatchAssignment -- all reached from ros/gisnav/gisnav/core/pose_node.py:285-287.
//
// Tiling: 128x128 block tile, 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA tiles of 32x32, BK = 32.
// The MFMA's k index is free to permute as long as A and B agree, so each lane fetches FOUR
// consecutive k with one ds_read_b128 and feeds them to four back-to-back MFMAs: per 8-deep k chunk a
// wave issues 4 LDS reads for 16 MFMAs (1024 matrix-pipe cycles).
// This file holds the f32-input generations: k_gemm_f32_v3 (exact-f32 MFMA), k_gemm_f32x3 (3 x bf16 split) and
// k_gemm_f16x2 (2 x fp16 split on the fly); the shipped f16x2 path with pre-split operands is gn_gemm_p2.hip.
// (Earlier register-staged generations v0-v2 were retired once v3 covered every epilogue.)
#include "gn_common.h"

namespace gn {

namespace {
constexpr int BM = 128, BN = 128, BK = 32;

// two f32 -> one dword of two bf16, round to nearest even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}

constexpr int ES = 68;  // epilogue slab row stride (floats): each wave parks its 64x64 accumulator block in LDS and re-reads it row-wise

// ------------------------------------------------------------------------------------------------
// Variant 3: tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction),
// no VGPR staging and no ds_write pass (ablation: the register-staged copy cost ~20 % of the loop).
// The LDS-DMA writes lane-linear (base + 16 * lane), so rows are stored unpadded (128 B) and the bank
// conflicts are removed by an XOR swizzle applied on the SOURCE side: 16-byte chunk c of tile row r
// lands at chunk position c ^ f(r), f(r) = (r ^ (r >> 3)) & 7, and the fragment reads apply the same
// involution.  With it every 16-lane ds_read_b128 group touches 16 distinct 16-byte bank slots.
// Schedule per k-tile: issue the DMA of tile t+1 into the idle buffer, chunks 0-2 with register
// double-buffered fragments, barrier (its release drains the DMA: tile t+1 has landed), fetch the
// first fragments of tile t+1, chunk 3.
#ifndef GN_V3_LDS_PAD
#define GN_V3_LDS_PAD 0
#endif
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int EPI, bool PIN = true>
__global__ __launch_bounds__(256) void k_gemm_f32_v3(GemmArgs a) {
  constexpr int TILE = (BM + BN) * BK;                 // floats per buffer (unpadded rows of 32 floats)
  constexpr int SLAB = 4 * 64 * ES;
  __shared__ __attribute__((aligned(16))) float smem[((2 * TILE > SLAB) ? 2 * TILE : SLAB) + GN_V3_LDS_PAD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int bx, by;
  {
    const int gx = gridDim.x, nwg = gx * gridDim.y;
    const int L = blockIdx.y * gx + blockIdx.x;
    const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    bx = v % gx; by = v / gx;
  }
  const int bm = by * BM, bn = bx * BN;
  if (a.mlim != nullptr && (a.mlim_seg > 0 ? bm % a.mlim_seg : bm) >= a.mlim[0] * a.mlim_mul) return;     // GemmArgs::mlim: a tile of rows nobody will read
  const float* A = a.A + (long long)blockIdx.z * a.strideA;
  const float* W = a.W + (long long)blockIdx.z * a.strideW;
  float* Y = a.Y + (long long)blockIdx.z * a.strideY;
  const float* const A2 = a.A2;
  const int lda = a.lda, lda2 = a.lda2, ldw = a.ldw, K1 = a.K1, K = a.K;

  // DMA source addressing: wave w stages tile rows [32w, 32w + 32) of A and of B, 8 rows per instruction.
  // lane -> (row offset lane >> 3, chunk position lane & 7) fetches source chunk pos ^ f(row).
  const int drow = wave * 32 + (lane >> 3);            // + 8 * j
  const int dpos = lane & 7;
  // f(row) for row = drow + 8j: (row ^ (row >> 3)) & 7; row >> 3 = 4 * wave + j
  const float* asrc[4]; const float* a2src[4]; const float* wsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = drow + 8 * j;
    const int c = dpos ^ ((row ^ (row >> 3)) & 7);
    asrc[j] = A + (size_t)(bm + row) * lda + c * 4;
    a2src[j] = A2 ? A2 + (size_t)(bm + row) * lda2 + c * 4 - K1 : nullptr;
    wsrc[j] = W + (size_t)(bn + row) * ldw + c * 4;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define GN_DMA_TILE(buf, k0)                                                                      \
  {                                                                                               \
    const bool second = (A2 != nullptr) && ((k0) >= K1);                                          \
    float* la_ = smem + (buf) * TILE + (wave_u * 32) * BK;                                        \
    float* lb_ = la_ + BM * BK;                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
      const float* ga_ = (second ? a2src[j] : asrc[j]) + (k0);                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)ga_, (lptr_t)(la_ + j * 8 * BK), 16, 0, 0);       \
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + (k0)), (lptr_t)(lb_ + j * 8 * BK), 16, 0, 0); \
    }                                                                                             \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addressing (floats): row * 32 + 4 * (chunk ^ f(row)), chunk = 2 kc + hh
  const int hh = lane >> 5;
  int arow_[2], brow_[2], ga_[2], gb_[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra_ = wr * 64 + 32 * i + (lane & 31), rb_ = wc * 64 + 32 * i + (lane & 31);
    arow_[i] = ra_ * BK; brow_[i] = BM * BK + rb_ * BK;
    ga_[i] = hh ^ ((ra_ ^ (ra_ >> 3)) & 7); gb_[i] = hh ^ ((rb_ ^ (rb_ >> 3)) & 7);
  }
  const int nt = K / BK;

#define GN_FRAG_READ(fa, fb, buf, kc)                                                             \
  {                                                                                               \
    const float* b_ = smem + (buf) * TILE;                                                        \
    fa[0] = *reinterpret_cast<const f32x4*>(b_ + arow_[0] + 4 * (((kc) << 1) ^ ga_[0]));          \
    fa[1] = *reinterpret_cast<const f32x4*>(b_ + arow_[1] + 4 * (((kc) << 1) ^ ga_[1]));          \
    fb[0] = *reinterpret_cast<const f32x4*>(b_ + brow_[0] + 4 * (((kc) << 1) ^ gb_[0]));          \
    fb[1] = *reinterpret_cast<const f32x4*>(b_ + brow_[1] + 4 * (((kc) << 1) ^ gb_[1]));          \
  }
#define GN_MFMA4(i, j, fa, fb)                                                                    \
  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);         \
  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);         \
  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);         \
  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
#define GN_MFMA16(fa, fb) { GN_MFMA4(0, 0, fa, fb) GN_MFMA4(0, 1, fa, fb) GN_MFMA4(1, 0, fa, fb) GN_MFMA4(1, 1, fa, fb) }

  GN_DMA_TILE(0, 0);
  { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
  GN_FRAG_READ(fa0, fb0, 0, 0);

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    GN_FRAG_READ(fa1, fb1, cur, 1);
    if (t + 1 < nt) {
      // chunk 0 with the 8 DMA pieces of tile t+1 threaded between the MFMA groups
      const int k0 = (t + 1) * BK;
      const bool second = (A2 != nullptr) && (k0 >= K1);
      float* la_ = smem + (cur ^ 1) * TILE + (wave_u * 32) * BK;
      float* lb_ = la_ + BM * BK;
#define GN_DMA_PAIR(j)                                                                            \
      __builtin_amdgcn_global_load_lds((gptr_t)((second ? a2src[j] : asrc[j]) + k0), (lptr_t)(la_ + (j) * 8 * BK), 16, 0, 0); \
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + k0), (lptr_t)(lb_ + (j) * 8 * BK), 16, 0, 0);
      GN_MFMA4(0, 0, fa0, fb0)
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      GN_DMA_PAIR(0)
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      GN_MFMA4(0, 1, fa0, fb0)
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      GN_DMA_PAIR(1)
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      GN_MFMA4(1, 0, fa0, fb0)
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      GN_DMA_PAIR(2)
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      GN_MFMA4(1, 1, fa0, fb0)
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      GN_DMA_PAIR(3)
      if (PIN) __builtin_amdgcn_sched_barrier(0);
#undef GN_DMA_PAIR
    } else {
      GN_MFMA16(fa0, fb0)
    }
    GN_FRAG_READ(fa0, fb0, cur, 2);
    GN_MFMA16(fa1, fb1)
    GN_FRAG_READ(fa1, fb1, cur, 3);
    GN_MFMA16(fa0, fb0)
    { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    if (t + 1 < nt) GN_FRAG_READ(fa0, fb0, cur ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    GN_MFMA16(fa1, fb1)
  }
#undef GN_DMA_TILE
#undef GN_FRAG_READ
#undef GN_MFMA4
#undef GN_MFMA16

  float* slab = smem + wave * 64 * ES;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        slab[row * ES + j * 32 + (lane & 31)] = acc[i][j][r];
      }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr bool kBf16Out = (EPI == EPI_ROTARY_BF16 || EPI == EPI_SCALE_BF16);
  const int colbase = bn + wc * 64;
  if (kBf16Out && colbase >= a.vt_start) {
    // V panel: this wave's 64 columns are one head; emit V^T as bf16 [slot][head][d][npad].
    // lane = feature d; 8 consecutive tokens are packed into one 16-byte store.
    const int head = (colbase - a.vt_start) >> 6;
    const int row0 = bm + wr * 64;
    const int slot = row0 / a.npad, i0 = row0 - slot * a.npad;
    const float bias = a.bias ? a.bias[colbase + lane] : 0.f;
    uint16_t* dst = a.Vt + (((size_t)slot * kHeads + head) * kHeadDim + lane) * a.npad + i0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      unsigned int w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // vt_perm: within each 16-token group, 16-byte chunk hh holds tokens 4hh + {0..3, 8..11} -- the key order
        // one lane of the attention kernel's P^T operand carries, so its V^T fragment is one ds_read_b128
        const int t0 = a.vt_perm ? 16 * (c >> 1) + 4 * (c & 1) + ((2 * e) & 3) + 8 * ((2 * e) >> 2) : 8 * c + 2 * e;
        const float lo = slab[t0 * ES + lane] + bias;
        const float hi = slab[(t0 + 1) * ES + lane] + bias;
        w[e] = pack16_rt(lo, hi, a.half_fmt);
      }
      *reinterpret_cast<uint4*>(dst + 8 * c) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return;
  }
  const int c4 = (lane & 15) * 4;
  const int col = bn + wc * 64 + c4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (EPI != EPI_PLAIN && a.bias != nullptr) bias4 = *reinterpret_cast<const f32x4*>(a.bias + col);
  const bool do_scale = (EPI == EPI_SCALE_COLS || EPI == EPI_SCALE_BF16) && (col < a.scale_cols);
  const bool do_rot = (EPI == EPI_ROTARY || EPI == EPI_ROTARY_BF16) && (col < a.rot_cols);
  const int f0 = (col & 63) >> 1;
  // All 16 row fragments are pulled out of the slab into their OWN registers before the first store is
  // issued: a 16-byte global store followed by an LDS read that returns into the store's data registers can
  // corrupt the store when the memory pipeline is back-pressured (observed on gfx950: one float4 component
  // of a 16-lane group replaced by the next row's raw accumulator).
  // rotary tables / residual rows of the 16 rows this lane touches, requested back to back and outside any per-lane
  // branch (a 64-column sub-tile is entirely inside or outside the rotated range): one L2 latency, not one per row
  float2 cs16[16], sn16[16];
  f32x4 res16[16];
  if ((EPI == EPI_ROTARY || EPI == EPI_ROTARY_BF16) && (bn + wc * 64) < a.rot_cols) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const size_t row = (size_t)(bm + wr * 64 + it * 4 + (lane >> 4));
      cs16[it] = *reinterpret_cast<const float2*>(a.cos_t + row * kFreq + f0);
      sn16[it] = *reinterpret_cast<const float2*>(a.sin_t + row * kFreq + f0);
    }
  }
  if (EPI == EPI_RESIDUAL) {
#pragma unroll
    for (int it = 0; it < 16; ++it) res16[it] = *reinterpret_cast<const f32x4*>(a.resid + (size_t)(bm + wr * 64 + it * 4 + (lane >> 4)) * a.ldr + col);
  }
  f32x4 vals[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) vals[it] = *reinterpret_cast<const f32x4*>(&slab[(it * 4 + (lane >> 4)) * ES + c4]);
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int lr_ = it * 4 + (lane >> 4);
    const int row = bm + wr * 64 + lr_;
    f32x4 v = vals[it];
    v += bias4;
    if (EPI == EPI_SCALE_COLS || EPI == EPI_SCALE_BF16) {
      if (do_scale) v *= a.scale;
    } else if (EPI == EPI_ROTARY || EPI == EPI_ROTARY_BF16) {
      if (do_rot) {
        const float2 cs = cs16[it], sn = sn16[it];
        f32x4 o;
        o.x = v.x * cs.x + (-v.y) * sn.x;
        o.y = v.y * cs.x + v.x * sn.x;
        o.z = v.z * cs.y + (-v.w) * sn.y;
        o.w = v.w * cs.y + v.z * sn.y;
        v = o;
      }
    } else if (EPI == EPI_RESIDUAL) {
      v += res16[it];
    } else if (EPI == EPI_RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (kBf16Out) {
      if (col < a.q_cols) v *= a.qscale;
      uint2 pk;
      pk.x = pack16_rt(v.x, v.y, a.half_fmt);
      pk.y = pack16_rt(v.z, v.w, a.half_fmt);
      *reinterpret_cast<uint2*>(a.Yb + (size_t)row * a.ldyb + col) = pk;
    } else {
      *reinterpret_cast<f32x4*>(Y + (size_t)row * a.ldy + col) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The exact-f32 GEMM on 64 x 128 tiles (round 5) for grids that leave most of the chip idle on 128 x 128 tiles: LoFTR's coarse cross halves
// are M = 4864 rows x N = 256 / 512 columns = 76 / 152 workgroups of k_gemm_f32_v3 on 256 CUs, each a 17 us chain of MFMAs (29 us per launch for
// 0.6 GFLOP: profiles/r05e_loftr_layers_exact_f32.txt).  Same operands, swizzle, LDS-DMA staging and k order as k_gemm_f32_v3 -- every output
// element sums the same products in the same order: same bits --, half the rows per workgroup: wave w owns rows 32 (w >> 1) .. and columns
// 64 (w & 1) ..; plain / bias / ReLU (the coarse transformer) and column-scale / rotary / residual epilogues (the matcher in the exact-f32 mode at
// one or two pairs per call: 32-96 workgroups of k_gemm_f32_v3, 35 us against 19 here), f32 output only; compiler-scheduled.
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f32_m64(GemmArgs a) {
  constexpr int BM2 = 64;
  constexpr int TILE = (BM2 + BN) * BK;               // floats per buffer
  constexpr int SLAB = 4 * 32 * ES;
  __shared__ __attribute__((aligned(16))) float smem[(2 * TILE > SLAB) ? 2 * TILE : SLAB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int bm = blockIdx.y * BM2, bn = blockIdx.x * BN;
  if (a.mlim != nullptr && (a.mlim_seg > 0 ? bm % a.mlim_seg : bm) >= a.mlim[0] * a.mlim_mul) return;     // GemmArgs::mlim: a tile of rows nobody will read
  const float* A = a.A + (long long)blockIdx.z * a.strideA;
  const float* W = a.W + (long long)blockIdx.z * a.strideW;
  float* Y = a.Y + (long long)blockIdx.z * a.strideY;
  const float* const A2 = a.A2;
  const int lda = a.lda, lda2 = a.lda2, ldw = a.ldw, K1 = a.K1, K = a.K;
  // staging: wave w brings rows [16 w, 16 w + 16) of the A tile (2 instructions of 8 rows) and rows [32 w, 32 w + 32) of the B tile (4);
  // lane -> (row offset lane >> 3, chunk position lane & 7) fetches source chunk pos ^ f(row), f(row) = (row ^ (row >> 3)) & 7
  const int dpos = lane & 7;
  const float* asrc[2]; const float* a2src[2]; const float* wsrc[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wave * 16 + 8 * j + (lane >> 3);
    const int c = dpos ^ ((row ^ (row >> 3)) & 7);
    asrc[j] = A + (size_t)(bm + row) * lda + c * 4;
    a2src[j] = A2 ? A2 + (size_t)(bm + row) * lda2 + c * 4 - K1 : nullptr;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = wave * 32 + 8 * j + (lane >> 3);
    const int c = dpos ^ ((row ^ (row >> 3)) & 7);
    wsrc[j] = W + (size_t)(bn + row) * ldw + c * 4;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_tile = [&](int buf, int k0) __attribute__((always_inline)) {
    const bool second = (A2 != nullptr) && (k0 >= K1);
    float* la_ = smem + buf * TILE + (wave_u * 16) * BK;
    float* lb_ = smem + buf * TILE + BM2 * BK + (wave_u * 32) * BK;
#pragma unroll
    for (int j = 0; j < 2; ++j) __builtin_amdgcn_global_load_lds((gptr_t)((second ? a2src[j] : asrc[j]) + k0), (lptr_t)(la_ + j * 8 * BK), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + k0), (lptr_t)(lb_ + j * 8 * BK), 16, 0, 0);
  };
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int hh = lane >> 5;
  const int ra_ = wr * 32 + (lane & 31);
  const int arow = ra_ * BK, ga = hh ^ ((ra_ ^ (ra_ >> 3)) & 7);
  int brow[2], gb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rb_ = wc * 64 + 32 * j + (lane & 31);
    brow[j] = BM2 * BK + rb_ * BK; gb[j] = hh ^ ((rb_ ^ (rb_ >> 3)) & 7);
  }
  const int nt = K / BK;
  dma_tile(0, 0);
  { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) dma_tile(cur ^ 1, (t + 1) * BK);
    const float* b_ = smem + cur * TILE;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(b_ + arow + 4 * ((kc << 1) ^ ga));
      f32x4 fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b_ + brow[j] + 4 * ((kc << 1) ^ gb[j]));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[j].x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[j].y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[j].z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[j].w, acc[j], 0, 0, 0);
      }
    }
    { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  }
  // epilogue: the wave's 32 x 64 block through its LDS slab, read back row-wise (16 lanes x 16 bytes = one 256-byte row segment)
  float* slab = smem + wave * 32 * ES;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      slab[row * ES + j * 32 + (lane & 31)] = acc[j][r];
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int c4 = (lane & 15) * 4;
  const int col = bn + wc * 64 + c4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (EPI != EPI_PLAIN && a.bias != nullptr) bias4 = *reinterpret_cast<const f32x4*>(a.bias + col);
  // column scale / rotary / residual (the matcher's exact-f32 mode at one or two pairs per call): k_gemm_f32_v3's expressions, term for term
  const bool do_scale = EPI == EPI_SCALE_COLS && col < a.scale_cols;
  const bool do_rot = EPI == EPI_ROTARY && col < a.rot_cols;
  const int f0 = (col & 63) >> 1;
  float2 cs8[8], sn8[8];
  f32x4 res8[8];
  if (EPI == EPI_ROTARY && (bn + wc * 64) < a.rot_cols) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const size_t row = (size_t)(bm + wr * 32 + it * 4 + (lane >> 4));
      cs8[it] = *reinterpret_cast<const float2*>(a.cos_t + row * kFreq + f0);
      sn8[it] = *reinterpret_cast<const float2*>(a.sin_t + row * kFreq + f0);
    }
  }
  if (EPI == EPI_RESIDUAL) {
#pragma unroll
    for (int it = 0; it < 8; ++it) res8[it] = *reinterpret_cast<const f32x4*>(a.resid + (size_t)(bm + wr * 32 + it * 4 + (lane >> 4)) * a.ldr + col);
  }
  f32x4 vals[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) vals[it] = *reinterpret_cast<const f32x4*>(&slab[(it * 4 + (lane >> 4)) * ES + c4]);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = bm + wr * 32 + it * 4 + (lane >> 4);
    f32x4 v = vals[it];
    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
    if (EPI == EPI_SCALE_COLS) {
      if (do_scale) v *= a.scale;
    } else if (EPI == EPI_ROTARY) {
      if (do_rot) {
        const float2 cs = cs8[it], sn = sn8[it];
        f32x4 o;
        o.x = v.x * cs.x + (-v.y) * sn.x;
        o.y = v.y * cs.x + v.x * sn.x;
        o.z = v.z * cs.y + (-v.w) * sn.y;
        o.w = v.w * cs.y + v.z * sn.y;
        v = o;
      }
    } else if (EPI == EPI_RESIDUAL) {
      v += res8[it];
    }
    if (EPI == EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<f32x4*>(Y + (size_t)row * a.ldy + col) = v;
  }
}

// k_gemm_f32_r64 (late round 5): k_gemm_f32_m64 with a FOUR-slot LDS ring -- three k-tiles in flight behind a counted s_waitcnt instead of one
// (a 64-row workgroup computes a k-tile in ~1 k (BN2 = 64) / ~2 k cycles (128) and then waited a whole DMA latency for the next: the 74 GEMMs of a
// one-pair call in GN_PREC_F32 took 19.7 us each for 0.3-0.8 GFLOP) -- and, for grids that still fill at most half of the chip on 64 x 128 tiles,
// 64 x 64 tiles (BN2 = 64: wave w owns rows 32 (w >> 1) .., columns 32 (w & 1) ..).  Same operands, swizzle, k order and MFMA sequence per output
// element as k_gemm_f32_v3: same bits.  One workgroup barrier per k-tile.  Measured (tools/small_batch.py 1 --precision f32 --table --knob 44:k): the ring alone
// buys 19.9 -> 19.4 us per GEMM, the 64 x 64 tiles 19.9 -> 14.5 (twice the workgroups on a chip that was a quarter to a half full): shipped for those grids only.
template <int EPI, int BN2>
__global__ __launch_bounds__(256) void k_gemm_f32_r64(GemmArgs a) {
  constexpr int BM2 = 64, NJ = BN2 / 64, NB = BN2 / 32, RING = 4;
  constexpr int TILE = (BM2 + BN2) * BK;              // floats per ring slot
  constexpr int SLAB = 4 * 32 * ES;
  __shared__ __attribute__((aligned(16))) float smem[(RING * TILE > SLAB) ? RING * TILE : SLAB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int bm = blockIdx.y * BM2, bn = blockIdx.x * BN2;
  if (a.mlim != nullptr && (a.mlim_seg > 0 ? bm % a.mlim_seg : bm) >= a.mlim[0] * a.mlim_mul) return;     // GemmArgs::mlim: a tile of rows nobody will read
  const float* A = a.A + (long long)blockIdx.z * a.strideA;
  const float* W = a.W + (long long)blockIdx.z * a.strideW;
  float* Y = a.Y + (long long)blockIdx.z * a.strideY;
  const float* const A2 = a.A2;
  const int lda = a.lda, lda2 = a.lda2, ldw = a.ldw, K1 = a.K1, K = a.K;
  // staging: wave w brings rows [16 w, 16 w + 16) of the A tile (2 instructions of 8 rows) and rows [8 NB w, 8 NB w + 8 NB) of the B tile (NB);
  // lane -> (row offset lane >> 3, chunk position lane & 7) fetches source chunk pos ^ f(row), f(row) = (row ^ (row >> 3)) & 7.
  // The DMA is issued from assembly: the compiler neither sees nor waits for it (the builtin makes it drain every load in flight in front of
  // the next LDS read), so that the ring can keep three k-tiles in flight behind a counted s_waitcnt.
  const int dpos = lane & 7;
  const float* asrc[2]; const float* a2src[2]; const float* wsrc[NB];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wave * 16 + 8 * j + (lane >> 3);
    const int c = dpos ^ ((row ^ (row >> 3)) & 7);
    asrc[j] = A + (size_t)(bm + row) * lda + c * 4;
    a2src[j] = A2 ? A2 + (size_t)(bm + row) * lda2 + c * 4 - K1 : nullptr;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int row = wave * (8 * NB) + 8 * j + (lane >> 3);
    const int c = dpos ^ ((row ^ (row >> 3)) & 7);
    wsrc[j] = W + (size_t)(bn + row) * ldw + c * 4;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  auto dma1 = [&](const float* src, unsigned dst_floats) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds0 + dst_floats * 4u) : "memory");
  };
  auto dma_tile = [&](int buf, int k0) __attribute__((always_inline)) {
    const bool second = (A2 != nullptr) && (k0 >= K1);
    const unsigned la_ = (unsigned)(buf * TILE + (wave_u * 16) * BK);
    const unsigned lb_ = (unsigned)(buf * TILE + BM2 * BK + (wave_u * 8 * NB) * BK);
#pragma unroll
    for (int j = 0; j < 2; ++j) dma1((second ? a2src[j] : asrc[j]) + k0, la_ + j * 8 * BK);
#pragma unroll
    for (int j = 0; j < NB; ++j) dma1(wsrc[j] + k0, lb_ + j * 8 * BK);
  };
  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int hh = lane >> 5;
  const int ra_ = wr * 32 + (lane & 31);
  const int arow = ra_ * BK, ga = hh ^ ((ra_ ^ (ra_ >> 3)) & 7);
  int brow[NJ], gb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int rb_ = wc * (BN2 / 2) + 32 * j + (lane & 31);
    brow[j] = BM2 * BK + rb_ * BK; gb[j] = hh ^ ((rb_ ^ (rb_ >> 3)) & 7);
  }
  const int nt = K / BK;
  constexpr int PER = 2 + NB;                       // DMA instructions per wave and k-tile
#pragma unroll
  for (int q = 0; q < RING - 1; ++q)
    if (q < nt) dma_tile(q, q * BK);
  for (int t = 0; t < nt; ++t) {
    // tile t has landed once at most the (up to two) later tiles of this wave are still in flight; the barrier then says the same of every wave's
    // share -- and that every wave is done with tile t - 1, whose slot the next request overwrites
    const int later = nt - 1 - t;
    if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + RING - 1 < nt) dma_tile((t + RING - 1) % RING, (t + RING - 1) * BK);
    const float* b_ = smem + (t % RING) * TILE;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(b_ + arow + 4 * ((kc << 1) ^ ga));
      f32x4 fb[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b_ + brow[j] + 4 * ((kc << 1) ^ gb[j]));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[j].x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[j].y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[j].z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[j].w, acc[j], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                  // every wave is done with the last tile: the slabs alias the ring
  // epilogue: the wave's 32 x (BN2 / 2) block through its LDS slab, read back row-wise (16 (8) lanes x 16 bytes = one row segment)
  float* slab = smem + wave * 32 * ES;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      slab[row * ES + j * 32 + (lane & 31)] = acc[j][r];
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int CW = BN2 / 2, LPR = CW / 4, RPI = 64 / LPR, NIT = 32 / RPI;      // columns per wave, lanes per row segment, rows per pass, passes
  const int c4 = (lane % LPR) * 4;
  const int col = bn + wc * CW + c4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (EPI != EPI_PLAIN && a.bias != nullptr) bias4 = *reinterpret_cast<const f32x4*>(a.bias + col);
  // column scale / rotary / residual: k_gemm_f32_v3's expressions, term for term
  const bool do_scale = EPI == EPI_SCALE_COLS && col < a.scale_cols;
  const bool do_rot = EPI == EPI_ROTARY && col < a.rot_cols;
  const int f0 = (col & 63) >> 1;
  float2 cs8[NIT], sn8[NIT];
  f32x4 res8[NIT];
  if (EPI == EPI_ROTARY && (bn + wc * CW) < a.rot_cols) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const size_t row = (size_t)(bm + wr * 32 + it * RPI + lane / LPR);
      cs8[it] = *reinterpret_cast<const float2*>(a.cos_t + row * kFreq + f0);
      sn8[it] = *reinterpret_cast<const float2*>(a.sin_t + row * kFreq + f0);
    }
  }
  if (EPI == EPI_RESIDUAL) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) res8[it] = *reinterpret_cast<const f32x4*>(a.resid + (size_t)(bm + wr * 32 + it * RPI + lane / LPR) * a.ldr + col);
  }
  f32x4 vals[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) vals[it] = *reinterpret_cast<const f32x4*>(&slab[(it * RPI + lane / LPR) * ES + c4]);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = bm + wr * 32 + it * RPI + lane / LPR;
    f32x4 v = vals[it];
    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
    if (EPI == EPI_SCALE_COLS) {
      if (do_scale) v *= a.scale;
    } else if (EPI == EPI_ROTARY) {
      if (do_rot) {
        const float2 cs = cs8[it], sn = sn8[it];
        f32x4 o;
        o.x = v.x * cs.x + (-v.y) * sn.x;
        o.y = v.y * cs.x + v.x * sn.x;
        o.z = v.z * cs.y + (-v.w) * sn.y;
        o.w = v.w * cs.y + v.z * sn.y;
        v = o;
      }
    } else if (EPI == EPI_RESIDUAL) {
      v += res8[it];
    }
    if (EPI == EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<f32x4*>(Y + (size_t)row * a.ldy + col) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// f32x3: f32-accurate GEMM on the bf16 matrix pipe.  Every f32 operand is split exactly into three
// bf16 terms (x = xh + xm + xl, 8 + 8 + 8 mantissa bits) while its fragment sits in registers, and each
// 32x32x16 block is accumulated from the six products whose magnitude exceeds 2^-24 of the leading one
// (hh, hm, mh, hl, lh, mm) in f32.  Measured error vs fp64 is at or below that of the exact-f32 MFMA path
// (tests/test_gpu_parity.py) at 6 x 32 = 192 matrix-pipe cycles per 32x32x16 instead of 8 x 64 = 512.
// Tiles travel exactly as in variant 3 (f32 in HBM/LDS, LDS-DMA, XOR swizzle); the split is VALU work
// that overlaps the MFMAs of the previous k-step.
#ifndef GN_X3_LDS_PAD
#define GN_X3_LDS_PAD 0
#endif
// Exact 3-way split by TRUNCATION, integer ops only (no dependence on the SLP vectoriser): hi = top 16 bits
// of x, r = x - hi (exact), mid = top 16 bits of r, lo = top 16 bits of r - mid.  x has 24 significant bits,
// hi keeps 8, r at most 16, mid 8 of those, so lo (<= 8 bits) is exact too: x == hi + mid + lo.
// v_perm_b32 packs the upper halves of two dwords into one bf16 pair.
__device__ __forceinline__ void split2(float x0, float x1, unsigned int& h, unsigned int& m, unsigned int& l) {
  const unsigned int u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  const float r0 = x0 - __uint_as_float(u0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(u1 & 0xFFFF0000u);
  const unsigned int v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  const float s0 = r0 - __uint_as_float(v0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(v1 & 0xFFFF0000u);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& h, bf16x8& m, bf16x8& l) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  unsigned int h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
  split2(a.x, a.y, h0, m0, l0);
  split2(a.z, a.w, h1, m1, l1);
  split2(b.x, b.y, h2, m2, l2);
  split2(b.z, b.w, h3, m3, l3);
  const u32x4 hv = {h0, h1, h2, h3}, mv = {m0, m1, m2, m3}, lv = {l0, l1, l2, l3};
  h = __builtin_bit_cast(bf16x8, hv); m = __builtin_bit_cast(bf16x8, mv); l = __builtin_bit_cast(bf16x8, lv);
}

// WP = true: the B operand (weights) arrives PRE-SPLIT as three bf16 planes [3][N][K] (made once at load time),
// so only the A fragments are split in registers.  B plane tiles are [plane][128 rows][32 k] bf16 (64-byte
// rows); 16-byte chunk c of row r sits at position c ^ ((r >> 2) & 3), which makes every ds_read_b128 lane
// group hit 16 distinct bank slots.
template <int EPI, bool WP, int ABL = 0>   // ABL: timing-only ablations (1: no operand split, 2: no DMA in the loop, 4: no barrier)
__global__ __launch_bounds__(256) void k_gemm_f32x3(GemmArgs a) {
  constexpr int TILE = WP ? (BM * BK + 3 * BN * BK / 2) : (BM + BN) * BK;  // floats per buffer: f32 A tile + (f32 | 3 bf16-plane) B tile
  constexpr int SLAB = 4 * 64 * ES;
  __shared__ __attribute__((aligned(16))) float smem[((2 * TILE > SLAB) ? 2 * TILE : SLAB) + GN_X3_LDS_PAD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int bx, by;
  {
    const int gx = gridDim.x, nwg = gx * gridDim.y;
    const int L = blockIdx.y * gx + blockIdx.x;
    const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    bx = v % gx; by = v / gx;
  }
  const int bm = by * BM, bn = bx * BN;
  const float* A = a.A + (long long)blockIdx.z * a.strideA;
  const float* W = a.W + (long long)blockIdx.z * a.strideW;
  float* Y = a.Y + (long long)blockIdx.z * a.strideY;
  const float* const A2 = a.A2;
  const int lda = a.lda, lda2 = a.lda2, ldw = a.ldw, K1 = a.K1, K = a.K;

  // DMA source addressing: wave w stages tile rows [32w, 32w + 32) of A and of B, 8 rows per instruction.
  // lane -> (row offset lane >> 3, chunk position lane & 7) fetches source chunk pos ^ f(row).
  const int drow = wave * 32 + (lane >> 3);            // + 8 * j
  const int dpos = lane & 7;
  // f(row) for row = drow + 8j: (row ^ (row >> 3)) & 7; row >> 3 = 4 * wave + j
  const float* asrc[4]; const float* a2src[4]; const float* wsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = drow + 8 * j;
    const int c = dpos ^ ((row ^ (row >> 3)) & 7);
    asrc[j] = A + (size_t)(bm + row) * lda + c * 4;
    a2src[j] = A2 ? A2 + (size_t)(bm + row) * lda2 + c * 4 - K1 : nullptr;
    wsrc[j] = W + (size_t)(bn + row) * ldw + c * 4;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // pre-split weight planes: piece = 16 rows x 64 B; lane -> (row lane >> 2, position lane & 3) fetches chunk pos ^ g(row)
  const unsigned short* wpsrc[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int piece = wave_u * 6 + q, plane = piece >> 3, row = (piece & 7) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    wpsrc[q] = WP ? a.Wp + (size_t)plane * a.wp_plane + (size_t)(bn + row) * ldw + c * 8 : nullptr;
  }
#define GN_DMA_TILE(buf, k0)                                                                      \
  {                                                                                               \
    const bool second = (A2 != nullptr) && ((k0) >= K1);                                          \
    float* la_ = smem + (buf) * TILE + (wave_u * 32) * BK;                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
      const float* ga_ = (second ? a2src[j] : asrc[j]) + (k0);                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)ga_, (lptr_t)(la_ + j * 8 * BK), 16, 0, 0);       \
    }                                                                                             \
    if (WP) {                                                                                     \
      unsigned short* lb_ = reinterpret_cast<unsigned short*>(smem + (buf) * TILE + BM * BK);    \
      _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                             \
        const int piece = wave_u * 6 + q;                    /* 24 pieces: plane = piece / 8 */   \
        __builtin_amdgcn_global_load_lds((gptr_t)(wpsrc[q] + (k0)), (lptr_t)(lb_ + piece * 512), 16, 0, 0); \
      }                                                                                           \
    } else {                                                                                      \
      float* lb_ = smem + (buf) * TILE + BM * BK + (wave_u * 32) * BK;                            \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                               \
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + (k0)), (lptr_t)(lb_ + j * 8 * BK), 16, 0, 0); \
    }                                                                                             \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addressing (floats): row * 32 + 4 * (chunk ^ f(row)), chunk = 2 kc + hh
  const int hh = lane >> 5;
  int arow_[2], brow_[2], ga_[2], gb_[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra_ = wr * 64 + 32 * i + (lane & 31), rb_ = wc * 64 + 32 * i + (lane & 31);
    arow_[i] = ra_ * BK; brow_[i] = BM * BK + rb_ * BK;
    ga_[i] = hh ^ ((ra_ ^ (ra_ >> 3)) & 7); gb_[i] = hh ^ ((rb_ ^ (rb_ >> 3)) & 7);
  }
  const int nt = K / BK;

// raw f32 fragments of k-step s (16 k values): lane (row, hh) holds k = 16 s + 8 hh + 0..7 = two 16-byte chunks
#define GN_RAW_READ(ra_, rb_, buf, s_)                                                            \
  {                                                                                               \
    const float* b_ = smem + (buf) * TILE;                                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                               \
      ra_[i][0] = *reinterpret_cast<const f32x4*>(b_ + arow_[i] + 4 * ((4 * (s_) + 0) ^ ga2_[i])); \
      ra_[i][1] = *reinterpret_cast<const f32x4*>(b_ + arow_[i] + 4 * ((4 * (s_) + 1) ^ ga2_[i])); \
      if (!WP) {                                                                                  \
        rb_[i][0] = *reinterpret_cast<const f32x4*>(b_ + brow_[i] + 4 * ((4 * (s_) + 0) ^ gb2_[i])); \
        rb_[i][1] = *reinterpret_cast<const f32x4*>(b_ + brow_[i] + 4 * ((4 * (s_) + 1) ^ gb2_[i])); \
      }                                                                                           \
    }                                                                                             \
  }
#define GN_SPLIT(ra_, rb_, A_, B_)                                                                \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                               \
      if (ABL & 1) { A_[i][0] = __builtin_bit_cast(bf16x8, ra_[i][0]); A_[i][1] = __builtin_bit_cast(bf16x8, ra_[i][1]); A_[i][2] = A_[i][0]; } \
      else split8(ra_[i][0], ra_[i][1], A_[i][0], A_[i][1], A_[i][2]);                            \
      if (!WP) split8(rb_[i][0], rb_[i][1], B_[i][0], B_[i][1], B_[i][2]);                        \
    }                                                                                             \
  }
// pre-split B fragments straight from the plane tiles (no VALU): lane (col, hh), k-step s -> chunk 2 s + hh
#define GN_BPLANE_READ(B_, buf, s_)                                                               \
  {                                                                                               \
    const unsigned short* pb_ = reinterpret_cast<const unsigned short*>(smem + (buf) * TILE + BM * BK); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                            \
        B_[i][pl] = *reinterpret_cast<const bf16x8*>(pb_ + pl * (BN * BK) + bprow_[i] + 8 * ((2 * (s_)) ^ gbp_[i])); \
  }
// x y = (xh + xm + xl)(yh + ym + yl) ~ xh yh + xh ym + xm yh + xh yl + xl yh + xm ym, smallest terms first
// x y = (xh + xm + xl)(yh + ym + yl) ~ xh yh + xh ym + xm yh + xh yl + xl yh + xm ym (smallest terms first).
// The four accumulators are visited round-robin inside every product so that consecutive MFMAs are independent.
#define GN_MFMA_P(pa, pb, A_, B_)                                                                  \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0][pa], B_[0][pb], acc[0][0], 0, 0, 0);  \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0][pa], B_[1][pb], acc[0][1], 0, 0, 0);  \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[1][pa], B_[0][pb], acc[1][0], 0, 0, 0);  \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[1][pa], B_[1][pb], acc[1][1], 0, 0, 0);
#define GN_MFMA24(A_, B_) { GN_MFMA_P(1, 1, A_, B_) GN_MFMA_P(2, 0, A_, B_) GN_MFMA_P(0, 2, A_, B_) GN_MFMA_P(1, 0, A_, B_) GN_MFMA_P(0, 1, A_, B_) GN_MFMA_P(0, 0, A_, B_) }

  // chunk index for k-step s and half hh is 4 s + 2 hh + {0, 1}: fold hh and f(row) into one xor mask
  int ga2_[2], gb2_[2], bprow_[2], gbp_[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ga2_[i] = (2 * hh) ^ (ga_[i] ^ hh); gb2_[i] = (2 * hh) ^ (gb_[i] ^ hh);
    const int rb_ = wc * 64 + 32 * i + (lane & 31);
    bprow_[i] = rb_ * BK;                       // halves: 32 bf16 per row
    gbp_[i] = hh ^ ((rb_ >> 2) & 3);            // chunk 2 s + hh at position (2 s) ^ hh ^ g(row)
  }

  GN_DMA_TILE(0, 0);
  { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  if (nt > 1) GN_DMA_TILE(1, BK);
  f32x4 rawa[2][2], rawb[2][2];
  bf16x8 A0[2][3], B0[2][3], A1[2][3], B1[2][3];
  GN_RAW_READ(rawa, rawb, 0, 0);
  if (WP) GN_BPLANE_READ(B0, 0, 0);
  GN_SPLIT(rawa, rawb, A0, B0);

  // One barrier per k-tile.  Its release drains the DMA of tile t+1 (issued a FULL k-tile earlier, right after
  // the previous barrier); at that point every wave has also finished reading buffer `cur`, so the DMA of
  // tile t+2 into `cur` is issued immediately -- the longest prefetch distance two LDS buffers allow.
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    GN_RAW_READ(rawa, rawb, cur, 1);
    if (WP) GN_BPLANE_READ(B1, cur, 1);
    GN_MFMA24(A0, B0)
    GN_SPLIT(rawa, rawb, A1, B1);
    if (!(ABL & 4)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    if (!(ABL & 2) && t + 2 < nt) GN_DMA_TILE(cur, (t + 2) * BK);
    if (t + 1 < nt) {
      GN_RAW_READ(rawa, rawb, cur ^ 1, 0);
      if (WP) GN_BPLANE_READ(B0, cur ^ 1, 0);
    }
    GN_MFMA24(A1, B1)
    if (t + 1 < nt) GN_SPLIT(rawa, rawb, A0, B0);
  }
#undef GN_DMA_TILE
#undef GN_RAW_READ
#undef GN_SPLIT
#undef GN_BPLANE_READ
#undef GN_MFMA_P
#undef GN_MFMA24

  float* slab = smem + wave * 64 * ES;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        slab[row * ES + j * 32 + (lane & 31)] = acc[i][j][r];
      }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr bool kBf16Out = (EPI == EPI_ROTARY_BF16 || EPI == EPI_SCALE_BF16);
  const int colbase = bn + wc * 64;
  if (kBf16Out && colbase >= a.vt_start) {
    // V panel: this wave's 64 columns are one head; emit V^T as bf16 [slot][head][d][npad].
    // lane = feature d; 8 consecutive tokens are packed into one 16-byte store.
    const int head = (colbase - a.vt_start) >> 6;
    const int row0 = bm + wr * 64;
    const int slot = row0 / a.npad, i0 = row0 - slot * a.npad;
    const float bias = a.bias ? a.bias[colbase + lane] : 0.f;
    uint16_t* dst = a.Vt + (((size_t)slot * kHeads + head) * kHeadDim + lane) * a.npad + i0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      unsigned int w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // vt_perm: within each 16-token group, 16-byte chunk hh holds tokens 4hh + {0..3, 8..11} -- the key order
        // one lane of the attention kernel's P^T operand carries, so its V^T fragment is one ds_read_b128
        const int t0 = a.vt_perm ? 16 * (c >> 1) + 4 * (c & 1) + ((2 * e) & 3) + 8 * ((2 * e) >> 2) : 8 * c + 2 * e;
        const float lo = slab[t0 * ES + lane] + bias;
        const float hi = slab[(t0 + 1) * ES + lane] + bias;
        w[e] = pack16_rt(lo, hi, a.half_fmt);
      }
      *reinterpret_cast<uint4*>(dst + 8 * c) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return;
  }
  const int c4 = (lane & 15) * 4;
  const int col = bn + wc * 64 + c4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (EPI != EPI_PLAIN && a.bias != nullptr) bias4 = *reinterpret_cast<const f32x4*>(a.bias + col);
  const bool do_scale = (EPI == EPI_SCALE_COLS || EPI == EPI_SCALE_BF16) && (col < a.scale_cols);
  const bool do_rot = (EPI == EPI_ROTARY || EPI == EPI_ROTARY_BF16) && (col < a.rot_cols);
  const int f0 = (col & 63) >> 1;
  // All 16 row fragments are pulled out of the slab into their OWN registers before the first store is
  // issued: a 16-byte global store followed by an LDS read that returns into the store's data registers can
  // corrupt the store when the memory pipeline is back-pressured (observed on gfx950: one float4 component
  // of a 16-lane group replaced by the next row's raw accumulator).
  // rotary tables / residual rows of the 16 rows this lane touches, requested back to back and outside any per-lane
  // branch (a 64-column sub-tile is entirely inside or outside the rotated range): one L2 latency, not one per row
  float2 cs16[16], sn16[16];
  f32x4 res16[16];
  if ((EPI == EPI_ROTARY || EPI == EPI_ROTARY_BF16) && (bn + wc * 64) < a.rot_cols) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const size_t row = (size_t)(bm + wr * 64 + it * 4 + (lane >> 4));
      cs16[it] = *reinterpret_cast<const float2*>(a.cos_t + row * kFreq + f0);
      sn16[it] = *reinterpret_cast<const float2*>(a.sin_t + row * kFreq + f0);
    }
  }
  if (EPI == EPI_RESIDUAL) {
#pragma unroll
    for (int it = 0; it < 16; ++it) res16[it] = *reinterpret_cast<const f32x4*>(a.resid + (size_t)(bm + wr * 64 + it * 4 + (lane >> 4)) * a.ldr + col);
  }
  f32x4 vals[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) vals[it] = *reinterpret_cast<const f32x4*>(&slab[(it * 4 + (lane >> 4)) * ES + c4]);
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int lr_ = it * 4 + (lane >> 4);
    const int row = bm + wr * 64 + lr_;
    f32x4 v = vals[it];
    v += bias4;
    if (EPI == EPI_SCALE_COLS || EPI == EPI_SCALE_BF16) {
      if (do_scale) v *= a.scale;
    } else if (EPI == EPI_ROTARY || EPI == EPI_ROTARY_BF16) {
      if (do_rot) {
        const float2 cs = cs16[it], sn = sn16[it];
        f32x4 o;
        o.x = v.x * cs.x + (-v.y) * sn.x;
        o.y = v.y * cs.x + v.x * sn.x;
        o.z = v.z * cs.y + (-v.w) * sn.y;
        o.w = v.w * cs.y + v.z * sn.y;
        v = o;
      }
    } else if (EPI == EPI_RESIDUAL) {
      v += res16[it];
    } else if (EPI == EPI_RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (kBf16Out) {
      if (col < a.q_cols) v *= a.qscale;
      uint2 pk;
      pk.x = pack16_rt(v.x, v.y, a.half_fmt);
      pk.y = pack16_rt(v.z, v.w, a.half_fmt);
      vals[it].x = __uint_as_float(pk.x); vals[it].y = __uint_as_float(pk.y);
    } else {
      vals[it] = v;
    }
  }
  // stores go last, from registers nothing writes any more
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = bm + wr * 64 + it * 4 + (lane >> 4);
    if (kBf16Out) {
      uint2 pk; pk.x = __float_as_uint(vals[it].x); pk.y = __float_as_uint(vals[it].y);
      *reinterpret_cast<uint2*>(a.Yb + (size_t)row * a.ldyb + col) = pk;
    } else {
      *reinterpret_cast<f32x4*>(Y + (size_t)row * a.ldy + col) = vals[it];
    }
  }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
// two-term fp16 split with round-to-nearest: h = fp16(x), m = fp16(x - h)   (v_cvt_pk_f16_f32, v_cvt_f32_f16, v_sub)
__device__ __forceinline__ void split2h(float x0, float x1, unsigned int& h, unsigned int& m) {
  f32x2v x = {x0, x1};
  const f16x2v hv = __builtin_convertvector(x, f16x2v);
  f32x2v r = {x0 - (float)hv[0], x1 - (float)hv[1]};
  const f16x2v mv = __builtin_convertvector(r, f16x2v);
  h = __builtin_bit_cast(unsigned int, hv);
  m = __builtin_bit_cast(unsigned int, mv);
}
__device__ __forceinline__ void split8h(const f32x4& a, const f32x4& b, f16x8& h, f16x8& m) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  unsigned int h0, h1, h2, h3, m0, m1, m2, m3;
  split2h(a.x, a.y, h0, m0);
  split2h(a.z, a.w, h1, m1);
  split2h(b.x, b.y, h2, m2);
  split2h(b.z, b.w, h3, m3);
  const u32x4 hv = {h0, h1, h2, h3}, mv = {m0, m1, m2, m3};
  h = __builtin_bit_cast(f16x8, hv); m = __builtin_bit_cast(f16x8, mv);
}

// ------------------------------------------------------------------------------------------------
// f16x2: f32-class GEMM on the fp16 matrix pipe at HALF the MFMA work of f32x3.  Every f32 operand is split into
// two fp16 terms with round-to-nearest (x = xh + xm + e, |e| <= max(2^-22 |x|, 2^-25): the fp16 matrix pipe of
// gfx950 honours subnormal inputs, checked by tools/probes/f16_denorm.hip) and each 32x32x16 block is
// accumulated in f32 from three products (xm yh, xh ym, xh yh); the dropped terms are <= 3 * 2^-22 |x y|,
// below the rounding noise of an f32 accumulation over K >= 128.  Weights arrive PRE-SPLIT as two fp16 planes
// [2][N][K], scaled by a power of two so that max |w| sits near 2^13 (keeps both planes far from the
// subnormal range; the epilogue multiplies the accumulator by the exact inverse, GemmArgs::acc_scale).
// Domain: |activation| < 65504 (fp16 range) -- LightGlue's activations are O(1..100).
// Tiles, DMA, swizzles and epilogues are those of k_gemm_f32x3.
template <int EPI, bool WP, int ABL = 0>   // ABL: timing-only ablations (1: no operand split, 2: no DMA in the loop, 4: no barrier)
__global__ __launch_bounds__(256) void k_gemm_f16x2(GemmArgs a) {
  constexpr int TILE = (BM + BN) * BK;  // floats per buffer: f32 A tile + (f32 | 2 fp16-plane) B tile, 16 KB each
  constexpr int SLAB = 4 * 64 * ES;
  __shared__ __attribute__((aligned(16))) float smem[((2 * TILE > SLAB) ? 2 * TILE : SLAB) + GN_X3_LDS_PAD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int bx, by;
  {
    const int gx = gridDim.x, nwg = gx * gridDim.y;
    const int L = blockIdx.y * gx + blockIdx.x;
    const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    bx = v % gx; by = v / gx;
  }
  const int bm = by * BM, bn = bx * BN;
  if (a.mlim != nullptr && (a.mlim_seg > 0 ? bm % a.mlim_seg : bm) >= a.mlim[0] * a.mlim_mul) return;     // GemmArgs::mlim: a tile of rows nobody will read
  const float* A = a.A + (long long)blockIdx.z * a.strideA;
  const float* W = a.W + (long long)blockIdx.z * a.strideW;
  float* Y = a.Y + (long long)blockIdx.z * a.strideY;
  const float* const A2 = a.A2;
  const int lda = a.lda, lda2 = a.lda2, ldw = a.ldw, K1 = a.K1, K = a.K;

  // DMA source addressing: wave w stages tile rows [32w, 32w + 32) of A and of B, 8 rows per instruction.
  // lane -> (row offset lane >> 3, chunk position lane & 7) fetches source chunk pos ^ f(row).
  const int drow = wave * 32 + (lane >> 3);            // + 8 * j
  const int dpos = lane & 7;
  // f(row) for row = drow + 8j: (row ^ (row >> 3)) & 7; row >> 3 = 4 * wave + j
  const float* asrc[4]; const float* a2src[4]; const float* wsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = drow + 8 * j;
    const int c = dpos ^ ((row ^ (row >> 3)) & 7);
    asrc[j] = A + (size_t)(bm + row) * lda + c * 4;
    a2src[j] = A2 ? A2 + (size_t)(bm + row) * lda2 + c * 4 - K1 : nullptr;
    wsrc[j] = W + (size_t)(bn + row) * ldw + c * 4;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // pre-split weight planes: piece = 16 rows x 64 B; lane -> (row lane >> 2, position lane & 3) fetches chunk pos ^ g(row)
  const unsigned short* wpsrc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int piece = wave_u * 4 + q, plane = piece >> 3, row = (piece & 7) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    wpsrc[q] = WP ? a.Wp + (size_t)plane * a.wp_plane + (size_t)(bn + row) * ldw + c * 8 : nullptr;
  }
#define GN_DMA_TILE(buf, k0)                                                                      \
  {                                                                                               \
    const bool second = (A2 != nullptr) && ((k0) >= K1);                                          \
    float* la_ = smem + (buf) * TILE + (wave_u * 32) * BK;                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
      const float* ga_ = (second ? a2src[j] : asrc[j]) + (k0);                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)ga_, (lptr_t)(la_ + j * 8 * BK), 16, 0, 0);       \
    }                                                                                             \
    if (WP) {                                                                                     \
      unsigned short* lb_ = reinterpret_cast<unsigned short*>(smem + (buf) * TILE + BM * BK);    \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
        const int piece = wave_u * 4 + q;                    /* 16 pieces: plane = piece / 8 */   \
        __builtin_amdgcn_global_load_lds((gptr_t)(wpsrc[q] + (k0)), (lptr_t)(lb_ + piece * 512), 16, 0, 0); \
      }                                                                                           \
    } else {                                                                                      \
      float* lb_ = smem + (buf) * TILE + BM * BK + (wave_u * 32) * BK;                            \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                               \
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + (k0)), (lptr_t)(lb_ + j * 8 * BK), 16, 0, 0); \
    }                                                                                             \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addressing (floats): row * 32 + 4 * (chunk ^ f(row)), chunk = 2 kc + hh
  const int hh = lane >> 5;
  int arow_[2], brow_[2], ga_[2], gb_[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra_ = wr * 64 + 32 * i + (lane & 31), rb_ = wc * 64 + 32 * i + (lane & 31);
    arow_[i] = ra_ * BK; brow_[i] = BM * BK + rb_ * BK;
    ga_[i] = hh ^ ((ra_ ^ (ra_ >> 3)) & 7); gb_[i] = hh ^ ((rb_ ^ (rb_ >> 3)) & 7);
  }
  const int nt = K / BK;

// raw f32 fragments of k-step s (16 k values): lane (row, hh) holds k = 16 s + 8 hh + 0..7 = two 16-byte chunks
#define GN_RAW_READ(ra_, rb_, buf, s_)                                                            \
  {                                                                                               \
    const float* b_ = smem + (buf) * TILE;                                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                               \
      ra_[i][0] = *reinterpret_cast<const f32x4*>(b_ + arow_[i] + 4 * ((4 * (s_) + 0) ^ ga2_[i])); \
      ra_[i][1] = *reinterpret_cast<const f32x4*>(b_ + arow_[i] + 4 * ((4 * (s_) + 1) ^ ga2_[i])); \
      if (!WP) {                                                                                  \
        rb_[i][0] = *reinterpret_cast<const f32x4*>(b_ + brow_[i] + 4 * ((4 * (s_) + 0) ^ gb2_[i])); \
        rb_[i][1] = *reinterpret_cast<const f32x4*>(b_ + brow_[i] + 4 * ((4 * (s_) + 1) ^ gb2_[i])); \
      }                                                                                           \
    }                                                                                             \
  }
#define GN_SPLIT(ra_, rb_, A_, B_)                                                                \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                               \
      if (ABL & 1) { A_[i][0] = __builtin_bit_cast(f16x8, ra_[i][0]); A_[i][1] = __builtin_bit_cast(f16x8, ra_[i][1]); } \
      else split8h(ra_[i][0], ra_[i][1], A_[i][0], A_[i][1]);                                     \
      if (!WP) split8h(rb_[i][0], rb_[i][1], B_[i][0], B_[i][1]);                                 \
    }                                                                                             \
  }
// pre-split B fragments straight from the plane tiles (no VALU): lane (col, hh), k-step s -> chunk 2 s + hh
#define GN_BPLANE_READ(B_, buf, s_)                                                               \
  {                                                                                               \
    const unsigned short* pb_ = reinterpret_cast<const unsigned short*>(smem + (buf) * TILE + BM * BK); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
      _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)                                            \
        B_[i][pl] = *reinterpret_cast<const f16x8*>(pb_ + pl * (BN * BK) + bprow_[i] + 8 * ((2 * (s_)) ^ gbp_[i])); \
  }
// x y = (xh + xm)(yh + ym) ~ xm yh + xh ym + xh yh (small terms first).
// The four accumulators are visited round-robin inside every product so that consecutive MFMAs are independent.
#define GN_MFMA_P(pa, pb, A_, B_)                                                                  \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[0][pa], B_[0][pb], acc[0][0], 0, 0, 0);  \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[0][pa], B_[1][pb], acc[0][1], 0, 0, 0);  \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[1][pa], B_[0][pb], acc[1][0], 0, 0, 0);  \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[1][pa], B_[1][pb], acc[1][1], 0, 0, 0);
#define GN_MFMA24(A_, B_) { GN_MFMA_P(1, 0, A_, B_) GN_MFMA_P(0, 1, A_, B_) GN_MFMA_P(0, 0, A_, B_) }

  // chunk index for k-step s and half hh is 4 s + 2 hh + {0, 1}: fold hh and f(row) into one xor mask
  int ga2_[2], gb2_[2], bprow_[2], gbp_[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ga2_[i] = (2 * hh) ^ (ga_[i] ^ hh); gb2_[i] = (2 * hh) ^ (gb_[i] ^ hh);
    const int rb_ = wc * 64 + 32 * i + (lane & 31);
    bprow_[i] = rb_ * BK;                       // halves: 32 bf16 per row
    gbp_[i] = hh ^ ((rb_ >> 2) & 3);            // chunk 2 s + hh at position (2 s) ^ hh ^ g(row)
  }

  GN_DMA_TILE(0, 0);
  { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  if (nt > 1) GN_DMA_TILE(1, BK);
  f32x4 rawa[2][2], rawb[2][2];
  f16x8 A0[2][2], B0[2][2], A1[2][2], B1[2][2];
  GN_RAW_READ(rawa, rawb, 0, 0);
  if (WP) GN_BPLANE_READ(B0, 0, 0);
  GN_SPLIT(rawa, rawb, A0, B0);

  // One barrier per k-tile.  Its release drains the DMA of tile t+1 (issued a FULL k-tile earlier, right after
  // the previous barrier); at that point every wave has also finished reading buffer `cur`, so the DMA of
  // tile t+2 into `cur` is issued immediately -- the longest prefetch distance two LDS buffers allow.
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    GN_RAW_READ(rawa, rawb, cur, 1);
    if (WP) GN_BPLANE_READ(B1, cur, 1);
    GN_MFMA24(A0, B0)
    GN_SPLIT(rawa, rawb, A1, B1);
    if (!(ABL & 4)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    if (!(ABL & 2) && t + 2 < nt) GN_DMA_TILE(cur, (t + 2) * BK);
    if (t + 1 < nt) {
      GN_RAW_READ(rawa, rawb, cur ^ 1, 0);
      if (WP) GN_BPLANE_READ(B0, cur ^ 1, 0);
    }
    GN_MFMA24(A1, B1)
    if (t + 1 < nt) GN_SPLIT(rawa, rawb, A0, B0);
  }
#undef GN_DMA_TILE
#undef GN_RAW_READ
#undef GN_SPLIT
#undef GN_BPLANE_READ
#undef GN_MFMA_P
#undef GN_MFMA24

  float* slab = smem + wave * 64 * ES;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        slab[row * ES + j * 32 + (lane & 31)] = acc[i][j][r];
      }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr bool kBf16Out = (EPI == EPI_ROTARY_BF16 || EPI == EPI_SCALE_BF16);
  const int colbase = bn + wc * 64;
  if (kBf16Out && colbase >= a.vt_start) {
    // V panel: this wave's 64 columns are one head; emit V^T as bf16 [slot][head][d][npad].
    // lane = feature d; 8 consecutive tokens are packed into one 16-byte store.
    const int head = (colbase - a.vt_start) >> 6;
    const int row0 = bm + wr * 64;
    const int slot = row0 / a.npad, i0 = row0 - slot * a.npad;
    const float bias = a.bias ? a.bias[colbase + lane] : 0.f;
    uint16_t* dst = a.Vt + (((size_t)slot * kHeads + head) * kHeadDim + lane) * a.npad + i0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      unsigned int w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // vt_perm: within each 16-token group, 16-byte chunk hh holds tokens 4hh + {0..3, 8..11} -- the key order
        // one lane of the attention kernel's P^T operand carries, so its V^T fragment is one ds_read_b128
        const int t0 = a.vt_perm ? 16 * (c >> 1) + 4 * (c & 1) + ((2 * e) & 3) + 8 * ((2 * e) >> 2) : 8 * c + 2 * e;
        const float lo = slab[t0 * ES + lane] * a.acc_scale + bias;
        const float hi = slab[(t0 + 1) * ES + lane] * a.acc_scale + bias;
        w[e] = pack16_rt(lo, hi, a.half_fmt);
      }
      *reinterpret_cast<uint4*>(dst + 8 * c) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return;
  }
  const int c4 = (lane & 15) * 4;
  const int col = bn + wc * 64 + c4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (EPI != EPI_PLAIN && a.bias != nullptr) bias4 = *reinterpret_cast<const f32x4*>(a.bias + col);
  const bool do_scale = (EPI == EPI_SCALE_COLS || EPI == EPI_SCALE_BF16) && (col < a.scale_cols);
  const bool do_rot = (EPI == EPI_ROTARY || EPI == EPI_ROTARY_BF16) && (col < a.rot_cols);
  const int f0 = (col & 63) >> 1;
  // All 16 row fragments are pulled out of the slab into their OWN registers before the first store is
  // issued: a 16-byte global store followed by an LDS read that returns into the store's data registers can
  // corrupt the store when the memory pipeline is back-pressured (observed on gfx950: one float4 component
  // of a 16-lane group replaced by the next row's raw accumulator).
  // rotary tables / residual rows of the 16 rows this lane touches, requested back to back and outside any per-lane
  // branch (a 64-column sub-tile is entirely inside or outside the rotated range): one L2 latency, not one per row
  float2 cs16[16], sn16[16];
  f32x4 res16[16];
  if ((EPI == EPI_ROTARY || EPI == EPI_ROTARY_BF16) && (bn + wc * 64) < a.rot_cols) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const size_t row = (size_t)(bm + wr * 64 + it * 4 + (lane >> 4));
      cs16[it] = *reinterpret_cast<const float2*>(a.cos_t + row * kFreq + f0);
      sn16[it] = *reinterpret_cast<const float2*>(a.sin_t + row * kFreq + f0);
    }
  }
  if (EPI == EPI_RESIDUAL) {
#pragma unroll
    for (int it = 0; it < 16; ++it) res16[it] = *reinterpret_cast<const f32x4*>(a.resid + (size_t)(bm + wr * 64 + it * 4 + (lane >> 4)) * a.ldr + col);
  }
  f32x4 vals[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) vals[it] = *reinterpret_cast<const f32x4*>(&slab[(it * 4 + (lane >> 4)) * ES + c4]);
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int lr_ = it * 4 + (lane >> 4);
    const int row = bm + wr * 64 + lr_;
    f32x4 v = vals[it] * a.acc_scale;
    v += bias4;
    if (EPI == EPI_SCALE_COLS || EPI == EPI_SCALE_BF16) {
      if (do_scale) v *= a.scale;
    } else if (EPI == EPI_ROTARY || EPI == EPI_ROTARY_BF16) {
      if (do_rot) {
        const float2 cs = cs16[it], sn = sn16[it];
        f32x4 o;
        o.x = v.x * cs.x + (-v.y) * sn.x;
        o.y = v.y * cs.x + v.x * sn.x;
        o.z = v.z * cs.y + (-v.w) * sn.y;
        o.w = v.w * cs.y + v.z * sn.y;
        v = o;
      }
    } else if (EPI == EPI_RESIDUAL) {
      v += res16[it];
    } else if (EPI == EPI_RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (kBf16Out) {
      if (col < a.q_cols) v *= a.qscale;
      uint2 pk;
      pk.x = pack16_rt(v.x, v.y, a.half_fmt);
      pk.y = pack16_rt(v.z, v.w, a.half_fmt);
      vals[it].x = __uint_as_float(pk.x); vals[it].y = __uint_as_float(pk.y);
    } else {
      vals[it] = v;
    }
  }
  // stores go last, from registers nothing writes any more
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = bm + wr * 64 + it * 4 + (lane >> 4);
    if (kBf16Out) {
      uint2 pk; pk.x = __float_as_uint(vals[it].x); pk.y = __float_as_uint(vals[it].y);
      *reinterpret_cast<uint2*>(a.Yb + (size_t)row * a.ldyb + col) = pk;
    } else {
      *reinterpret_cast<f32x4*>(Y + (size_t)row * a.ldy + col) = vals[it];
    }
  }
}
}  // namespace

thread_local int g_gemm_variant = 3;
int g_lf_conv_knob = 0;  // developer knob 42: bit 0 = LoFTR's convolutions stage their halo tiles the way rounds 3-4 did (no register prefetch); bits 8.. = the overhead term of lf_conv's cost model x 100
int g_gemm_r64 = 1;     // developer knob 44 (launch_gemm_f32)
int g_gemm_m64 = 320;    // developer knob 41: the exact-f32 GEMM runs on 64-row tiles when its 128 x 128 grid has at most this many workgroups (0 = never)

// Pure-MFMA ceiling probe: 4 waves per CU-resident block, 8 independent accumulators, no memory traffic.
__global__ __launch_bounds__(256) void k_mfma_probe(float* out, int iters) {
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // iters > 0: constant operands (lowest switching activity); iters < 0: pseudo-random full-range
  // operands that differ per lane and per MFMA (what a real GEMM feeds the pipe)
  const bool rnd = iters < 0;
  const int n = rnd ? -iters : iters;
  float av[8], bv[8];
  unsigned int h = (threadIdx.x + 1u) * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u;
    av[i] = rnd ? ((int)(h >> 8) & 0xffff) * (1.0f / 32768.0f) - 1.0f : threadIdx.x * 1e-3f;
    h = h * 1664525u + 1013904223u;
    bv[i] = rnd ? ((int)(h >> 8) & 0xffff) * (1.0f / 32768.0f) - 1.0f : 1.0f + blockIdx.x * 1e-6f;
  }
  for (int it = 0; it < n; it += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[(i + j) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

// LDS-DMA addressing probe: every block fills 80 KB of LDS with 1 KB DMA pieces from a pattern buffer and
// verifies each piece with plain ds_reads; mismatching words are counted per piece in out[0..79], and
// out[80] counts blocks that ran.  spin > 0 keeps blocks resident longer so two blocks share a CU.
__global__ __launch_bounds__(256) void k_lds_dma_probe(const float* pattern, unsigned int* out, int spin) {
  __shared__ __attribute__((aligned(16))) float smem[20480];   // 80 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int rep = 0; rep < (spin > 0 ? spin : 1); ++rep) {
    for (int q = 0; q < 20; ++q) {
      const int piece = wave * 20 + q;                       // 80 pieces of 1 KB
      const float* src = pattern + ((size_t)(blockIdx.x & 7) * 80 + piece) * 256 + lane * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + piece * 256), 16, 0, 0);
    }
    __syncthreads();
    for (int piece = 0; piece < 80; ++piece) {
      const float got = smem[piece * 256 + tid];
      const float want = pattern[((size_t)(blockIdx.x & 7) * 80 + piece) * 256 + tid];
      if (got != want) atomicAdd(&out[piece], 1u);
    }
    __syncthreads();
    for (int i = tid; i < 20480; i += 256) smem[i] = -1.0f;   // poison before the next repetition
    __syncthreads();
  }
  if (tid == 0) atomicAdd(&out[80], 1u);
}

void launch_lds_dma_probe(const float* pattern, unsigned int* out, int blocks, int spin, hipStream_t s) {
  hipLaunchKernelGGL(k_lds_dma_probe, dim3(blocks), dim3(256), 0, s, pattern, out, spin);
}

void launch_mfma_probe(float* out, int blocks, int iters, hipStream_t s) {
  hipLaunchKernelGGL(k_mfma_probe, dim3(blocks), dim3(256), 0, s, out, iters);
}

void launch_gemm_f32(int epi, const GemmArgs& a, int batch, hipStream_t s) {
  g_last_kernel = g_gemm_variant == 5 || (g_gemm_variant >= 50 && g_gemm_variant < 60) ? "k_gemm_f32x3<" : g_gemm_variant >= 6 ? "k_gemm_f16x2<" : "k_gemm_f32_v3<";
  dim3 grid(a.N / BN, a.M / BM, batch), block(256);
  if (g_gemm_variant == 4 && epi == EPI_BIAS) {
    hipLaunchKernelGGL((k_gemm_f32_v3<EPI_BIAS, false>), grid, block, 0, s, a);
    return;
  }
  if (g_gemm_variant >= 60 && epi == EPI_BIAS && a.Wp) {   // f16x2 ablation builds (timing only, wrong numbers)
    switch (g_gemm_variant) {
      case 61: hipLaunchKernelGGL((k_gemm_f16x2<EPI_BIAS, true, 1>), grid, block, 0, s, a); break;
      case 62: hipLaunchKernelGGL((k_gemm_f16x2<EPI_BIAS, true, 2>), grid, block, 0, s, a); break;
      case 63: hipLaunchKernelGGL((k_gemm_f16x2<EPI_BIAS, true, 3>), grid, block, 0, s, a); break;
      case 64: hipLaunchKernelGGL((k_gemm_f16x2<EPI_BIAS, true, 4>), grid, block, 0, s, a); break;
      case 67: hipLaunchKernelGGL((k_gemm_f16x2<EPI_BIAS, true, 7>), grid, block, 0, s, a); break;
      default: hipLaunchKernelGGL((k_gemm_f16x2<EPI_BIAS, true, 0>), grid, block, 0, s, a); break;
    }
    return;
  }
  if (g_gemm_variant >= 50 && epi == EPI_BIAS && a.Wp) {   // ablation builds (timing only, wrong numbers)
    switch (g_gemm_variant) {
      case 51: hipLaunchKernelGGL((k_gemm_f32x3<EPI_BIAS, true, 1>), grid, block, 0, s, a); break;
      case 52: hipLaunchKernelGGL((k_gemm_f32x3<EPI_BIAS, true, 2>), grid, block, 0, s, a); break;
      case 53: hipLaunchKernelGGL((k_gemm_f32x3<EPI_BIAS, true, 3>), grid, block, 0, s, a); break;
      case 54: hipLaunchKernelGGL((k_gemm_f32x3<EPI_BIAS, true, 4>), grid, block, 0, s, a); break;
      case 57: hipLaunchKernelGGL((k_gemm_f32x3<EPI_BIAS, true, 7>), grid, block, 0, s, a); break;
      default: hipLaunchKernelGGL((k_gemm_f32x3<EPI_BIAS, true, 0>), grid, block, 0, s, a); break;
    }
    return;
  }
  if (g_gemm_variant == 6) {
#define GN_H2(E)                                                                                   \
  if (a.Wp) hipLaunchKernelGGL((k_gemm_f16x2<E, true>), grid, block, 0, s, a);                     \
  else hipLaunchKernelGGL((k_gemm_f16x2<E, false>), grid, block, 0, s, a);
    switch (epi) {
      case EPI_BIAS: GN_H2(EPI_BIAS) break;
      case EPI_SCALE_COLS: GN_H2(EPI_SCALE_COLS) break;
      case EPI_ROTARY: GN_H2(EPI_ROTARY) break;
      case EPI_RESIDUAL: GN_H2(EPI_RESIDUAL) break;
      case EPI_ROTARY_BF16: GN_H2(EPI_ROTARY_BF16) break;
      case EPI_SCALE_BF16: GN_H2(EPI_SCALE_BF16) break;
      case EPI_RELU: GN_H2(EPI_RELU) break;
      default: GN_H2(EPI_PLAIN) break;
    }
#undef GN_H2
    return;
  }
  if (g_gemm_variant == 5) {
#define GN_X3(E)                                                                                   \
  if (a.Wp) hipLaunchKernelGGL((k_gemm_f32x3<E, true>), grid, block, 0, s, a);                     \
  else hipLaunchKernelGGL((k_gemm_f32x3<E, false>), grid, block, 0, s, a);
    switch (epi) {
      case EPI_BIAS: GN_X3(EPI_BIAS) break;
      case EPI_SCALE_COLS: GN_X3(EPI_SCALE_COLS) break;
      case EPI_ROTARY: GN_X3(EPI_ROTARY) break;
      case EPI_RESIDUAL: GN_X3(EPI_RESIDUAL) break;
      case EPI_ROTARY_BF16: GN_X3(EPI_ROTARY_BF16) break;
      case EPI_SCALE_BF16: GN_X3(EPI_SCALE_BF16) break;
      default: GN_X3(EPI_PLAIN) break;
    }
#undef GN_X3
    return;
  }
  // small grids: 64-row tiles when 128 x 128 tiles would leave more than a third of the CUs without a workgroup (same bits: k_gemm_f32_m64)
  if (g_gemm_m64 && (epi == EPI_BIAS || epi == EPI_PLAIN || epi == EPI_RELU || epi == EPI_SCALE_COLS || epi == EPI_ROTARY || epi == EPI_RESIDUAL) &&
      (long long)grid.x * grid.y * grid.z <= g_gemm_m64) {
    const dim3 g64(a.N / BN, a.M / 64, batch);
    // developer knob 44: 0 = k_gemm_f32_m64 always; 1 (default) = k_gemm_f32_r64 on 64 x 64 tiles when 64 x 128 tiles leave half of the CUs idle (at most
    // 128 workgroups: 19.9 -> 14.5 us per GEMM of a one-pair call), k_gemm_f32_m64 otherwise; 2 = the ring kernel on 64 x 128 tiles everywhere (measured:
    // 19.4 against 19.9 us -- it is the tile count, not the ring, that pays; its 96 KB of LDS leave one workgroup per CU)
    const bool n64 = g_gemm_r64 == 1 && (long long)g64.x * g64.y * g64.z <= 128 && a.N % 64 == 0;
    if (n64 || g_gemm_r64 == 2) {
      const dim3 gr(n64 ? a.N / 64 : a.N / BN, a.M / 64, batch);
      g_last_kernel = "k_gemm_f32_r64<";
#define GN_R64(E) do { if (n64) hipLaunchKernelGGL((k_gemm_f32_r64<E, 64>), gr, block, 0, s, a); else hipLaunchKernelGGL((k_gemm_f32_r64<E, 128>), gr, block, 0, s, a); } while (0)
      if (epi == EPI_BIAS) GN_R64(EPI_BIAS); else if (epi == EPI_RELU) GN_R64(EPI_RELU); else if (epi == EPI_SCALE_COLS) GN_R64(EPI_SCALE_COLS);
      else if (epi == EPI_ROTARY) GN_R64(EPI_ROTARY); else if (epi == EPI_RESIDUAL) GN_R64(EPI_RESIDUAL); else GN_R64(EPI_PLAIN);
#undef GN_R64
      return;
    }
    g_last_kernel = "k_gemm_f32_m64<";
    if (epi == EPI_BIAS) hipLaunchKernelGGL(k_gemm_f32_m64<EPI_BIAS>, g64, block, 0, s, a);
    else if (epi == EPI_RELU) hipLaunchKernelGGL(k_gemm_f32_m64<EPI_RELU>, g64, block, 0, s, a);
    else if (epi == EPI_SCALE_COLS) hipLaunchKernelGGL(k_gemm_f32_m64<EPI_SCALE_COLS>, g64, block, 0, s, a);
    else if (epi == EPI_ROTARY) hipLaunchKernelGGL(k_gemm_f32_m64<EPI_ROTARY>, g64, block, 0, s, a);
    else if (epi == EPI_RESIDUAL) hipLaunchKernelGGL(k_gemm_f32_m64<EPI_RESIDUAL>, g64, block, 0, s, a);
    else hipLaunchKernelGGL(k_gemm_f32_m64<EPI_PLAIN>, g64, block, 0, s, a);
    return;
  }
  // exact-f32 MFMA (k_gemm_f32_v3: LDS-DMA, double-buffered, LDS-slab epilogue) -- GN_PREC_F32 / GN_PREC_BF16_ATTN and the VO matcher
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL(k_gemm_f32_v3<EPI_BIAS>, grid, block, 0, s, a); break;
    case EPI_SCALE_COLS: hipLaunchKernelGGL(k_gemm_f32_v3<EPI_SCALE_COLS>, grid, block, 0, s, a); break;
    case EPI_ROTARY: hipLaunchKernelGGL(k_gemm_f32_v3<EPI_ROTARY>, grid, block, 0, s, a); break;
    case EPI_RESIDUAL: hipLaunchKernelGGL(k_gemm_f32_v3<EPI_RESIDUAL>, grid, block, 0, s, a); break;
    case EPI_ROTARY_BF16: hipLaunchKernelGGL(k_gemm_f32_v3<EPI_ROTARY_BF16>, grid, block, 0, s, a); break;
    case EPI_SCALE_BF16: hipLaunchKernelGGL(k_gemm_f32_v3<EPI_SCALE_BF16>, grid, block, 0, s, a); break;
    case EPI_RELU: hipLaunchKernelGGL(k_gemm_f32_v3<EPI_RELU>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(k_gemm_f32_v3<EPI_PLAIN>, grid, block, 0, s, a); break;
  }
}

}  // namespace gn
