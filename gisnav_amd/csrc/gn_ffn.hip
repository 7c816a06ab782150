// Fused transformer-block tail of the f16x2 precision mode: one launch per block computes, for 64 tokens per workgroup,
//
//     x  <-  x + ffn.3( GELU( LayerNorm( ffn.0( [x | msg] ) ) ) )                                  (k_ffn_fused)
//
// i.e. kornia's TransformerLayer / CrossBlock tail `x + self.ffn(torch.cat([x, message], -1))` (reached from
// ros/gisnav/gisnav/core/pose_node.py:285-287).  The 512-wide hidden tensor never leaves the CU: it goes from the accumulators
// of the first GEMM through LayerNorm + GELU in registers into LDS (128 KB as hm16) and is the operand of the second GEMM.
// Replaces k_gemm_p2ln + k_gemm_p2w<EPI_RESIDUAL> (gn_gemm_p2.hip), which wrote and re-read 134 MB of hidden rows per FFN.
//
// Why it is laid out the way it is (measured facts from round 1):
//  * the split-operand GEMMs of this path were bound by the L2 -> LDS fill rate (LDS-DMA, ~10 B/clk/CU), most of it WEIGHT tiles
//    that every workgroup re-streams.  Here the weights never touch LDS: they are re-laid-out ONCE, at load time, into MFMA
//    fragment order ("wf": per (32-row tile, 16-wide k-step, term) one 1 KB block, lane l -> bytes [16 l, 16 l + 16)), so a wave
//    fetches a fragment with one fully coalesced global_load_dwordx4 straight into the registers the MFMA reads.  Only the
//    token tile (8 KB per 32-wide k-tile) is staged through LDS, by plain loads + ds_write (no LDS-DMA in the loop, so the
//    compiler's counted vmcnt waits stay exact).
//  * both GEMMs are computed TRANSPOSED (H^T = W1 X^T, Y^T = W2 H^T): weights are the MFMA A operand, tokens the B operand.
//    A lane of the first GEMM's result then owns ONE token and 16 hidden units per tile -- LayerNorm statistics are an
//    in-lane sum + one cross-half shuffle + a 2 KB LDS exchange between the eight waves, and the 8 values a lane holds per
//    k-step are exactly one 16-byte B-operand fragment of the second GEMM once W2's columns are permuted inside 16-groups
//    (done in the weight re-layout): publishing the hidden tile is 2 ds_write_b128 per fragment, no cross-lane traffic.
//
// 8 waves; wave w owns hidden units [64 w, 64 w + 64) in the first GEMM (2 x 2 MFMA tiles of 32 x 32) and output features
// [32 w, 32 w + 32) in the second (1 x 2 tiles).  Arithmetic is the hm16 scheme of gn_gemm_p2.hip: x = xh + xm in fp16, three
// v_mfma_f32_32x32x16_f16 per block (small terms first), f32 accumulation.
#include "gn_common.h"

namespace gn {

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int TM = 64;                  // tokens per workgroup
constexpr int KT = TM * 128;            // bytes of one 32-wide k-tile of 64 token rows (hm16: 128 B per row)
constexpr int HBUF = 16 * KT;           // hidden tile: 512 units = 16 k-tiles (128 KB)
constexpr int STAT = HBUF;              // two [8 waves][64 tokens] float arrays behind it
constexpr int YP = 260;                 // float pitch of the output tile staged for the row-wise epilogue (aliases the hidden tile)
constexpr int SMEM = HBUF + 2 * 8 * TM * 4;

__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }

// 8 f32 -> 8 fp16 high terms and the 8 fp16 residual terms (round to nearest), as two 16-byte fragments
__device__ __forceinline__ void split8(const float* v, uint4& h, uint4& m) {
  unsigned int hw[4], mw[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x2v x = {v[2 * e], v[2 * e + 1]};
    const f16x2v hv = __builtin_convertvector(x, f16x2v);
    const f32x2v r = {x[0] - (float)hv[0], x[1] - (float)hv[1]};
    const f16x2v mv = __builtin_convertvector(r, f16x2v);
    hw[e] = __builtin_bit_cast(unsigned int, hv);
    mw[e] = __builtin_bit_cast(unsigned int, mv);
  }
  h = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  m = make_uint4(mw[0], mw[1], mw[2], mw[3]);
}

__global__ __launch_bounds__(512) void k_ffn_fused(FfnArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int bm = blockIdx.x * TM;

  // ---------------------------------------------------------------- GEMM 1 (transposed): H^T[512][64] = W1[512][512] . [x | msg]^T
  // token-tile staging: thread -> (row, 16-byte chunk) of the k-tile; k < 256 comes from x, k >= 256 from msg (both hm16, 1 KB rows)
  const int srow = tid >> 3, schunk = tid & 7;
  const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(a.xp) + (size_t)(bm + srow) * 1024 + schunk * 16;
  const unsigned char* msrc = reinterpret_cast<const unsigned char*>(a.mp) + (size_t)(bm + srow) * 1024 + schunk * 16;
  const int sdst = srow * 128 + ((schunk ^ swz(srow)) * 16);
  auto load_x = [&](int t) -> uint4 {
    const int k0 = t * 32;
    return *reinterpret_cast<const uint4*>(k0 < 256 ? xsrc + k0 * 4 : msrc + (k0 - 256) * 4);
  };
  // weight fragments: block ((tile * 32 + kstep) * 2 + term) of 64 uint4
  const uint4* w1f = reinterpret_cast<const uint4*>(a.w1s) + lane;
  f16x8 fa[2][2][2];   // [buffer][hidden tile][term]
  auto load_a = [&](int buf, int kk) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        fa[buf][i][pl] = __builtin_bit_cast(f16x8, w1f[(size_t)(((2 * wave + i) * 32 + kk) * 2 + pl) * 64]);
  };
  // token fragments of k-step ks of a stage: lane (row 32 j + ql, hh), term pl -> chunk 4 ks + 2 pl + hh
  int brow[2], bsw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { brow[j] = (32 * j + ql) * 128; bsw[j] = swz(32 * j + ql); }
  f16x8 fb[2][2];      // [token tile][term]
  auto read_b = [&](const unsigned char* base, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        fb[j][pl] = *reinterpret_cast<const f16x8*>(base + brow[j] + (((4 * ks + 2 * pl + hh) ^ bsw[j]) * 16));
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int NT = 16;   // k-tiles of 32 (K = 512)
  uint4 xr = load_x(0);
  *reinterpret_cast<uint4*>(smem + sdst) = xr;
  xr = load_x(1);
  load_a(0, 0);
  __syncthreads();
#pragma unroll 1
  for (int t = 0; t < NT; ++t) {
    unsigned char* const cur = smem + (t & 1) * KT;
    if (t + 1 < NT) *reinterpret_cast<uint4*>(smem + ((t + 1) & 1) * KT + sdst) = xr;   // tile t+1 (its stage was last read in iteration t-1)
    if (t + 2 < NT) xr = load_x(t + 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = 2 * t + ks;
      if (kk + 1 < 2 * NT) load_a((ks + 1) & 1, kk + 1);
      read_b(cur, ks);
      // products: W_m X_h, W_h X_m, W_h X_h (small terms first); the four accumulators are visited round-robin
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i][p == 0 ? 1 : 0], fb[j][p == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- bias, LayerNorm(512) (two-pass, eps 1e-5), erf GELU -- in the accumulators
  // lane (ql, hh) holds, for tokens 32 j + ql, the hidden units  64 w + 32 i + 8 g + 4 hh + c   (register r = 4 g + c)
  float* const stat1 = reinterpret_cast<float*>(smem + STAT);
  float* const stat2 = stat1 + 8 * TM;
  const float s1 = a.w1_scale;
  float sum[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(a.b1 + 64 * wave + 32 * i + 8 * g + 4 * hh);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v = acc[i][j][4 * g + c] * s1 + b[c];
          acc[i][j][4 * g + c] = v;
          sum[j] += v;
        }
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sum[j] += __shfl_xor(sum[j], 32);
    if (hh == 0) stat1[wave * TM + 32 * j + ql] = sum[j];
  }
  __syncthreads();
  float mean[2], rstd[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float t_ = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) t_ += stat1[w8 * TM + 32 * j + ql];
    mean[j] = t_ * (1.0f / 512.0f);
  }
  float sq[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[i][j][r] - mean[j];
        acc[i][j][r] = d;
        sq[j] += d * d;
      }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sq[j] += __shfl_xor(sq[j], 32);
    if (hh == 0) stat2[wave * TM + 32 * j + ql] = sq[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float t_ = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) t_ += stat2[w8 * TM + 32 * j + ql];
    rstd[j] = 1.0f / sqrtf(t_ * (1.0f / 512.0f) + 1e-5f);
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.ln_g + 64 * wave + 32 * i + 8 * g + 4 * hh);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(a.ln_b + 64 * wave + 32 * i + 8 * g + 4 * hh);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float y = gelu_erf(acc[i][j][4 * g + c] * rstd[j] * gm[c] + bt[c]);
          acc[i][j][4 * g + c] = y;
          amax = fmaxf(amax, fabsf(y));
        }
    }
  ovf_commit(a.ovf, amax);

  // ---------------------------------------------------------------- publish the hidden tile: k-tile 2 w + i, rows = tokens, the lane's 8 values of
  // k-step ks' (registers 8 ks' .. 8 ks' + 7) are one fragment (W2's columns are permuted to this order in the re-layout).
  // (the token stages of GEMM 1 alias this region: every wave has passed the two barriers above, so nobody reads them any more)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ksp = 0; ksp < 2; ++ksp) {
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = acc[i][j][8 * ksp + e];
        uint4 h4, m4;
        split8(v8, h4, m4);
        unsigned char* const base = smem + (2 * wave + i) * KT + brow[j];
        *reinterpret_cast<uint4*>(base + (((4 * ksp + hh) ^ bsw[j]) * 16)) = h4;
        *reinterpret_cast<uint4*>(base + (((4 * ksp + 2 + hh) ^ bsw[j]) * 16)) = m4;
      }
  __syncthreads();

  // ---------------------------------------------------------------- GEMM 2 (transposed): Y^T[256][64] = W2[256][512] . H^T;  wave w: output features 32 w ..
  const uint4* w2f = reinterpret_cast<const uint4*>(a.w2s) + lane + (size_t)wave * 32 * 2 * 64;
  f32x16 acc2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
  f16x8 ga[3][2];   // weight fragments, fetched two k-steps ahead
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) { ga[0][pl] = __builtin_bit_cast(f16x8, w2f[(0 * 2 + pl) * 64]); ga[1][pl] = __builtin_bit_cast(f16x8, w2f[(1 * 2 + pl) * 64]); }
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) {
    if (kk + 2 < 32) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) ga[(kk + 2) % 3][pl] = __builtin_bit_cast(f16x8, w2f[((kk + 2) * 2 + pl) * 64]);
    }
    read_b(smem + (kk >> 1) * KT, kk & 1);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[kk % 3][p == 0 ? 1 : 0], fb[j][p == 1 ? 1 : 0], acc2[j], 0, 0, 0);
  }
  __syncthreads();   // the hidden tile is dead: its space becomes the [64 tokens][256 features] f32 tile of the row-wise epilogue

  // ---------------------------------------------------------------- epilogue: + bias + residual x, hm16 (and optionally f32) rows
  float* const yt = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {acc2[j][4 * g], acc2[j][4 * g + 1], acc2[j][4 * g + 2], acc2[j][4 * g + 3]};
      *reinterpret_cast<f32x4*>(yt + (32 * j + ql) * YP + 32 * wave + 8 * g + 4 * hh) = v;
    }
  __syncthreads();
  const float s2 = a.w2_scale;
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.b2 + 4 * lane);
  f16x4 rh[8], rm[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {   // residual rows requested up front: one memory latency, not eight
    const uint16_t* rp = a.xp + hm16_off((size_t)(bm + 8 * wave + rr), kDim, 4 * lane);
    rh[rr] = *reinterpret_cast<const f16x4*>(rp);
    rm[rr] = *reinterpret_cast<const f16x4*>(rp + 16);
  }
  float amax2 = 0.f;
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int row = 8 * wave + rr;
    f32x4 v = *reinterpret_cast<const f32x4*>(yt + row * YP + 4 * lane) * s2;
    v += bias4;
    v += __builtin_convertvector(rh[rr], f32x4) + __builtin_convertvector(rm[rr], f32x4);
    const f16x4 hv = __builtin_convertvector(v, f16x4);
    const f16x4 mv = __builtin_convertvector(v - __builtin_convertvector(hv, f32x4), f16x4);
    uint16_t* yp = a.yp + hm16_off((size_t)(bm + row), kDim, 4 * lane);
    *reinterpret_cast<f16x4*>(yp) = hv;
    *reinterpret_cast<f16x4*>(yp + 16) = mv;
    if (a.y != nullptr) *reinterpret_cast<f32x4*>(a.y + (size_t)(bm + row) * kDim + 4 * lane) = v;
    ovf_track(amax2, v.x, v.y); ovf_track(amax2, v.z, v.w);
  }
  ovf_commit(a.ovf, amax2);
}
}  // namespace

void launch_ffn_fused(const FfnArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_ffn_fused, dim3(a.T / TM), dim3(512), 0, s, a);
}

// Weight re-layout into MFMA fragment order (host side, once per tensor at load time).
//   w [N][K] f32 (row = output feature), scale = power of two applied before the fp16 split (the kernels multiply the
//   accumulator by its inverse).  Block ((tile * (K / 16) + kstep) * 2 + term) holds, for lane l = (n = l & 31, hh = l >> 5),
//   the 8 halfs  term( w[32 tile + n][16 kstep + kperm(hh, e)] ), e = 0..7:
//     permute_k = 0: kperm = 8 hh + e                      (B operand read from an hm16 k-tile: natural order)
//     permute_k = 1: kperm = (e & 3) + 4 hh + 8 (e >> 2)   (B operand = the accumulator registers of a preceding transposed GEMM)
void build_weight_fragments(const float* w, int N, int K, float scale, int permute_k, uint16_t* out) {
  const int ksteps = K / 16;
  for (int tile = 0; tile < N / 32; ++tile)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int l = 0; l < 64; ++l) {
        const int n = l & 31, hh = l >> 5;
        for (int e = 0; e < 8; ++e) {
          const int kp = permute_k ? (e & 3) + 4 * hh + 8 * (e >> 2) : 8 * hh + e;
          const float x = w[(size_t)(32 * tile + n) * K + 16 * ks + kp] * scale;
          const _Float16 h = (_Float16)x;
          const _Float16 m = (_Float16)(x - (float)h);
          const size_t blk = ((size_t)tile * ksteps + ks) * 2;
          out[(blk + 0) * 512 + l * 8 + e] = __builtin_bit_cast(uint16_t, h);
          out[(blk + 1) * 512 + l * 8 + e] = __builtin_bit_cast(uint16_t, m);
        }
      }
}

}  // namespace gn
