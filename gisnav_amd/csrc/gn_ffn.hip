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
#include "gn_ffn_util.h"

namespace gn {

namespace {
// FOLD: the message is not read from memory but computed here, msg = out_proj(ctx) (kornia `self.out_proj` / `self.to_out`), from
// the attention output rows -- the out_proj GEMM launch and the msg round trip through HBM disappear.
// Shapes of the same kernel: NW waves x NJ token tiles of 32 per workgroup.  (8, 2) is the bulk shape (64 tokens, one workgroup per
// CU); (8, 1) halves the tokens per workgroup for small grids (batch 1: 2048 tokens are 32 workgroups of 64 on 256 CUs).  (4, 1) --
// two workgroups per CU, whose VALU and MFMA phases could overlap -- was measured SLOWER at every batch size (each wave then streams
// twice the weight bytes with the same number of loads in flight) and is not instantiated.
//   NJ            token tiles of 32 per workgroup (every wave covers all of them)
//   NI = 16 / NW  hidden tiles of 32 per wave in GEMM 1;   NO = 8 / NW  output tiles of 32 per wave in GEMM 0 and GEMM 2
template <int ABL = 0, bool FOLD = true, int NW = 8, int NJ = 2>   // ABL, timing-only ablations: 1 no weight loads inside the loops, 2 no second GEMM, 4 no GELU; 8 = s_memtime stamps per phase into a.dbg_ts
__global__ __launch_bounds__(NW * 64) void k_ffn_fused(FfnArgs a) {
  constexpr int NI = 16 / NW, NO = 8 / NW;
  constexpr int TM = 32 * NJ;             // tokens per workgroup
  constexpr int KT = TM * 128;            // bytes of one 32-wide k-tile of TM token rows (hm16: 128 B per row)
  constexpr int HBUF = 16 * KT;           // hidden tile: 512 units = 16 k-tiles (128 KB / 64 KB)
  constexpr int STAT = HBUF;              // two [NW waves][TM tokens] float arrays behind it
  constexpr int SMEM = HBUF + 2 * NW * TM * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int bm = blockIdx.x * TM;
  long long ts[8];
  auto stamp = [&](int k) __attribute__((always_inline)) { if (ABL & 8) ts[k] = (long long)__builtin_amdgcn_s_memtime(); };
  stamp(0);

  // LDS: 16 slots of one k-tile each (64 token rows x 128 B, chunk c of row r at position c ^ swz(r)).  Slots 0..7 hold, in turn, the
  // attention-output tile (operand of the folded out_proj) and the x tile; slots 8..15 the message tile; all 16 the hidden tile.
  // token-tile staging: thread -> (row, 16-byte chunk) of a k-tile; all sources are hm16 rows of 256 values (1 KB)
  const int srow = (tid >> 3) & (TM - 1), schunk = tid & 7;   // TM < 8 NW: the upper waves stage the same rows again (identical bytes, no branch)
  const size_t soff = (size_t)(bm + srow) * 1024 + schunk * 16;
  const unsigned char* const xsrc = reinterpret_cast<const unsigned char*>(a.xp) + soff;
  const unsigned char* const msrc = reinterpret_cast<const unsigned char*>(FOLD ? a.cp : a.mp) + soff;   // ctx rows (FOLD) or msg rows
  const int sdst = srow * 128 + ((schunk ^ swz(srow)) * 16);
  // token fragments of k-step ks of a slot: lane (row 32 j + ql, hh), term pl -> chunk 4 ks + 2 pl + hh
  int brow[NJ], bsw[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { brow[j] = (32 * j + ql) * 128; bsw[j] = swz(32 * j + ql); }
  f16x8 fb[2][NJ][2];   // [buffer][token tile][term]: the fragments of k-step kk + 1 are read while k-step kk is on the matrix pipe
  auto read_b = [&](int buf, const unsigned char* base, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        fb[buf][j][pl] = *reinterpret_cast<const f16x8*>(base + brow[j] + (((4 * ks + 2 * pl + hh) ^ bsw[j]) * 16));
  };
  // a lane's accumulator registers 8 ks' .. 8 ks' + 7 of tile (., j) ARE one 16-byte B-operand fragment of a following transposed
  // GEMM (whose weight columns are permuted to this order in the re-layout): publishing costs two ds_write_b128 per fragment
  float amax = 0.f;
  auto publish = [&](const f32x16& v, int slot, int j, float scale, const float* bias) __attribute__((always_inline)) {
#pragma unroll
    for (int ksp = 0; ksp < 2; ++ksp) {
      float v8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v8[e] = bias ? v[8 * ksp + e] * scale + bias[8 * ksp + e] : v[8 * ksp + e];
        amax = fmaxf(amax, fabsf(v8[e]));
      }
      uint4 h4, m4;
      split8(v8, h4, m4);
      unsigned char* const base = smem + slot * KT + brow[j];
      *reinterpret_cast<uint4*>(base + (((4 * ksp + hh) ^ bsw[j]) * 16)) = h4;
      *reinterpret_cast<uint4*>(base + (((4 * ksp + 2 + hh) ^ bsw[j]) * 16)) = m4;
    }
  };

  // ---------------------------------------------------------------- prologue: the message side of the token tile
  uint4 xr0, xr1, xr2;   // three named registers, not an array: an array here is promoted to LDS (48 B per lane) before the unroll makes its indices constant
  auto xr = [&](int q) __attribute__((always_inline)) -> uint4& { return q == 0 ? xr0 : (q == 1 ? xr1 : xr2); };
  {
    uint4 mt[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) mt[q] = *reinterpret_cast<const uint4*>(msrc + q * 128);
#pragma unroll
    for (int q = 0; q < 3; ++q) xr(q) = *reinterpret_cast<const uint4*>(xsrc + q * 128);      // first x k-tiles, written after the message phase
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<uint4*>(smem + ((FOLD ? 0 : 8) + q) * KT + sdst) = mt[q];
  }
  // weight fragments travel L2 -> registers with ~1-2 us of latency under load: rings of k-steps keep 16-24 KB per wave in flight;
  // the slot of k-step kk is refilled as soon as its MFMAs are issued.  Every loop is fully unrolled and branch-free, so each
  // s_waitcnt the compiler places is an exact count.
  const uint4* w1f = reinterpret_cast<const uint4*>(a.w1s) + lane;
  constexpr int RA = NW == 8 ? (NJ == 2 ? 6 : 8) : 4;     // k-steps in flight: NI x 4 KB per wave each
  f16x8 fa[RA][NI][2];   // [ring slot = k-step % RA][hidden tile][term]
  // GEMM 1 visits the k-tiles in the order 8..15, 0..7 (message first: it is in LDS already): iteration n -> slot (n + 8) & 15
  auto load_a = [&](int slot, int n2) __attribute__((always_inline)) {      // n2 = 2 * iteration + k-step
    const int kk = (n2 + 16) & 31;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        fa[slot][i][pl] = __builtin_bit_cast(f16x8, w1f[(size_t)(((NI * wave + i) * 32 + kk) * 2 + pl) * 64]);
  };
  if (FOLD) {
    // ------------------------------------------------------------ GEMM 0 (transposed): Msg^T[256][64] = Wo[256][256] . Ctx^T; wave w: features 32 w ..
    const uint4* wof = reinterpret_cast<const uint4*>(a.wos) + lane + (size_t)(NO * wave) * 16 * 2 * 64;      // output tile NO w + o: block ((tile * 16 + kstep) * 2 + term)
    constexpr int RO = NW == 8 ? 8 : 6;
    f16x8 go[RO][NO][2];
#pragma unroll
    for (int q = 0; q < RO; ++q)
#pragma unroll
      for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) go[q][o][pl] = __builtin_bit_cast(f16x8, wof[((o * 16 + q) * 2 + pl) * 64]);
    f32x16 acc0[NO][NJ];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[o][j][r] = 0.f;
    __syncthreads();
    stamp(1);
    read_b(0, smem, 0);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int cb = kk & 1;
      if (kk + 1 < 16) read_b(cb ^ 1, smem + ((kk + 1) >> 1) * KT, (kk + 1) & 1);
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc0[o][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(go[kk % RO][o][p == 0 ? 1 : 0], fb[cb][j][p == 1 ? 1 : 0], acc0[o][j], 0, 0, 0);
      if (!(ABL & 1) && kk + RO < 16) {
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) go[kk % RO][o][pl] = __builtin_bit_cast(f16x8, wof[((o * 16 + kk + RO) * 2 + pl) * 64]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < RA; ++q) load_a(q, q);
    // message = acc0 * scale + bias: feature 32 (NO w + o) + 8 g + 4 hh + c in register 4 g + c -> slot 8 + NO w + o
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      float bo[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bo + 32 * (NO * wave + o) + 8 * g + 4 * hh);
        bo[4 * g] = b4.x; bo[4 * g + 1] = b4.y; bo[4 * g + 2] = b4.z; bo[4 * g + 3] = b4.w;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) publish(acc0[o][j], 8 + NO * wave + o, j, a.wo_scale, bo);
    }
    __syncthreads();     // message tile complete; every wave is done with the attention-output tile in slots 0..7
  } else {
#pragma unroll
    for (int q = 0; q < RA; ++q) load_a(q, q);
    __syncthreads();
    stamp(1);
  }

  // ---------------------------------------------------------------- GEMM 1 (transposed): H^T[512][64] = W1[512][512] . [x | msg]^T
  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // iterations 0..7 consume the message slots and, on the side, stage x k-tile n into slot n (loaded three iterations earlier);
  // ONE barrier before iteration 8 makes the x slots readable
  read_b(0, smem + 8 * KT, 0);
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    if (n < 8) {
      *reinterpret_cast<uint4*>(smem + n * KT + sdst) = xr(n % 3);
      if (n + 3 < 8) xr(n % 3) = *reinterpret_cast<const uint4*>(xsrc + (n + 3) * 128);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int n2 = 2 * n + ks, slot = n2 % RA, cb = n2 & 1;
      if (n2 + 1 < 32 && n2 + 1 != 16) read_b(cb ^ 1, smem + ((((n2 + 1) >> 1) + 8) & 15) * KT, (n2 + 1) & 1);
      // products: W_m X_h, W_h X_m, W_h X_h (small terms first); the four accumulators are visited round-robin
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[slot][i][p == 0 ? 1 : 0], fb[cb][j][p == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
      if (!(ABL & 1) && n2 + RA < 32) load_a(slot, n2 + RA);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (n == 7) { __syncthreads(); read_b(0, smem, 0); }     // x tile complete (n2 + 1 == 16 is the one fragment set that could not be prefetched)
  }
  __syncthreads();   // every wave is done with the token tile: its space is reused below
  stamp(2);

  // the second GEMM's first weight fragments are requested now, so that their latency hides behind LayerNorm + GELU
  const uint4* w2f = reinterpret_cast<const uint4*>(a.w2s) + lane + (size_t)(NO * wave) * 32 * 2 * 64;        // output tile NO w + o: block ((tile * 32 + kstep) * 2 + term)
  constexpr int RG = NW == 8 ? 8 : 6;
  f16x8 ga[RG][NO][2];   // weight fragments: a ring of k-steps (that loop has 6 MFMAs per k-step and wave in both shapes)
#pragma unroll
  for (int q = 0; q < RG; ++q)
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) ga[q][o][pl] = __builtin_bit_cast(f16x8, w2f[((o * 32 + q) * 2 + pl) * 64]);

  // ---------------------------------------------------------------- bias, LayerNorm(512) (two-pass, eps 1e-5), erf GELU -- in the accumulators
  // lane (ql, hh) holds, for tokens 32 j + ql, the hidden units  32 (NI w + i) + 8 g + 4 hh + c   (register r = 4 g + c)
  float* const stat1 = reinterpret_cast<float*>(smem + STAT);
  float* const stat2 = stat1 + NW * TM;
  const float s1 = a.w1_scale;
  f32x2v sum2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) sum2[j] = splat2(0.f);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(a.b1 + 32 * (NI * wave + i) + 8 * g + 4 * hh);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          const f32x2v v = pair(acc[i][j], 4 * g + c) * splat2(s1) + (f32x2v){b[c], b[c + 1]};
          set_pair(acc[i][j], 4 * g + c, v);
          sum2[j] += v;
        }
    }
  float sum[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    sum[j] = sum2[j][0] + sum2[j][1];
    sum[j] += __shfl_xor(sum[j], 32);
    if (hh == 0) stat1[wave * TM + 32 * j + ql] = sum[j];
  }
  __syncthreads();
  float mean[NJ], rstd[NJ], sq[NJ];
  f32x2v sq2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float t_ = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < NW; ++w8) t_ += stat1[w8 * TM + 32 * j + ql];
    mean[j] = t_ * (1.0f / 512.0f);
    sq2[j] = splat2(0.f);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2v d = pair(acc[i][j], r) - splat2(mean[j]);
        set_pair(acc[i][j], r, d);
        sq2[j] += d * d;
      }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    sq[j] = sq2[j][0] + sq2[j][1];
    sq[j] += __shfl_xor(sq[j], 32);
    if (hh == 0) stat2[wave * TM + 32 * j + ql] = sq[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float t_ = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < NW; ++w8) t_ += stat2[w8 * TM + 32 * j + ql];
    rstd[j] = 1.0f / sqrtf(t_ * (1.0f / 512.0f) + 1e-5f);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.ln_g + 32 * (NI * wave + i) + 8 * g + 4 * hh);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(a.ln_b + 32 * (NI * wave + i) + 8 * g + 4 * hh);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          const f32x2v yn = pair(acc[i][j], 4 * g + c) * splat2(rstd[j]) * (f32x2v){gm[c], gm[c + 1]} + (f32x2v){bt[c], bt[c + 1]};
          set_pair(acc[i][j], 4 * g + c, (ABL & 4) ? yn : gelu_erf2(yn));
        }
    }
  stamp(3);

  // ---------------------------------------------------------------- publish the hidden tile: wave w's units 32 (NI w + i) .. -> slot NI w + i
  // (the token tile of GEMM 1 occupied this region: every wave has passed the barriers above, so nobody reads it any more)
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) publish(acc[i][j], NI * wave + i, j, 1.f, nullptr);
  ovf_commit(a.ovf, amax);
  __syncthreads();
  stamp(4);

  // ---------------------------------------------------------------- GEMM 2 (transposed): Y^T[256][TM] = W2[256][512] . H^T;  wave w: output features 32 NO w ..
  f32x16 acc2[NO][NJ];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[o][j][r] = 0.f;
  if (!(ABL & 2)) {
    read_b(0, smem, 0);
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      const int cb = kk & 1;
      if (kk + 1 < 32) read_b(cb ^ 1, smem + ((kk + 1) >> 1) * KT, (kk + 1) & 1);
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc2[o][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[kk % RG][o][p == 0 ? 1 : 0], fb[cb][j][p == 1 ? 1 : 0], acc2[o][j], 0, 0, 0);
      if (!(ABL & 1) && kk + RG < 32) {
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) ga[kk % RG][o][pl] = __builtin_bit_cast(f16x8, w2f[((o * 32 + kk + RG) * 2 + pl) * 64]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  stamp(5);
  __syncthreads();   // the hidden tile is dead: its space becomes the [TM tokens][256 features] f32 tile of the row-wise epilogue
  stamp(6);

  // ---------------------------------------------------------------- epilogue: + bias + residual x, hm16 (and optionally f32) rows
  float* const yt = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc2[o][j][4 * g], acc2[o][j][4 * g + 1], acc2[o][j][4 * g + 2], acc2[o][j][4 * g + 3]};
        *reinterpret_cast<f32x4*>(yt + (32 * j + ql) * YP + 32 * (NO * wave + o) + 8 * g + 4 * hh) = v;
      }
  __syncthreads();
  const float s2 = a.w2_scale;
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.b2 + 4 * lane);
  constexpr int RW = TM / NW;   // token rows per wave
  f16x4 rh[RW], rm[RW];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {   // residual rows requested up front: one memory latency, not RW
    const uint16_t* rp = a.xp + hm16_off((size_t)(bm + RW * wave + rr), kDim, 4 * lane);
    rh[rr] = *reinterpret_cast<const f16x4*>(rp);
    rm[rr] = *reinterpret_cast<const f16x4*>(rp + 16);
  }
  float amax2 = 0.f;
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int row = RW * wave + rr;
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
  if (ABL & 8) {
    stamp(7);
    if (a.dbg_ts != nullptr && lane == 0)
      for (int k = 0; k < 8; ++k) a.dbg_ts[((size_t)blockIdx.x * NW + wave) * 8 + k] = ts[k];
  }
}
}  // namespace

int g_ffn_ablate = 0;   // developer knob 12: timing-only ablations of k_ffn_fused (wrong results)
int g_ffn_shape = 0;    // developer knob 14: 0 = automatic, 64 / 32 = force the 64-token / 32-token workgroup shape
bool ffn_selects_128(const FfnArgs& a) {
  return a.cp != nullptr && a.T % 128 == 0 && (g_ffn_shape == 128 || (g_ffn_shape == 0 && a.T / 128 >= 256));
}
void launch_ffn_fused(const FfnArgs& a_in, hipStream_t s) {
  FfnArgs a = a_in;
  if (a.composed && !(a.T % 128 == 0 && (g_ffn_shape == 128 || (g_ffn_shape == 0 && a.T / 128 >= 256)))) {
    // composed weights on a small grid: k_ffn_fused's "message rows from memory" form IS ffn.0 over [x | rows] -- the rows are the attention output
    a.mp = a.cp; a.cp = nullptr; a.composed = 0;
  }
  // small grids: 32-token workgroups give twice the workgroups (fewer than one 64-token workgroup per CU leaves CUs idle)
  const bool small = g_ffn_shape == 32 || (g_ffn_shape == 0 && a.T / 64 < 256);
  if (a.cp == nullptr) {   // message rows from memory (separate out_proj launch)
    if (small) { hipLaunchKernelGGL((k_ffn_fused<0, false, 8, 1>), dim3(a.T / 32), dim3(512), 0, s, a); g_last_kernel = "k_ffn_fused<0, false, 8, 1>"; }
    else { hipLaunchKernelGGL((k_ffn_fused<0, false, 8, 2>), dim3(a.T / 64), dim3(512), 0, s, a); g_last_kernel = "k_ffn_fused<0, false, 8, 2>"; }
    return;
  }
  // bulk grids: 128 tokens per workgroup, one wave per SIMD (gn_ffn128.hip); developer knob 14 = 128 / 64 / 32 forces a shape
  if (a.T % 128 == 0 && (g_ffn_shape == 128 || (g_ffn_shape == 0 && a.T / 128 >= 256))) { launch_ffn128(a, g_ffn_ablate, s); return; }
  if (small) {
    if (g_ffn_ablate == 8) hipLaunchKernelGGL((k_ffn_fused<8, true, 8, 1>), dim3(a.T / 32), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((k_ffn_fused<0, true, 8, 1>), dim3(a.T / 32), dim3(512), 0, s, a);
    g_last_kernel = "k_ffn_fused<0, true, 8, 1>";
    return;
  }
  const dim3 grid(a.T / 64), block(512);
  switch (g_ffn_ablate) {
    case 1: hipLaunchKernelGGL((k_ffn_fused<1, true, 8, 2>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_ffn_fused<2, true, 8, 2>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_ffn_fused<3, true, 8, 2>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((k_ffn_fused<4, true, 8, 2>), grid, block, 0, s, a); break;
    case 7: hipLaunchKernelGGL((k_ffn_fused<7, true, 8, 2>), grid, block, 0, s, a); break;
    case 8: hipLaunchKernelGGL((k_ffn_fused<8, true, 8, 2>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((k_ffn_fused<0, true, 8, 2>), grid, block, 0, s, a); break;
  }
  g_last_kernel = "k_ffn_fused<0, true, 8, 2>";
}

// Weight re-layout into MFMA fragment order (host side, once per tensor at load time).
//   w [N][K] f32 (row = output feature), scale = power of two applied before the fp16 split (the kernels multiply the
//   accumulator by its inverse).  Block ((tile * (K / 16) + kstep) * 2 + term) holds, for lane l = (n = l & 31, hh = l >> 5),
//   the 8 halfs  term( w[32 tile + n][16 kstep + kperm(hh, e)] ), e = 0..7:
//     natural : kperm = 8 hh + e                      (B operand read from an hm16 k-tile as the producer kernel stored it)
//     permuted: kperm = (e & 3) + 4 hh + 8 (e >> 2)   (B operand = the accumulator registers of a preceding transposed GEMM)
//   permute_k = 0: natural everywhere; 1: permuted everywhere; 2: natural for k < K / 2, permuted for k >= K / 2 (ffn.0 with the
//   message half produced in-kernel)
void build_weight_fragments(const float* w, int N, int K, float scale, int permute_k, uint16_t* out) {
  const int ksteps = K / 16;
  for (int tile = 0; tile < N / 32; ++tile)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int l = 0; l < 64; ++l) {
        const int n = l & 31, hh = l >> 5;
        for (int e = 0; e < 8; ++e) {
          const bool perm = permute_k == 1 || (permute_k == 2 && 16 * ks >= K / 2);
          const int kp = perm ? (e & 3) + 4 * hh + 8 * (e >> 2) : 8 * hh + e;
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
