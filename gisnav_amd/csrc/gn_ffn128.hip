// Fused transformer-block tail, 128 tokens per workgroup (round 4): the same arithmetic as k_ffn_fused (gn_ffn.hip)
//
//     msg = out_proj(ctx);   x  <-  x + ffn.3( GELU( LayerNorm( ffn.0( [x | msg] ) ) ) )
//
// i.e. kornia's TransformerLayer / CrossBlock tail `x + self.ffn(torch.cat([x, message], -1))` (reached from
// ros/gisnav/gisnav/core/pose_node.py:285-287), re-shaped around what round 3 measured about k_ffn_fused: a 64-token workgroup
// pulls the block's 1.79 MB of weight fragments from L2 once per 64 tokens -- at the matrix-pipe floor of its GEMM 1 that is
// 42 B/clk/CU, more than a CU's L2 port delivers (~30 B/clk measured), so the k-loops ran at the weight stream's pace, and the
// LayerNorm / GELU / split phases of its eight waves (two per SIMD, in lock-step) never overlapped with matrix work.
//
// Here ONE wave per SIMD (4 waves, 256 arch + 256 accumulator registers each) covers 128 tokens:
//  * every weight byte is fetched once per 128 tokens: 21 B/clk/CU at the matrix-pipe floor;
//  * GEMM 1's accumulators (128 hidden units x 128 tokens per wave = 256 registers) stay in the accumulator file through LayerNorm:
//    the statistics pass only READS them (shifted one-pass sums per lane, merged with Chan's formula across the 8 partials of a token),
//    and normalisation is re-applied from the raw accumulator when a value is consumed;
//  * the hidden tensor (256 KB as hm16 for 128 tokens) never exists as a whole: it is normalised, GELU'd, split and published to LDS
//    a QUARTER at a time (32 units per wave = 4 k-tiles = 64 KB, two buffers); quarters 2 and 3 are produced INSIDE the second GEMM's
//    instruction stream, one ~8-instruction VALU stage in the shadow of each MFMA (VALU and MFMA overlap inside one wave's stream,
//    not between the two waves of a SIMD: tools/probes/overlap.hip);
//  * token rows (attention output, then x) stream through a two-slot ring of 32-wide k-tiles (16 KB each) behind the 128 KB message tile:
//    LDS = 160 KB exactly.
// The instruction order of every k-step is pinned (sched_barrier after each MFMA): one memory instruction or one VALU stage per MFMA
// gap -- left alone the machine scheduler clumps the weight loads (each stalls the wave's issue for ~30 cycles) and runs the VALU
// work of a GELU chunk as one dependent chain per value pair BETWEEN the MFMA groups.
// Default form since round 4 (COMP = true): out_proj is folded into ffn.0 at load time -- ffn.0([x | Wo ctx + bo]) = [W1x | W1m Wo] [x | ctx] + (b1 + W1m bo),
// gn_api.hip build_composed -- so there is no GEMM 0, no message tile and no round trip of it through LDS: the prologue stages the first token
// tiles, the weight ring and the bias, and GEMM 1 runs over all sixteen [x | ctx] k-tiles (51.5 instead of 60.1 GFLOP per launch; 156-160 us
// against 173-187 us).  The description above is the uncomposed form (COMP = false: knob 28 = 0, and every grid the composed weight was not
// built for); the two share everything from the LayerNorm statistics on.
// Weight fragments: the layouts of gn_ffn.hip (build_weight_fragments), unchanged.  Accumulation order per output element equals
// k_ffn_fused's for GEMM 0 and GEMM 1 (bitwise the same message and pre-LayerNorm values); the LayerNorm statistics, the last
// multiply of the GELU (one fma instead of mul + add + mul) and GEMM 2's k order differ at rounding level.
#include "gn_common.h"
#include <algorithm>
#include "gn_ffn_util.h"

namespace gn {

namespace {

#define GN_PIN() __builtin_amdgcn_sched_barrier(0)

// y - float(fp16 half of h): one v_fma_mix_f32 (the f16 operand is converted inside the instruction) instead of a conversion and a subtraction
__device__ __forceinline__ float resid_lo(unsigned int h, float y) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(y)); return r; }
// float(fp16 half of h) + y (the residual rows: x = x_h + x_m + v in two instructions)
__device__ __forceinline__ float addh_lo(unsigned int h, float y) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(y)); return r; }
__device__ __forceinline__ float addh_hi(unsigned int h, float y) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(y)); return r; }
__device__ __forceinline__ float resid_hi(unsigned int h, float y) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(y)); return r; }

// COMP: out_proj is composed into ffn.0 at load time (W1' = [W1_x | W1_m Wo], b1' = b1 + W1_m bo: gn_api.hip build_composed) -- no GEMM 0, no message
// tile: GEMM 1 runs over [x | ctx], all sixteen k-tiles through the ring
// QKV (round 5): 1 / 2 = the NEXT block's attention input projection (self: Wqkv + rotary; cross: to_qk | to_v) runs on the tile's new rows behind the epilogue
// PROD (round 6): 3 = every GEMM on the three partial products of the hm16 split (W_m X_h + W_h X_m + W_h X_h: f32-accurate); 2 = the products with the
// activations' fp16 HIGH term only (W_m X_h + W_h X_h: 22-bit weights x 11-bit activations, f32 accumulate) -- the arithmetic of the attention input
// projections since round 4, one rounding of the GEMM inputs to fp16, which the fp16 attention beside it applies to q, k, v and the probabilities anyway.
// A third of the matrix-pipe work of both GEMMs, half of the token-fragment reads, no residual split of the hidden tile.  Meant to run under the margin
// certificate (gn_set_certify), which makes the correspondence indices independent of the fast pass's arithmetic; composed form only.
template <int ABL, bool COMP, bool LOOP = false, int QKV = 0, int PROD = 3>   // LOOP: the workgroup walks the work list (one workgroup per CU); timing-only ablations (bits): 1 no weight loads inside the loops, 2 no token-row loads inside the loops, 4 no GELU polynomial, 16 no barriers inside the k-loops; 8 = s_memtime stamps per phase into a.dbg_ts (results stay valid)
__global__ __launch_bounds__(256) void k_ffn128(FfnArgs a_in) {
  static_assert(PROD == 3 || (PROD == 2 && COMP), "two partial products: composed form only");
  constexpr int NJ = 4, NI = 4, NO = 2, NW = 4, TM = 128;
  constexpr int KT = TM * 128;            // bytes of one 32-wide k-tile of 128 token rows (hm16: 128 B per row)
  constexpr int RING = 8 * KT;            // two staging slots behind the message tile
  constexpr int STAT = 8 * KT;            // LayerNorm statistics (4 KB) and the LayerNorm constants (4 KB) alias the ring (dead after GEMM 1)
  constexpr int CST = STAT + 4096;        // [2][512] floats: LayerNorm weight, LayerNorm bias
  constexpr int SMEM = 10 * KT;           // 163,840 B
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
  // persistent form: the workgroup walks the call's list of tiles that hold valid tokens (or, without a list, every gridDim.x-th tile of T)
  // LOOP = false is the one-tile kernel: the loop below is left after its first pass, and the compiler keeps nothing alive around it (the walking
  // form holds the arguments and the loop state in scalar registers through the weight streams: a few spills, ~5 % slower per tile)
  const int n_work = a_in.tiles != nullptr ? a_in.tiles[0] : a_in.T / TM;
#pragma unroll 1
  for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
#if defined(__HIP_DEVICE_COMPILE__)
  // the walking form reads its arguments from the kernel-argument segment again for every tile (through a pointer the compiler cannot see through), so
  // that they are not held in scalar registers across the loop
  typedef __attribute__((address_space(4))) const FfnArgs* karg_t;
  karg_t ap = (karg_t)__builtin_amdgcn_kernarg_segment_ptr();
  if (LOOP) asm volatile("" : "+s"(ap));
  const FfnArgs a = *ap;
#else
  const FfnArgs a = a_in;
#endif
  // (the thread index is opaque per iteration: everything derived from it -- lane, wave, every address -- is then recomputed inside the body instead of
  // being hoisted out of the loop, where it would live across the whole tile and push the kernel over its register budget)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int bm = (a.tiles != nullptr ? a.tiles[kTileListBase + work] : work) * TM;
  long long ts[12], ts2[16];
  auto stamp = [&](int k) __attribute__((always_inline)) { if (ABL & 8) ts[k] = (long long)__builtin_amdgcn_s_memtime(); };
  stamp(0);
  if (!LOOP && a.nvalid != nullptr) {
    // one-tile form: a tile that holds only padding (tokens at or behind its slot's valid count) leaves at once -- its rows of the residual stream
    // stay as they are (finite; no valid token reads them: keys are masked, everything else is row-wise); only an f32 copy wanted by the caller
    // (the last block's, read by the match head) is defined: zeros.  (Still one workgroup dispatch per skipped tile, ~40 ns each.)
    const int slot = bm / a.npad;
    if (bm - slot * a.npad >= a.nvalid[slot]) {
      if (a.y != nullptr) {
        float4* dst = reinterpret_cast<float4*>(a.y + (size_t)bm * kDim);
        for (int i = tid; i < TM * kDim / 4; i += 256) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
  }
  f32x4 b1q = {0.f, 0.f, 0.f, 0.f};      // ffn.0's bias, requested first: it is needed after GEMM 0 (see below)
  if (tid < 128) b1q = *reinterpret_cast<const f32x4*>(a.b1 + 4 * tid);

  // Addressing discipline (one wave per SIMD, 256 arch VGPRs beside 256 accumulator registers): every global access is
  // `buffer descriptor (SGPRs) + 32-bit lane offset (VGPR) + uniform offset (SGPR / literal)`, every LDS access `one of a few base
  // registers + 16-bit immediate` -- left to itself the compiler keeps one 64-bit pointer pair per weight load and one address
  // register per distinct LDS constant, and spills them around the k-loops.
  // ---- token-row staging: thread -> (row srow + 32 q, 16-byte chunk) of a k-tile; sources are hm16 rows of 256 values (1 KB)
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.cp) + (size_t)bm * 512, 0, TM * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.xp) + (size_t)bm * 512, 0, TM * 1024, 0x00020000);
  unsigned int soff[4], sdst[4];
  // (re-derived before the x half of GEMM 1, like the LDS windows below, from an OPAQUE copy of the thread index: otherwise the
  // compiler recognises the expressions of the first derivation and keeps -- i.e. spills and reloads -- those registers)
  auto stage_addr = [&]() __attribute__((always_inline)) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    const int sr = t_ >> 3, sc = t_ & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      soff[q] = (unsigned int)((sr + 32 * q) * 1024 + sc * 16);
      sdst[q] = (unsigned int)(RING + (sr + 32 * q) * 128 + ((sc ^ swz(sr + 32 * q)) * 16));
    }
  };
  stage_addr();
  // two register sets, staged tile t travels in set t & 1 -- eight named registers, not an array: an array indexed by the (unrolled) loop
  // counters is left in scratch memory
  uint4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
  auto stv = [&](int set, int q) __attribute__((always_inline)) -> uint4& {
    return set == 0 ? (q == 0 ? sa0 : (q == 1 ? sa1 : (q == 2 ? sa2 : sa3))) : (q == 0 ? sb0 : (q == 1 ? sb1 : (q == 2 ? sb2 : sb3)));
  };
  // staged tiles 0..7 = k-tiles of the attention output (GEMM 0), 8..15 = k-tiles of x (second half of GEMM 1); COMP: 0..7 = x, 8..15 = attention output; one piece = 32 rows
  auto stage_load = [&](int t, int q) __attribute__((always_inline)) {
    stv(t & 1, q) = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128((COMP ? t >= 8 : t < 8) ? crs : xrs, soff[q], (t & 7) * 128, 0));
  };
  auto stage_write = [&](int t, int q) __attribute__((always_inline)) {
    *reinterpret_cast<uint4*>(smem + sdst[q] + (t & 1) * KT) = stv(t & 1, q);
  };
  // ---- token fragments: lane (row 32 j + ql, hh), k-step ks of a k-tile, term pl -> chunk 4 ks + 2 pl + hh; swz(32 j + ql) depends on j & 1 only.
  // LDS byte address = bo[window][j & 1][2 ks + pl] + immediate < 64 KB: three 64 KB windows (made opaque so that the compiler keeps them)
  // (the windows a phase needs are re-derived at its start: a window register that lived from here to the second GEMM was spilled,
  // and each reload inside a k-loop costs a vmcnt(0) -- the whole weight ring drained)
  unsigned int bo[3][2][4];
  auto window = [&](int w) __attribute__((always_inline)) {
    int l_ = lane;
    asm volatile("" : "+v"(l_));
    const int q_ = l_ & 31, h_ = l_ >> 5;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bo[w][jp][c] = (unsigned int)((32 * jp + q_) * 128 + (((2 * c + h_) ^ swz(32 * jp + q_)) * 16)) + 65536u * w;
        asm volatile("" : "+v"(bo[w][jp][c]));
      }
  };
  window(2);
  // three fragment buffers: j-step g (k-step sequence number * 4 + token tile) lives in buffer g % 3 and is read one (GEMM 1: 12 MFMAs
  // per j-step) or two (GEMM 0 / 2: 6 MFMAs per j-step) j-steps ahead of its MFMAs
  f16x8 bq[3][2];
  // tile_off: byte offset of the k-tile inside smem (a compile-time constant after unrolling)
  auto read_b = [&](int g, int tile_off, int ks, int j) __attribute__((always_inline)) {
    const int off = tile_off + (j >> 1) * 8192;
#pragma unroll
    for (int pl = 0; pl < (PROD == 2 ? 1 : 2); ++pl) bq[g % 3][pl] = *reinterpret_cast<const f16x8*>(smem + bo[off >> 16][j & 1][2 * ks + pl] + (off & 65535));
  };
  // a lane's accumulator registers 8 ks' .. 8 ks' + 7 of tile (., j) ARE one 16-byte B-operand fragment of a following transposed
  // GEMM (whose weight columns are permuted to this order in the re-layout): publishing costs two ds_write_b128 per fragment.
  // pb: the publishing wave's base registers (its own k-tile inside the 64 KB window), off < 64 KB
  float amax = 0.f;
  unsigned int pb[2][4];
  const unsigned int lane16 = (unsigned int)lane * 16u;
  // one weight fragment: buffer descriptor of the wave's slice (SGPRs) + uniform byte offset (a multiple of 1 KB, an SGPR / literal) + lane * 16
  auto ldw = [&](const __amdgpu_buffer_rsrc_t& rs, int uoff) __attribute__((always_inline)) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, uoff, 0));
  };

  // ---- the VALU stage machine of one fragment (8 values of one token = one 16-byte B operand of the next GEMM): 8 independent
  // instructions per stage, so that a stage neither stalls on its own results nor outlasts the MFMA it hides behind
  float gy[8], gt[8], gq[8];
  unsigned int gh[4], gm[4];
  f32x4 c2[2], c3[2];             // LayerNorm weight, LayerNorm bias of the fragment's two groups of 4 units
  float rs[NJ], nmr[NJ];          // per token tile: s1 / sqrt(var + eps) (applied to the raw accumulator), -mean / sqrt(var + eps)
  const float s1 = a.w1_scale;    // a power of two: ffn.0's bias enters the accumulators as b1 / s1 before the first MFMA (exact)
  const unsigned int cbase = (unsigned int)(CST + hh * 16 + 32 * NI * wave * 4);
  // stages 18..21 (maximum, fp16 split, publish) are shared by the message and the hidden fragments
  auto tail_stage = [&](int st, int off, int j, int ksp) __attribute__((always_inline)) {
    if (st == 18) {
#pragma unroll
      for (int e = 0; e < 8; e += 2) { amax = fmaxf(amax, fmaxf(fabsf(gy[e]), fabsf(gy[e + 1]))); gh[e >> 1] = pack16<true>(gy[e], gy[e + 1]); }
    } else if (st == 19) {
      if constexpr (PROD == 3) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) { gt[e] = resid_lo(gh[e >> 1], gy[e]); gt[e + 1] = resid_hi(gh[e >> 1], gy[e + 1]); }
      }
    } else if (st == 20) {
    } else if (st == 21) {
      *reinterpret_cast<uint4*>(smem + pb[j & 1][2 * ksp] + (off + (j >> 1) * 8192)) = make_uint4(gh[0], gh[1], gh[2], gh[3]);
      if constexpr (PROD == 3) {      // (two products: the next GEMM reads the high terms only)
#pragma unroll
        for (int e = 0; e < 8; e += 2) gm[e >> 1] = pack16<true>(gt[e], gt[e + 1]);
        *reinterpret_cast<uint4*>(smem + pb[j & 1][2 * ksp + 1] + (off + (j >> 1) * 8192)) = make_uint4(gm[0], gm[1], gm[2], gm[3]);
      }
    }
  };

  // GEMM 1's weight ring is requested now: its latency hides behind the message publish
  const __amdgpu_buffer_rsrc_t w1b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w1s) + (size_t)(NI * wave) * 32 * 2 * 512, 0, NI * 32 * 2 * 1024, 0x00020000);
  constexpr int RA = 4;     // ring slots (8 KB per wave each); k-step n2 uses slot n2 % RA and refills slot (n2 - 1) % RA
  f16x8 fa[RA][NI][2];
  // GEMM 1 visits the k-tiles in the order 8..15 (message), 0..7 (x): sequence number n2 (k-steps) -> weight k-step (n2 + 16) & 31
  auto load_a = [&](int slot, int n2, int i, int pl) __attribute__((always_inline)) {
    fa[slot][i][pl] = ldw(w1b, ((i * 32 + (COMP ? n2 : ((n2 + 16) & 31))) * 2 + pl) * 1024);
  };
  constexpr int B1OFF = COMP ? 0 : RING + KT;      // where b1 / s1 waits for GEMM 1 (COMP: the message region is free; else ring slot 1, free until x tile 9 is staged)
  if constexpr (COMP) {
    // ---------------------------------------------------------------- prologue of the composed form: tiles 0, 1 (, 2), the whole weight ring, b1
#pragma unroll
    for (int q = 0; q < 4; ++q) stage_load(0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) stage_load(1, q);
#pragma unroll
    for (int q = 0; q < RA; ++q)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) load_a(q, q, i, pl);
    if (tid < 128) *reinterpret_cast<f32x4*>(smem + B1OFF + 16 * tid) = b1q * (1.0f / s1);
#pragma unroll
    for (int q = 0; q < 4; ++q) stage_write(0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) stage_load(2, q);
    __syncthreads();
    stamp(1); stamp(2); stamp(3);
  } else {
  // ================================================================ GEMM 0 (transposed): Msg^T[256][128] = Wo[256][256] . Ctx^T; wave w: features 64 w ..
  const __amdgpu_buffer_rsrc_t wob = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.wos) + (size_t)(NO * wave) * 16 * 2 * 512, 0, NO * 16 * 2 * 1024, 0x00020000);      // output tile NO w + o: 1 KB block ((tile * 16 + kstep) * 2 + term)
  constexpr int RO = 6;      // ring slots: k-step kk uses slot kk % RO and refills the slot k-step kk - 1 has freed (loads may then sit anywhere in the k-step)
  f16x8 go[RO][NO][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) stage_load(0, q);
#pragma unroll
  for (int q = 0; q < 4; ++q) stage_load(1, q);
#pragma unroll
  for (int q = 0; q < RO; ++q)
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) go[q][o][pl] = ldw(wob, ((o * 16 + q) * 2 + pl) * 1024);
  f32x16 acc0[NO][NJ];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[o][j][r] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) stage_write(0, q);
#pragma unroll
  for (int q = 0; q < 4; ++q) stage_load(2, q);
  __syncthreads();
  stamp(1);
  f32x4 bo4[NO][4];    // out_proj bias of the wave's features, requested during the last k-steps
  read_b(0, RING, 0, 0);
  read_b(1, RING, 0, 1);
  GN_PIN();
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int s = kk >> 1, ks = kk & 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int g = 4 * kk + j, g2 = g + 2, kk2 = g2 >> 2;
      if (kk2 < 16) read_b(g2, RING + ((kk2 >> 1) & 1) * KT, kk2 & 1, g2 & 3);
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          const int m = 6 * j + 2 * p + o;
          acc0[o][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(go[kk % RO][o][p == 0 ? 1 : 0], bq[g % 3][p == 1 ? 1 : 0], acc0[o][j], 0, 0, 0);
          if (ks == 0 && m % 6 == 1) stage_write(s + 1, m / 6);          // tile 8 = the first x k-tile
          if (!(ABL & 1) && m % 6 == 3 && kk >= 1 && kk + RO - 1 < 16) {
            const int lo = (m / 6) >> 1, lpl = (m / 6) & 1;
            go[(kk - 1) % RO][lo][lpl] = ldw(wob, ((lo * 16 + kk + RO - 1) * 2 + lpl) * 1024);
          }
          if (!(ABL & 2) && ks == 0 && m % 6 == 5 && s + 3 <= 8) stage_load(s + 3, m / 6);
          if (kk == 12 && m % 6 == 5)
#pragma unroll
            for (int g4 = 0; g4 < 2; ++g4) {
              const int u = 2 * (m / 6) + g4;      // 0..7 -> (o, g)
              bo4[u >> 2][u & 3] = *reinterpret_cast<const f32x4*>(a.bo + 32 * (NO * wave + (u >> 2)) + 8 * (u & 3) + 4 * hh);
            }
          GN_PIN();
        }
    }
    if (!(ABL & 16)) __syncthreads();     // after k-step 0: tile s + 1 is visible; after k-step 1: tile s's slot may be overwritten
    GN_PIN();
  }
  stamp(2);

#pragma unroll
  for (int q = 0; q < RA - 2; ++q)      // (the other two slots are requested after the publish: registers)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) load_a(q, q, i, pl);
  // ffn.0's bias, as b1 / s1, travels through ring slot 1 (free from here until x tile 9 is staged): requested in the prologue, readable after the
  // message barrier -- GEMM 1's accumulators start from it
  if (tid < 128) *reinterpret_cast<f32x4*>(smem + B1OFF + 16 * tid) = b1q * (1.0f / s1);
  // message = acc0 * scale + bias: feature 32 (NO w + o) + 8 g + 4 hh + c in register 4 g + c -> k-tile NO w + o of the message tile
  window(0);
#pragma unroll
  for (int jp = 0; jp < 2; ++jp)
#pragma unroll
    for (int c = 0; c < 4; ++c) pb[jp][c] = bo[0][jp][c] + (unsigned int)(NO * wave * KT);
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x2v y[8], t[8];
      unsigned int hw[8], mw[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) y[k] = pair(acc0[o][j], 2 * k) * splat2(a.wo_scale) + (f32x2v){bo4[o][k >> 1][2 * (k & 1)], bo4[o][k >> 1][2 * (k & 1) + 1]};
#pragma unroll
      for (int k = 0; k < 8; ++k) { amax = fmaxf(amax, fmaxf(fabsf(y[k][0]), fabsf(y[k][1]))); hw[k] = __builtin_bit_cast(unsigned int, __builtin_convertvector(y[k], f16x2v)); }
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = (f32x2v){resid_lo(hw[k], y[k][0]), resid_hi(hw[k], y[k][1])};
#pragma unroll
      for (int k = 0; k < 8; ++k) mw[k] = __builtin_bit_cast(unsigned int, __builtin_convertvector(t[k], f16x2v));
#pragma unroll
      for (int ksp = 0; ksp < 2; ++ksp) {
        *reinterpret_cast<uint4*>(smem + pb[j & 1][2 * ksp] + (o * KT + (j >> 1) * 8192)) = make_uint4(hw[4 * ksp], hw[4 * ksp + 1], hw[4 * ksp + 2], hw[4 * ksp + 3]);
        *reinterpret_cast<uint4*>(smem + pb[j & 1][2 * ksp + 1] + (o * KT + (j >> 1) * 8192)) = make_uint4(mw[4 * ksp], mw[4 * ksp + 1], mw[4 * ksp + 2], mw[4 * ksp + 3]);
      }
      GN_PIN();
    }
  __syncthreads();     // message tile complete
  stamp(3);

  }

  // ================================================================ GEMM 1 (transposed): H^T[512][128] = W1[512][512] . [x | msg]^T; wave w: hidden units 128 w ..
  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    f32x4 bi[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bi[g] = *reinterpret_cast<const f32x4*>(smem + B1OFF + (32 * (NI * wave + i) + 8 * g + 4 * hh) * 4);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j][4 * g + c] = bi[g][c];
    GN_PIN();
  }
  if constexpr (!COMP) {
#pragma unroll
    for (int q = RA - 2; q < RA; ++q)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) load_a(q, q, i, pl);
  }
  uint4 cst_a;      // the thread's share of the LayerNorm weight / bias arrays on their way to LDS
  read_b(0, COMP ? RING : 0, 0, 0);
  GN_PIN();
  auto gemm1_kstep = [&](int n2) __attribute__((always_inline)) {       // k-tile n = n2 >> 1: n < 8: message k-tile n; n >= 8: staged tile n (x k-tile n - 8) in ring slot n & 1
    const int n = n2 >> 1, ks = n2 & 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int g = 4 * n2 + j, g1 = g + 1, m2 = g1 >> 2, nn = m2 >> 1;
      if (m2 < 32 && !((ABL & 64) && n2 > 0)) read_b(g1, (!COMP && nn < 8) ? nn * KT : RING + (nn & 1) * KT, m2 & 1, g1 & 3);
      // products: W_m X_h, W_h X_m, W_h X_h (small terms first), the four hidden tiles round-robin
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          if (PROD == 2 && p == 1) continue;            // (W_h X_m: not in the two-product form)
          const int mj = 4 * p + i;       // MFMA number inside the j-step
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[n2 % RA][i][p == 0 ? 1 : 0], bq[g % 3][p == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
          // the memory instructions of the k-step, one per MFMA gap: three products: gaps 2 | 5, 10 | 7 of the j-step's 12; two products: 2 | 1, 9 | 3 of its 8 (mj 4..7 do not exist)
          constexpr int kLa0 = PROD == 3 ? 5 : 1, kLa1 = PROD == 3 ? 10 : 9, kSl = PROD == 3 ? 7 : 3;
          if ((COMP || n >= 8) && ks == 0 && n + 1 < 16 && mj == 2) stage_write(n + 1, j);
          if (!(ABL & 1) && (mj == kLa0 || mj == kLa1) && n2 >= 1 && n2 + RA - 1 < 32) {
            const int u = 2 * j + (mj == kLa1 ? 1 : 0);   // 0..7 -> (i, term)
            load_a((n2 - 1) % RA, n2 + RA - 1, u >> 1, u & 1);
          }
          if (!(ABL & 2) && ks == 0 && (COMP || n + 3 >= 9) && n + 3 < 16 && mj == kSl) stage_load(n + 3, j);     // (x tiles 9, 10 are requested during the last message k-tiles)
          if (n2 == 29 && mj == kSl && j == 0) cst_a = *reinterpret_cast<const uint4*>((tid < 128 ? a.ln_g : a.ln_b - 512) + 4 * tid);
          GN_PIN();
        }
    }
    if ((COMP || n >= 8) && !(ABL & 16)) __syncthreads();
    if (n2 == 15) stamp(10);
    if ((ABL & 128) && ks == 1) ts2[n] = (long long)__builtin_amdgcn_s_memtime();
    GN_PIN();
  };
  if constexpr (COMP) {
#pragma unroll
    for (int n2 = 0; n2 < 32; ++n2) gemm1_kstep(n2);
  } else {
#pragma unroll
    for (int n2 = 0; n2 < 7; ++n2) gemm1_kstep(n2);
    window(1);        // message k-tiles 4..7 (first use: the prefetch out of k-step 7)
#pragma unroll
    for (int n2 = 7; n2 < 11; ++n2) gemm1_kstep(n2);
    stage_addr();     // (first use: the loads of x tile 9 at k-tile 6)
#pragma unroll
    for (int n2 = 11; n2 < 15; ++n2) gemm1_kstep(n2);
    window(2);        // the x half reads the ring (first use: the prefetch out of k-step 15)
#pragma unroll
    for (int n2 = 15; n2 < 32; ++n2) gemm1_kstep(n2);
  }
  stamp(4);

  // wave w: output features 64 w ..; k-steps are visited quarter by quarter: sequence number c = 8 q + 2 w' + ks -> weight k-step 8 w' + 2 q + ks
  const __amdgpu_buffer_rsrc_t w2b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w2s) + (size_t)(NO * wave) * 32 * 2 * 512, 0, NO * 32 * 2 * 1024, 0x00020000);        // output tile NO w + o: 1 KB block ((tile * 32 + kstep) * 2 + term)
  constexpr int RG = 5;      // W2 k-tiles in flight.  (Six cost 16 more registers: the LayerNorm constants of the last two quarters then went to scratch and their
                             // reload in the middle of GEMM 2 drained the ring: 164 -> 162 us with five, same bits.)
  f16x8 ga[RG][NO][2];
  auto load_g = [&](int slot, int c, int o, int pl) __attribute__((always_inline)) {
    const int kstep = 8 * ((c >> 1) & 3) + 2 * (c >> 3) + (c & 1);
    ga[slot][o][pl] = ldw(w2b, ((o * 32 + kstep) * 2 + pl) * 1024);
  };
  // constants -> LDS (the ring is dead: every wave is past GEMM 1's last barrier)
  *reinterpret_cast<uint4*>(smem + CST + 16 * tid) = cst_a;                       // ln_g[0..511], ln_b[0..511]

  // ---------------------------------------------------------------- LayerNorm(512) statistics, eps 1e-5 -- read-only over the accumulators
  // lane (ql, hh) holds, for tokens 32 j + ql, the hidden units  32 (NI w + i) + 8 g + 4 hh + c   (register r = 4 g + c): 64 values per
  // token.  Shifted sums (shift = the lane's first value) -> (mean, M2) of the 64; Chan's merge with the other half-wave, then across
  // the 4 waves through LDS: one exchange, no catastrophic cancellation whatever the mean.
  {
    float* const stat = reinterpret_cast<float*>(smem + STAT);     // [NW][TM][2]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float K = acc[0][j][0];      // statistics in accumulator units (times s1 below)
      f32x2v sd = splat2(0.f), sq = splat2(0.f);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2v d = pair(acc[i][j], r) - splat2(K);
          sd += d;
          sq += d * d;
        }
        GN_PIN();      // (left alone the compiler copies all 256 accumulators into VGPRs first, and spills)
      }
      const float sdl = sd[0] + sd[1], sql = sq[0] + sq[1];
      const float mu = sdl * (1.0f / 64.0f);
      float mean_a = K + mu, m2_a = sql - sdl * mu;
      const float mean_b = __shfl_xor(mean_a, 32), m2_b = __shfl_xor(m2_a, 32);
      const float dl = mean_a - mean_b;
      m2_a = m2_a + m2_b + 32.0f * dl * dl;
      mean_a = 0.5f * (mean_a + mean_b);
      if (hh == 0) *reinterpret_cast<f32x2v*>(stat + ((wave * TM + 32 * j + ql) * 2)) = (f32x2v){mean_a, m2_a};
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x2v pw[NW];
#pragma unroll
      for (int w4 = 0; w4 < NW; ++w4) pw[w4] = *reinterpret_cast<const f32x2v*>(stat + ((w4 * TM + 32 * j + ql) * 2));
      const float mean = 0.25f * ((pw[0][0] + pw[1][0]) + (pw[2][0] + pw[3][0]));
      float m2 = (pw[0][1] + pw[1][1]) + (pw[2][1] + pw[3][1]);
#pragma unroll
      for (int w4 = 0; w4 < NW; ++w4) { const float dl = pw[w4][0] - mean; m2 += 128.0f * dl * dl; }
      const float rstd = 1.0f / sqrtf(m2 * (s1 * s1) * (1.0f / 512.0f) + 1e-5f);
      rs[j] = rstd * s1;
      nmr[j] = -(mean * s1) * rstd;
    }
  }
  stamp(5);
  // the second GEMM's first weight fragments: their latency hides behind the two exposed GELU quarters
#pragma unroll
  for (int q = 0; q < RG; ++q)
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) load_g(q, q, o, pl);

  // ---------------------------------------------------------------- normalise + erf GELU + split + publish of one fragment, as 22 stages
  // wave w's hidden units 32 (NI w + q) .. of quarter q -> k-tile w of buffer q & 1 (the message tile's space: GEMM 1 is done with it)
  auto gelu_stage = [&](int st, int q, int j, int ksp) __attribute__((always_inline)) {
    if (st == 0) {
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int cof = (32 * q + 8 * (2 * ksp + gg)) * 4;
        c2[gg] = *reinterpret_cast<const f32x4*>(smem + cbase + cof);
        c3[gg] = *reinterpret_cast<const f32x4*>(smem + cbase + cof + 2048);
      }
    } else if (st == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gy[e] = acc[q][j][8 * ksp + e] * rs[j] + nmr[j];
    } else if (st == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gy[e] = gy[e] * c2[e >> 2][e & 3] + c3[e >> 2][e & 3];
    } else if (ABL & 4) {
      if (st >= 18) tail_stage(st, 0, j, ksp);
    } else if (st == 3) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gt[e] = gy[e] * 0.70710678118654752440f;
    } else if (st == 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gt[e] = fminf(fabsf(gt[e]), 4.0f);
    } else if (st == 5) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gq[e] = 4.6081331674940884e-05f * gt[e] + -0.00045161080197431147f;
    } else if (st >= 6 && st <= 11) {
      const float cf = st == 6 ? 0.0015096671413630247f : st == 7 ? 0.0007409505778923631f : st == 8 ? -0.028223754838109016f
                     : st == 9 ? 0.1484677642583847f : st == 10 ? 0.918419361114502f : 1.6279083490371704f;
#pragma unroll
      for (int e = 0; e < 8; ++e) gq[e] = gq[e] * gt[e] + cf;
    } else if (st == 12) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gq[e] = gq[e] * gt[e];
    } else if (st == 13) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gq[e] = __builtin_amdgcn_exp2f(-gq[e]);
    } else if (st == 14) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gq[e] = 1.0f - gq[e];
    } else if (st == 15) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gq[e] = copysignf(gq[e], gy[e]);
    } else if (st == 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gt[e] = 0.5f * gy[e];
    } else if (st == 17) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gy[e] = gt[e] * gq[e] + gt[e];
    } else if (st >= 18) {
      tail_stage(st, 0, j, ksp);
    }
  };
  auto set_pb = [&](int q) __attribute__((always_inline)) {      // quarter q goes to k-tile `wave` of buffer q & 1 (64 KB each)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int c = 0; c < 4; ++c) pb[jp][c] = bo[q & 1][jp][c] + (unsigned int)(wave * KT);
  };
  window(0);
  window(1);
  // quarters 0 and 1 before the second GEMM starts (its 128 accumulators and the 256 of GEMM 1 do not fit the accumulator file together)
  // (no MFMA runs beside this code: two-wide v_pk_* arithmetic halves its instruction count -- one wave issues a VALU instruction
  // every ~6 cycles whatever its width (tools/probes/stream1w.hip) -- while beside MFMAs a packed instruction costs what two
  // scalar ones do; the staged scalar form above is for the quarters produced inside the second GEMM)
  auto gelu_tile_pk = [&](int q, int j) __attribute__((always_inline)) {
    f32x4 w4[4], b4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      w4[g] = *reinterpret_cast<const f32x4*>(smem + cbase + (32 * q + 8 * g) * 4);
      b4[g] = *reinterpret_cast<const f32x4*>(smem + cbase + (32 * q + 8 * g) * 4 + 2048);
    }
    f32x2v y[8], t[8], u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = pair(acc[q][j], 2 * k) * splat2(rs[j]) + splat2(nmr[j]);
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = y[k] * (f32x2v){w4[k >> 1][2 * (k & 1)], w4[k >> 1][2 * (k & 1) + 1]} + (f32x2v){b4[k >> 1][2 * (k & 1)], b4[k >> 1][2 * (k & 1) + 1]};
    if (!(ABL & 4)) {
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = y[k] * splat2(0.70710678118654752440f);
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = (f32x2v){fminf(fabsf(t[k][0]), 4.0f), fminf(fabsf(t[k][1]), 4.0f)};
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = splat2(4.6081331674940884e-05f) * t[k] + splat2(-0.00045161080197431147f);
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        const float cf = st == 0 ? 0.0015096671413630247f : st == 1 ? 0.0007409505778923631f : st == 2 ? -0.028223754838109016f
                       : st == 3 ? 0.1484677642583847f : st == 4 ? 0.918419361114502f : 1.6279083490371704f;
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = u[k] * t[k] + splat2(cf);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = u[k] * t[k];
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = (f32x2v){__builtin_amdgcn_exp2f(-u[k][0]), __builtin_amdgcn_exp2f(-u[k][1])};
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = splat2(1.0f) - u[k];
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = (f32x2v){copysignf(u[k][0], y[k][0]), copysignf(u[k][1], y[k][1])};
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = splat2(0.5f) * y[k];
#pragma unroll
      for (int k = 0; k < 8; ++k) y[k] = t[k] * u[k] + t[k];
    }
    // maximum, fp16 split, publish: registers 0..7 -> the fragment of k-step 0, 8..15 -> k-step 1
    unsigned int hw[8], mw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { amax = fmaxf(amax, fmaxf(fabsf(y[k][0]), fabsf(y[k][1]))); hw[k] = __builtin_bit_cast(unsigned int, __builtin_convertvector(y[k], f16x2v)); }
    if constexpr (PROD == 3) {
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = (f32x2v){resid_lo(hw[k], y[k][0]), resid_hi(hw[k], y[k][1])};
#pragma unroll
      for (int k = 0; k < 8; ++k) mw[k] = __builtin_bit_cast(unsigned int, __builtin_convertvector(t[k], f16x2v));
    }
#pragma unroll
    for (int ksp = 0; ksp < 2; ++ksp) {
      *reinterpret_cast<uint4*>(smem + pb[j & 1][2 * ksp] + ((j >> 1) * 8192)) = make_uint4(hw[4 * ksp], hw[4 * ksp + 1], hw[4 * ksp + 2], hw[4 * ksp + 3]);
      if constexpr (PROD == 3)
        *reinterpret_cast<uint4*>(smem + pb[j & 1][2 * ksp + 1] + ((j >> 1) * 8192)) = make_uint4(mw[4 * ksp], mw[4 * ksp + 1], mw[4 * ksp + 2], mw[4 * ksp + 3]);
    }
  };
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    set_pb(q);
#pragma unroll
    for (int j = 0; j < NJ; ++j) { gelu_tile_pk(q, j); GN_PIN(); }
  }
  __syncthreads();     // quarters 0 and 1 visible; every wave is past its LayerNorm statistics
  stamp(6);

  // residual rows of the epilogue: lane -> (row parity lane >> 5, feature octet lane & 31): a wave finishes two 1 KB rows per step with
  // 16-byte accesses; the first half of the rows is requested inside the last quarter of GEMM 2 (whose VALU stages are idle)
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const int oct = lane & 31, f0 = 8 * oct;
  constexpr int RW = TM / NW;   // token rows per wave
  const unsigned int roff = (unsigned int)(hh * 1024 + (oct >> 1) * 64 + (oct & 1) * 16);
  uint4 rh[RW / 2], rm[RW / 2];
  auto res_load = [&](int it) __attribute__((always_inline)) {
    rh[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, roff, (RW * wave + 2 * it) * 1024, 0));
    rm[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, roff + 32, (RW * wave + 2 * it) * 1024, 0));
  };
  // ================================================================ GEMM 2 (transposed): Y^T[256][128] = W2[256][512] . H^T;  wave w: output features 64 w ..
  f32x16 acc2[NO][NJ];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[o][j][r] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int buf = (q & 1) * 4 * KT;
    const bool fill = q >= 1 && q + 1 < 4;       // quarter q + 1 is produced (into the buffer quarter q - 1 has released) while quarter q is consumed
    if (fill) set_pb(q + 1);
    read_b(32 * q, buf, 0, 0);
    read_b(32 * q + 1, buf, 0, 1);
    GN_PIN();
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int c = 8 * q + cc, w4 = cc >> 1, ks = cc & 1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int g = 4 * c + j, g2 = g + 2, c2_ = g2 >> 2;
        if ((c2_ >> 3) == q) read_b(g2, buf + ((c2_ >> 1) & 3) * KT, c2_ & 1, g2 & 3);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int o = 0; o < NO; ++o) {
            if (PROD == 2 && p == 1) continue;            // (W_h H_m: not in the two-product form)
            acc2[o][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[c % RG][o][p == 0 ? 1 : 0], bq[g % 3][p == 1 ? 1 : 0], acc2[o][j], 0, 0, 0);
            if constexpr (PROD == 3) {
              const int m = 6 * j + 2 * p + o;
              // the VALU stage of the next quarter's fragment (token tile w4, k-step ks) that shares this MFMA's gap
              if (fill && m < 22) gelu_stage(m, q + 1, w4, ks);
              if (q == 3 && m == 2 && !LOOP) res_load(cc);      // (walking form: the registers are not there, all residual rows are requested in the epilogue)
              if (!(ABL & 1) && m % 6 == 3 && c >= 1 && c + RG - 1 < 32) load_g((c - 1) % RG, c + RG - 1, (m / 6) >> 1, (m / 6) & 1);
            } else {
              // sixteen MFMA gaps per (quarter, k-step) for the fragment's twenty stages (no residual split): the light neighbours share a gap
              const int m = 4 * j + (p == 2 ? 2 : 0) + o;
              if (fill) {
                constexpr int first[16] = {0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 18};
                constexpr int second[16] = {-1, -1, -1, 4, -1, -1, -1, -1, -1, -1, -1, -1, -1, 15, 17, 21};
                gelu_stage(first[m], q + 1, w4, ks);
                if (second[m] >= 0) gelu_stage(second[m], q + 1, w4, ks);
              }
              if (q == 3 && m == 2 && !LOOP) res_load(cc);
              if (!(ABL & 1) && m % 4 == 3 && c >= 1 && c + RG - 1 < 32) load_g((c - 1) % RG, c + RG - 1, (m / 4) >> 1, (m / 4) & 1);
            }
            GN_PIN();
          }
      }
    }
    __syncthreads();     // quarter q + 1 visible, buffer q & 1 free again
    if (ABL & 128) ts2[q] = (long long)__builtin_amdgcn_s_memtime();      // (overwrites the first GEMM 1 stamps: tools/ffn128_ab.py reads both)
    GN_PIN();
  }
  ovf_commit(a.ovf, amax);
  stamp(7);

  // ---------------------------------------------------------------- epilogue: + bias + residual x, hm16 (and optionally f32) rows
  // (the hidden buffers are dead: their space becomes the [128 tokens][256 features] f32 tile of the row-wise epilogue)
  float* const yt = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int it = LOOP ? 0 : RW / 4; it < RW / 2; ++it) res_load(it);     // second half of the residual rows (the first was requested during quarter 3)
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc2[o][j][4 * g], acc2[o][j][4 * g + 1], acc2[o][j][4 * g + 2], acc2[o][j][4 * g + 3]};
        *reinterpret_cast<f32x4*>(yt + (32 * j + ql) * YP + 32 * (NO * wave + o) + 8 * g + 4 * hh) = v;
      }
      GN_PIN();
    }
  const float s2 = a.w2_scale;
  const f32x4 bias_a = *reinterpret_cast<const f32x4*>(a.b2 + f0), bias_b = *reinterpret_cast<const f32x4*>(a.b2 + f0 + 4);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.yp + (size_t)bm * 512, 0, TM * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(a.y != nullptr ? a.y + (size_t)bm * kDim : reinterpret_cast<float*>(a.yp), 0, a.y != nullptr ? TM * 1024 : 0, 0x00020000);
  const unsigned int foff = (unsigned int)(hh * 1024 + f0 * 4);
  __syncthreads();
  stamp(8);
  float amax2 = 0.f;
  uint4 xh[QKV != 0 ? RW / 2 : 1];
#pragma unroll
  for (int it = 0; it < RW / 2; ++it) {
    const int rowl = RW * wave + 2 * it;      // (+ hh: in the lane offsets)
    const f32x4 ya = *reinterpret_cast<const f32x4*>(yt + (rowl + hh) * YP + f0), yb = *reinterpret_cast<const f32x4*>(yt + (rowl + hh) * YP + f0 + 4);
    const unsigned int xhw[4] = {rh[it].x, rh[it].y, rh[it].z, rh[it].w}, xmw[4] = {rm[it].x, rm[it].y, rm[it].z, rm[it].w};
    f32x2v v[4], t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x2v yk = k < 2 ? (f32x2v){ya[2 * k], ya[2 * k + 1]} : (f32x2v){yb[2 * k - 4], yb[2 * k - 3]};
      const f32x2v bk = k < 2 ? (f32x2v){bias_a[2 * k], bias_a[2 * k + 1]} : (f32x2v){bias_b[2 * k - 4], bias_b[2 * k - 3]};
      const f32x2v f = yk * splat2(s2) + bk;
      // + x = + x_m + x_h, the small term first (v_fma_mix_f32 converts the fp16 operand inside the instruction)
      v[k] = (f32x2v){addh_lo(xhw[k], addh_lo(xmw[k], f[0])), addh_hi(xhw[k], addh_hi(xmw[k], f[1]))};
    }
    unsigned int hw[4], mw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { amax2 = fmaxf(amax2, fmaxf(fabsf(v[k][0]), fabsf(v[k][1]))); hw[k] = __builtin_bit_cast(unsigned int, __builtin_convertvector(v[k], f16x2v)); }
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = (f32x2v){resid_lo(hw[k], v[k][0]), resid_hi(hw[k], v[k][1])};
#pragma unroll
    for (int k = 0; k < 4; ++k) mw[k] = __builtin_bit_cast(unsigned int, __builtin_convertvector(t[k], f16x2v));
    __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){hw[0], hw[1], hw[2], hw[3]}, yrs, roff, rowl * 1024, 0);
    __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){mw[0], mw[1], mw[2], mw[3]}, yrs, roff + 32, rowl * 1024, 0);
    if constexpr (QKV != 0) xh[it] = make_uint4(hw[0], hw[1], hw[2], hw[3]);     // the fp16 HIGH terms of the new rows: the fused projection's token operand
    if (a.y != nullptr) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (f32x4){v[0][0], v[0][1], v[1][0], v[1][1]}), frs, foff, rowl * 1024, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (f32x4){v[2][0], v[2][1], v[3][0], v[3][1]}), frs, foff + 16, rowl * 1024, 0);
    }
    GN_PIN();
  }
  ovf_commit(a.ovf, amax2);
  if constexpr (QKV != 0) {
    // ================================================================ the NEXT block's attention input projection on this tile's new rows (round 5)
    // k_qkv<CROSS, true, 2> (gn_qkv.hip) with 4 waves instead of 8: the same two partial products per block (W_m X_h, then W_h X_h) in the same k
    // order, the same epilogue expressions -> the same bits.  Token operand = the fp16 HIGH terms the row-wise epilogue just produced (xh[]), written
    // into the k-tile layout the other GEMMs of this kernel read; weights register-fed through a ring; wave w owns feature tiles 8 pass + 2 w + {0, 1}
    // of every pass (64 features x 128 tokens = 128 accumulator registers), i.e. one head of q, of k and of v.
    constexpr bool CROSS = QKV == 2;
    constexpr int NPASS = CROSS ? 2 : 3;
    constexpr int NQK = CROSS ? kDim : 2 * kDim;      // q | k (or qk) features = the pitch of the fp16 rows
    constexpr int ROT = 8 * KT;                       // [16 feature groups][128 tokens] f32x4 rotary entries behind the token tile (32 KB)
    constexpr int RQ = 4;                             // weight k-steps in flight (4 KB per wave each)
    const int slot = bm / a.npad, i0 = bm - slot * a.npad;
    f32x4 rq[CROSS ? 1 : 8];
    if constexpr (!CROSS) {      // the tile's rotary entries: thread -> entries tid + 256 k of [fg][token] (consecutive threads = consecutive tokens)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = tid + 256 * k;
        rq[k] = reinterpret_cast<const f32x4*>(a.q_rot4)[(size_t)(e >> 7) * a.q_rot_stride + (size_t)(bm + (e & 127))];
      }
    }
    __syncthreads();          // every wave has read its rows of the f32 tile: the space becomes the projection's token tile
    {
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      const int oc = l_ & 31, rp = l_ >> 5;
      const int ch = 4 * ((oc >> 1) & 1) + (oc & 1);       // chunk of the k-tile's 128-byte row segment: 4 (k-step & 1) + 2 term + half
#pragma unroll
      for (int it = 0; it < RW / 2; ++it) {
        const int row = RW * wave + 2 * it + rp;
        *reinterpret_cast<uint4*>(smem + (oc >> 2) * KT + row * 128 + ((ch ^ swz(row)) * 16)) = xh[it];
      }
    }
    if constexpr (!CROSS) {
#pragma unroll
      for (int k = 0; k < 8; ++k) *reinterpret_cast<f32x4*>(smem + ROT + (tid + 256 * k) * 16) = rq[k];
    }
    window(0);
    window(1);
    f16x8 qa[RQ][2][2];
    f16x8 qb[3][2];          // token fragments of one pair of token tiles (j = 2 jp, 2 jp + 1), high terms only
    auto read_q = [&](int gp) __attribute__((always_inline)) {      // pair-step gp = 2 ks + jp
      const int ks = gp >> 1, jp = gp & 1;
      const int off = (ks >> 1) * KT + jp * 8192;
#pragma unroll
      for (int jl = 0; jl < 2; ++jl) qb[gp % 3][jl] = *reinterpret_cast<const f16x8*>(smem + bo[off >> 16][jl][2 * (ks & 1)] + (off & 65535));
    };
    const float ascale = a.q_acc_scale;
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(a.q_qkb + (size_t)bm * NQK, 0, TM * NQK * 2, 0x00020000);
    float amaxq = 0.f;
    __syncthreads();          // token tile (and rotary entries) visible
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const bool vpass = pass == NPASS - 1;
      const __amdgpu_buffer_rsrc_t wq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.q_wf) + (size_t)(8 * pass + 2 * wave) * 16 * 2 * 512, 0, 2 * 16 * 2 * 1024, 0x00020000);
      auto load_q = [&](int slot_, int ks, int i2, int pl) __attribute__((always_inline)) { qa[slot_][i2][pl] = ldw(wq, ((i2 * 16 + ks) * 2 + pl) * 1024); };
#pragma unroll
      for (int q = 0; q < RQ; ++q)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) load_q(q, q, i2, pl);
      f32x4 bias4[2][4];
      if (!vpass) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int g = 0; g < 4; ++g) bias4[i2][g] = *reinterpret_cast<const f32x4*>(a.q_bias + 32 * (8 * pass + 2 * wave + i2) + 8 * g + 4 * hh);
      }
      f32x16 qacc[2][NJ];
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) qacc[i2][j][r] = 0.f;
      read_q(0);
      read_q(1);
      GN_PIN();
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int gp = 2 * ks + jp;
          if (gp + 2 < 32) read_q(gp + 2);
          // products: W_m X_h, then W_h X_h (k_qkv's order per accumulator); the four accumulators of the pair round-robin
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl)
#pragma unroll
              for (int i2 = 0; i2 < 2; ++i2) {
                const int j = 2 * jp + jl, m = 8 * jp + 4 * p + 2 * jl + i2;
                const f16x8 w = qa[ks % RQ][i2][p == 0 ? 1 : 0], x = qb[gp % 3][jl];
                qacc[i2][j] = vpass ? __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, qacc[i2][j], 0, 0, 0)     // rows = tokens, columns = features
                                    : __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, qacc[i2][j], 0, 0, 0);    // rows = features, columns = tokens
                if (m % 4 == 1 && ks >= 1 && ks + RQ - 1 < 16) load_q((ks - 1) % RQ, ks + RQ - 1, (m / 4) >> 1, (m / 4) & 1);
                GN_PIN();
              }
        }
      }
      if (!vpass) {
        // register r of tile (i2, j) <-> feature 32 tile + (r & 3) + 8 (r >> 2) + 4 hh, token 32 j + ql (k_qkv's q / k epilogue, expression for expression)
        const unsigned int qlo = (unsigned int)((ql * NQK + 8 * hh) * 2);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const int tile = 8 * pass + 2 * wave + i2;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            uint2 pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 v = {qacc[i2][j][4 * g], qacc[i2][j][4 * g + 1], qacc[i2][j][4 * g + 2], qacc[i2][j][4 * g + 3]};
              // SCALAR f32 arithmetic, component by component, each result pinned: written on the f32x4 vectors (k_qkv's source) hipcc emits
              // `v_pk_mul_f32 D, s[n:n+1], V op_sel_hi:[0,1]` for the scale factors, and in THIS kernel's company (one-tile form, self block) the
              // q rows then came back with garbage in (o.z, o.w) of two of the four feature groups in lanes 12..15 / 28..31 -- deterministic, moving
              // with every unrelated edit (K before Q: gone; a barrier in front: other positions), the LDS rotary entries and the accumulators
              // verified intact (tools/dbg_fused_qkv.py).  The same class as build.py's note on v_pk_fma_f32 in k_qkv.  The
              // products and their order are unchanged: (acc * s + b), the rotation's mul + fma pairs, * qscale -- same bits as k_qkv.
              v.x = v.x * ascale; v.y = v.y * ascale; v.z = v.z * ascale; v.w = v.w * ascale;
              asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
              v.x += bias4[i2][g].x; v.y += bias4[i2][g].y; v.z += bias4[i2][g].z; v.w += bias4[i2][g].w;
              asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
              if (CROSS) {
                v.x *= a.q_scale; v.y *= a.q_scale; v.z *= a.q_scale; v.w *= a.q_scale;
                asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
              } else {
                const f32x4 rot = *reinterpret_cast<const f32x4*>(smem + ROT + ((8 * i2 + 2 * g + hh) * 128 + 32 * j + ql) * 16);
                f32x4 o;
                o.x = v.x * rot.x + (-v.y) * rot.z;
                o.y = v.y * rot.x + v.x * rot.z;
                o.z = v.z * rot.y + (-v.w) * rot.w;
                o.w = v.w * rot.y + v.z * rot.w;
                v = o;
                if (pass == 0) { v.x *= a.q_qscale; v.y *= a.q_qscale; v.z *= a.q_qscale; v.w *= a.q_qscale; asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
              }
              pk[g].x = pack16<true>(v.x, v.y);
              pk[g].y = pack16<true>(v.z, v.w);
              ovf_track(amaxq, v.x, v.y); ovf_track(amaxq, v.z, v.w);
            }
            // the half-waves trade every other group of 4 features: a lane stores 8 consecutive features (16 bytes)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              const uint2 give = hh ? pk[2 * gp] : pk[2 * gp + 1];
              uint2 got;
              got.x = __shfl_xor(give.x, 32); got.y = __shfl_xor(give.y, 32);
              const uint2 own = hh ? pk[2 * gp + 1] : pk[2 * gp];
              const u32x4_t out = hh ? (u32x4_t){got.x, got.y, own.x, own.y} : (u32x4_t){own.x, own.y, got.x, got.y};
              __builtin_amdgcn_raw_buffer_store_b128(out, qrs, qlo, (32 * j * NQK + 32 * tile + 16 * gp) * 2, 0);
            }
          }
        }
      } else {
        // register r of tile (i2, j) <-> token 32 j + (r & 3) + 8 (r >> 2) + 4 hh, feature 32 (2 w + i2) + ql of the V panel = head w, dim 32 i2 + ql;
        // registers 8 m .. 8 m + 7 are the keys 16 m + 4 hh + {0..3, 8..11}: group 2 m + hh of the permuted V^T layout
        const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(a.q_vt + ((size_t)(slot * kHeads + wave) * kHeadDim) * a.npad + i0, 0, kHeadDim * a.npad * 2, 0x00020000);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const float bias = a.q_bias[NQK + 32 * (2 * wave + i2) + ql];
          const unsigned int vlo = (unsigned int)(((32 * i2 + ql) * a.npad + 8 * hh) * 2);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              unsigned int w4[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float lo = qacc[i2][j][8 * m + 2 * e] * ascale + bias;
                const float hi = qacc[i2][j][8 * m + 2 * e + 1] * ascale + bias;
                w4[e] = pack16<true>(lo, hi);
                ovf_track(amaxq, lo, hi);
              }
              __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){w4[0], w4[1], w4[2], w4[3]}, vrs, vlo, (32 * j + 16 * m) * 2, 0);
            }
        }
      }
      GN_PIN();
    }
    ovf_commit(a.ovf, amaxq);
  }
  if (ABL & 8) {
    stamp(9);
    if (a.dbg_ts != nullptr && lane == 0)
      for (int k = 0; k < 12; ++k) a.dbg_ts[((size_t)blockIdx.x * NW + wave) * 12 + k] = ts[k];
    if ((ABL & 128) && a.dbg_ts != nullptr && lane == 0)
      for (int k = 0; k < 16; ++k) a.dbg_ts[(size_t)gridDim.x * NW * 12 + ((size_t)blockIdx.x * NW + wave) * 16 + k] = ts2[k];
  }
  if (!LOOP) break;
  __syncthreads();     // every wave is done with the tile store before the next tile is staged
  }   // work
}
#undef GN_PIN
}  // namespace

void launch_ffn128(const FfnArgs& a, int ablate, hipStream_t s) {
  const int ncu = a.ncu > 0 ? a.ncu : device_cu_count();
  const bool walk = a.tiles != nullptr && a.walk && ablate == 0;
  const dim3 grid(walk ? std::min(a.T / 128, ncu) : a.T / 128), block(256);     // walking form: one workgroup per CU
  FfnArgs b = a;
  if (!walk) b.tiles = nullptr;
  if (walk || ablate != 0) b.nvalid = nullptr;
  if (a.composed && a.products == 2 && (ablate == 0 || ablate == 8)) {      // two partial products (round 6): composed form only
#define GN_F2(Q) do { if (ablate == 8) hipLaunchKernelGGL((k_ffn128<8, true, false, Q, 2>), grid, block, 0, s, b); \
                      else if (walk) hipLaunchKernelGGL((k_ffn128<0, true, true, Q, 2>), grid, block, 0, s, b); \
                      else hipLaunchKernelGGL((k_ffn128<0, true, false, Q, 2>), grid, block, 0, s, b); } while (0)
    const int qv = ablate == 0 ? a.qkv : 0;
    if (qv == 1) { GN_F2(1); g_last_kernel = walk ? "k_ffn128<0, true, true, 1, 2>" : "k_ffn128<0, true, false, 1, 2>"; }
    else if (qv == 2) { GN_F2(2); g_last_kernel = walk ? "k_ffn128<0, true, true, 2, 2>" : "k_ffn128<0, true, false, 2, 2>"; }
    else { GN_F2(0); g_last_kernel = walk ? "k_ffn128<0, true, true, 0, 2>" : "k_ffn128<0, true, false, 0, 2>"; }
#undef GN_F2
    return;
  }
  if (a.composed && a.qkv != 0 && ablate == 0) {      // the next block's attention input projection behind the tail (a.qkv: 1 self, 2 cross)
    if (a.qkv == 1) {
      if (walk) hipLaunchKernelGGL((k_ffn128<0, true, true, 1>), grid, block, 0, s, b); else hipLaunchKernelGGL((k_ffn128<0, true, false, 1>), grid, block, 0, s, b);
      g_last_kernel = walk ? "k_ffn128<0, true, true, 1, 3>" : "k_ffn128<0, true, false, 1, 3>";
    } else {
      if (walk) hipLaunchKernelGGL((k_ffn128<0, true, true, 2>), grid, block, 0, s, b); else hipLaunchKernelGGL((k_ffn128<0, true, false, 2>), grid, block, 0, s, b);
      g_last_kernel = walk ? "k_ffn128<0, true, true, 2, 3>" : "k_ffn128<0, true, false, 2, 3>";
    }
    return;
  }
  if (a.composed) {
    switch (ablate) {
      case 8: hipLaunchKernelGGL((k_ffn128<8, true>), grid, block, 0, s, b); break;
      case 136: hipLaunchKernelGGL((k_ffn128<136, true>), grid, block, 0, s, b); break;
      default: if (walk) hipLaunchKernelGGL((k_ffn128<0, true, true>), grid, block, 0, s, b); else hipLaunchKernelGGL((k_ffn128<0, true>), grid, block, 0, s, b); break;
    }
    g_last_kernel = walk ? "k_ffn128<0, true, true, 0, 3>" : "k_ffn128<0, true, false, 0, 3>";
    return;
  }
  switch (ablate) {
    case 8: hipLaunchKernelGGL((k_ffn128<8, false>), grid, block, 0, s, b); break;
    case 136: hipLaunchKernelGGL((k_ffn128<136, false>), grid, block, 0, s, b); break;
    default: if (walk) hipLaunchKernelGGL((k_ffn128<0, false, true>), grid, block, 0, s, b); else hipLaunchKernelGGL((k_ffn128<0, false>), grid, block, 0, s, b); break;
  }
  g_last_kernel = walk ? "k_ffn128<0, false, true, 0, 3>" : "k_ffn128<0, false, false, 0, 3>";
}

}  // namespace gn
