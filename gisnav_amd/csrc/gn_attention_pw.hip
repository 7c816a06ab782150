// Fused softmax(Q K^T) V, the bulk-grid kernel of round 4: ONE wave per SIMD, 64 queries per wave, instruction order pinned.
//
// Same arithmetic, operand layouts, LDS-DMA rings and lazy running maximum as k_attn16_v5 (gn_attention.hip; kornia's `Attention.forward`,
// reached from ros/gisnav/gisnav/core/pose_node.py:285-287) -- what changes is who issues what, when:
//   * k_attn16_v5 runs two 4-wave workgroups per CU (two waves per SIMD, 32 queries each) and leaves the order inside a key tile to the
//     compiler: score MFMAs, then the exponentials as a block, then the V^T P^T MFMAs.  Its matrix pipe is busy 0.34 of the time: a wave's VALU
//     work runs under MFMAs only when they are interleaved in ITS OWN instruction stream -- two waves of a SIMD do not overlap one's VALU with
//     the other's MFMAs (tools/probes/overlap.hip) -- and every K / V^T fragment read from LDS feeds one MFMA.
//   * here a wave owns 64 queries (two 32-query tiles): every fragment feeds two MFMAs, and the wave's single instruction stream is written out
//     MFMA by MFMA (sched_barrier after each) with its VALU work cut into steps of four instructions placed in the MFMA gaps (a lone wave hides
//     ~5 VALU instructions behind a 32-cycle MFMA and pays ~4.3 cycles for each one beyond that: tools/probes/stream1w.hip).
// The unit of the software pipeline is a SUB-TILE of 32 keys (20 MFMAs: 8 for the scores of the next sub-tile, 12 for V^T P^T and the
// denominators of this one; ~120 VALU instructions: 32 fma + 32 exp2 + 16 conversions for this sub-tile's probabilities, the maximum search of
// the next one).  The scores are the only accumulators the VALU reads, so THEY live in VGPRs (two buffers of 2 x 16 registers, MFMAs written in
// assembly with "v" destinations: no v_accvgpr_read per score) while the output and denominator accumulators stay in AGPRs (compiler
// builtins).  Nothing the compiler cannot see depends on MFMA latency: a score register is first read >= 6 MFMAs after the MFMA that wrote it
// (the one place that is not true -- the first sub-tile of a workgroup -- has explicit s_nops).
// Rings: K 4 stages, V^T 3 stages of 64 keys (LDS-DMA issued from assembly, counted vmcnt waits + one barrier per 64 keys).
// Rows leave through a per-wave LDS slab: a lane owns a query, so the direct store is 32 different cache lines per instruction (9.9 k cycles
// per workgroup measured); transposed, one instruction writes four whole 256-byte row segments.
#include "gn_common.h"
#include <type_traits>

namespace gn {

namespace {
constexpr int KT = 64;                 // keys per ring stage
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kRing = 64 * 64;         // shorts per ring stage (one 64 x 64 16-bit tile)
constexpr int kSlabPitch = 272;        // bytes per row of the output slab (256 + 16: b128 reads stay aligned, b64 writes 2-way conflicted)
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GN_PIN() __builtin_amdgcn_sched_barrier(0)

template <bool F16> __device__ __forceinline__ f32x16 mfma16(const bf16x8& x, const bf16x8& y, const f32x16& c) {
  if constexpr (F16) {
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, x), __builtin_bit_cast(f16x8_t, y), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
  }
}
// score MFMAs with the accumulator in VGPRs (the compiler's own choice for this kernel is AGPRs for every MFMA destination)
template <bool F16> __device__ __forceinline__ void mfma_v0(f32x16& s, const bf16x8& x, const bf16x8& y) {
  if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s) : "v"(x), "v"(y));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(x), "v"(y));
}
template <bool F16> __device__ __forceinline__ void mfma_v(f32x16& s, const bf16x8& x, const bf16x8& y) {
  if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "v"(x), "v"(y));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(x), "v"(y));
}

// ABL (timing only, wrong results): 1 no exponentials, 2 no LDS-DMA in the loop, 4 no barrier in the loop, 16 no maximum search, 32 no probability steps,
// 64 no score MFMAs, 256 no fragment reads, 524288 the rescale test without its branch; 8 = s_memtime stamps into a.part (results stay valid).
// (Removing the V^T P^T MFMAs is not a usable probe: the probabilities become dead code and the compiler deletes their steps.)
template <bool F16, int ABL>
__global__ __launch_bounds__(256) void k_attn_pw(AttnArgs a, int nitems) {
  constexpr int NDK = 4, NDV = 3, NW = 4, NQ = 2, IPW = 8 / NW;
  // K ring [NDK][64 keys][64], V^T ring [NDV][64 dims][64 keys], output slabs [NW][32 rows][kSlabPitch bytes]
  __shared__ __attribute__((aligned(1024))) unsigned short smem[(NDK + NDV) * kRing + NW * 32 * kSlabPitch / 2];
  // persistent form: one workgroup per CU walks the call's list of (slot, head, 256-query block) items that hold valid queries (or, without a list,
  // every item).  XCD-aware order: workgroup L lives on XCD L & 7 and takes positions (L & 7) * (gridDim.x / 8) + (L >> 3) + gridDim.x * i of the list, so
  // that the query blocks of one (slot, head) -- neighbours in the list -- run at the same time on the same L2.
  const int gx = a.npad / (NW * 32 * NQ);
  const int n_work = a.tiles != nullptr ? a.tiles[1] : nitems;
  const int first = (int)(blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
#pragma unroll 1
  for (int work = first; work < n_work; work += gridDim.x) {
  int tid = threadIdx.x;      // opaque per iteration: nothing derived from it is hoisted out of the item loop (register budget)
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, ql = lane & 31;
  int qblk, h, bs;
  {
    const int v = a.tiles != nullptr ? a.tiles[kTileListBase + a.BS * (a.npad / 128) + work] : work;
    qblk = v % gx;
    const int g = v / gx;
    h = g % kHeads; bs = g / kHeads;
  }
  long long ts[8];
  auto stamp = [&](int k) __attribute__((always_inline)) { if (ABL & 8) ts[k] = (long long)__builtin_amdgcn_s_memtime(); };
  stamp(0);
  const int kvs = a.cross ? (bs ^ 1) : bs;
  const int q0 = qblk * (NW * 32 * NQ) + wave * 32 * NQ;

  // LDS-DMA addressing (source-side swizzle f(row) = (row ^ (row >> 3)) & 7), as in k_attn16_v5.  The DMA instruction is written in assembly:
  // to the compiler `global_load_lds` is a FLAT access that may return out of order with DS reads, so while one is in flight (always, here) it
  // turns every wait for a fragment read into lgkmcnt(0) -- which also waits for the PREFETCH issued just before, exposing the LDS latency
  // once per fragment.  Hidden from it, the fragment waits are counted; the ring protocol is the explicit waits + barriers below.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned kvo[IPW], vvo[IPW];              // per-lane byte offsets inside a tile's share
  const unsigned short* kbase[IPW]; const unsigned short* vbase[IPW];   // wave-uniform bases
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int r0 = (IPW * wave_u + j) * 8, r = r0 + (lane >> 3);
    const int c = (lane & 7) ^ ((r ^ (r >> 3)) & 7);
    kvo[j] = (unsigned)(((lane >> 3) * a.ldkb + c * 8) * 2);
    vvo[j] = (unsigned)(((lane >> 3) * a.npad + c * 8) * 2);
    kbase[j] = a.kb + ((size_t)kvs * a.npad + r0) * a.ldkb + h * 64;
    vbase[j] = a.vt + (((size_t)kvs * kHeads + h) * kHeadDim + r0) * a.npad;
  }
  const size_t kstep = (size_t)KT * a.ldkb;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  auto dma16 = [&](unsigned voff, const unsigned short* sbase, unsigned lds_byte) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
  };
#define GN_DMA_K(stage, t)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < IPW; ++j)                                                                       \
    dma16(kvo[j], kbase[j] + (size_t)(t) * kstep, lds0 + 2 * ((stage) * kRing + (IPW * wave_u + j) * 512));
#define GN_DMA_V(stage, t)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < IPW; ++j)                                                                       \
    dma16(vvo[j], vbase[j] + (t) * KT, lds0 + 2 * ((NDK + (stage)) * kRing + (IPW * wave_u + j) * 512));

  // the first three K tiles and two V^T tiles are requested before anything else (npad >= 256: the rows exist whatever the key count is)
  GN_DMA_K(0, 0);
  GN_DMA_K(1, 1); GN_DMA_V(0, 0);
  GN_DMA_K(2, 2); GN_DMA_V(1, 1);
  const int nkv = a.nvalid[kvs];
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
  float m_run[NQ] = {-INFINITY, -INFINITY}, mneg[NQ] = {0.f, 0.f};

  // fragment addresses (bytes, relative to a ring stage): K rows kt * 32 + ql, 16-byte chunk (2 c + hh) ^ f(row); V^T rows d * 32 + ql, chunk
  // (4 kt + 2 u + hh) ^ f(row)
  unsigned kfo[2][4], vfo[2][2][2];
#pragma unroll
  for (int i2 = 0; i2 < 2; ++i2) {
    const int row = i2 * 32 + ql, f = (row ^ (row >> 3)) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) kfo[i2][c] = (unsigned)(row * 128 + (((2 * c + hh) ^ f) << 4));
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) vfo[i2][kt][u] = (unsigned)(row * 128 + (((4 * kt + 2 * u + hh) ^ f) << 4));
  }
  const unsigned char* const lds = reinterpret_cast<const unsigned char*>(smem);
  auto k_frag = [&](unsigned stage_byte, int kt, int c) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>(lds + (stage_byte + kfo[kt][c]));
  };
  auto v_frag = [&](unsigned stage_byte, int d, int kt, int u) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>(lds + (stage_byte + vfo[d][kt][u]));
  };

  // ---- per-wave state of the software pipeline
  f32x16 Sa[NQ], Sb[NQ];    // score accumulators (VGPRs) of two consecutive sub-tiles
  bf16x8 pf[NQ][2];         // probabilities of the current sub-tile as B operands: [query tile][16-key group]
  float mloc[NQ];
  bool grow = false;

  // maximum search of one query tile of a sub-tile (keys key0 .. key0 + 31) in three steps: five independent v_max3 over the lane's 16 scores
  // (st 0), their merge (st 1: a dependent chain of v_max3 costs a lone wave ~9 cycles per link), the test against the reference (st 2).
  // A lane holds half of a query's keys (hh): the test is made per half -- the wave-wide ballot sees either -- and the halves only meet in the
  // rescale itself (rare), so the tile loop has no cross-lane operation.  Masked keys become -inf first.
  float mt[NQ][5];
  auto m_step = [&](f32x16 (&S)[NQ], int qi, int st, int key0, auto mask_tag) __attribute__((always_inline)) {
    if (st == 0) {
      if constexpr (decltype(mask_tag)::value) {
        const int lim = nkv - key0 - 4 * hh;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((r & 3) + 8 * (r >> 2) >= lim) S[qi][r] = -INFINITY;
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) mt[qi][i] = __builtin_fmaxf(__builtin_fmaxf(S[qi][3 * i], S[qi][3 * i + 1]), S[qi][3 * i + 2]);
    } else if (st == 1) {
      const float x = __builtin_fmaxf(__builtin_fmaxf(mt[qi][0], mt[qi][1]), mt[qi][2]);
      const float y = __builtin_fmaxf(__builtin_fmaxf(mt[qi][3], mt[qi][4]), S[qi][15]);
      mloc[qi] = __builtin_fmaxf(x, y);
    } else {
      grow = grow || (mloc[qi] - m_run[qi]) * kLog2e > 8.0f;
    }
  };
  // lazy maximum: keep the stale reference unless some query's maximum grew by more than 2^8 (probabilities stay <= 256)
  auto rescale = [&]() __attribute__((always_inline)) {
    if (ABL & 524288) { asm volatile("" ::"v"(grow ? 1 : 0)); }     // timing: the test without the branch
    else if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        const unsigned mu = __float_as_uint(mloc[qi]);
        const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);     // {low half in both halves, high half in both halves}
        const unsigned lo = sw[0], hi = sw[1];     // (by value: __builtin_bit_cast of the vector's elements read element 0 twice with this compiler)
        const float m_new = fmaxf(m_run[qi], __builtin_fmaxf(__uint_as_float(lo), __uint_as_float(hi)));
        const float alpha = __builtin_amdgcn_exp2f((m_run[qi] - m_new) * kLog2e);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[qi][0][r] *= alpha; o[qi][1][r] *= alpha; ol[qi][r] *= alpha; }
        m_run[qi] = m_new;
      }
    }
    grow = false;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) mneg[qi] = -m_run[qi] * kLog2e;
  };
  // probabilities of one fragment (query tile qi, 16-key group u) of a sub-tile in five steps of four instructions
  float xe[8];
  auto e_step = [&](const f32x16 (&S)[NQ], int f, int st) __attribute__((always_inline)) {      // f = 2 u + qi
    const int qi = f & 1, u = f >> 1;
    if (st < 2) {
#pragma unroll
      for (int e = 4 * st; e < 4 * st + 4; ++e) xe[e] = __builtin_fmaf(S[qi][8 * u + e], kLog2e, mneg[qi]);
    } else if (st < 4) {
#pragma unroll
      for (int e = 4 * (st - 2); e < 4 * (st - 2) + 4; ++e) xe[e] = (ABL & 1) ? xe[e] : __builtin_amdgcn_exp2f(xe[e]);
    } else {
      u32x4 pw;
#pragma unroll
      for (int e = 0; e < 4; ++e) pw[e] = pack16<F16>(xe[2 * e], xe[2 * e + 1]);
      pf[qi][u] = __builtin_bit_cast(bf16x8, pw);
    }
  };

  // a wait the compiler SEES, on every path into the tile loop: its model holds the Q-fragment loads as pending, and it would otherwise wait for
  // them inside the loop with vmcnt(N) counts that also drain the (invisible) DMA groups
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("s_barrier" ::: "memory");
  stamp(1);

  bf16x8 kf[3];     // K fragments in flight: the one being multiplied and two prefetched
  if (ABL & (32 | 256)) { kf[0] = kf[1] = kf[2] = ones; pf[0][0] = pf[0][1] = pf[1][0] = pf[1][1] = ones; }
  if (ntiles > 0) {
    // scores of sub-tile 0 (not pipelined: once per workgroup)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8 kk = k_frag(0, 0, c);
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) { if (c == 0) mfma_v0<F16>(Sa[qi], kk, qf[qi][c]); else mfma_v<F16>(Sa[qi], kk, qf[qi][c]); }
    }
    // the compiler does not know the latency of an MFMA written in assembly: the wait carries the accumulators, or the first reads are scheduled in front of it
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(Sa[0]), "+v"(Sa[1]));
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) { m_step(Sa, qi, 0, 0, std::true_type{}); m_step(Sa, qi, 1, 0, std::true_type{}); m_step(Sa, qi, 2, 0, std::true_type{}); }
    rescale();
    kf[0] = k_frag(0, 1, 0);
    kf[1] = k_frag(0, 1, 1);
  }
  stamp(2);

  // one sub-tile: [8 score MFMAs of the NEXT sub-tile (into SN) || probabilities of this one, first 16 keys] -> [6 MFMAs V^T P^T + denominators of
  // the first 16 keys || probabilities of the other 16] -> [6 MFMAs for those || maximum search of the next sub-tile] -> lazy rescale
  //   kst, ktn:   ring stage (bytes) and key half of the K rows the score MFMAs read;  kst2, kt2: the same for the sub-tile after (fragment prefetch)
  //   vst, kt:    ring stage (bytes) of the V^T tile and this sub-tile's key half
  // No control flow inside: sched_barrier pins the schedulers, not MachineSink -- with a branch in the stream the probability steps were sunk
  // into the block of their first use, behind three MFMAs.  MASK (the next sub-tile holds keys >= nkv) is therefore a compile-time variant.
  auto subtile = [&](const f32x16 (&SC)[NQ], f32x16 (&SN)[NQ], unsigned kst, int ktn, unsigned kst2, int kt2, unsigned vst, int kt, int next_key0,
                     auto mask, bool live) __attribute__((always_inline)) {
    bf16x8 vf[2][2];
    if (ABL & 256) vf[0][0] = vf[0][1] = vf[1][0] = vf[1][1] = ones;
    // ---- slots 0..7: scores of the next sub-tile
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        const int m = 2 * c + qi;
        if (!(ABL & 64)) { if (c == 0) mfma_v0<F16>(SN[qi], kf[c % 3], qf[qi][c]); else mfma_v<F16>(SN[qi], kf[c % 3], qf[qi][c]); }
        // steps [10 m / 8, 10 (m + 1) / 8) of the 10 steps of fragments (qi = 0, u = 0), (1, 0)
#pragma unroll
        for (int st = (10 * m) / 8; st < (10 * (m + 1)) / 8; ++st) if (!(ABL & 32)) e_step(SC, st / 5, st % 5);
        if (qi == 0 && !(ABL & 256)) {   // fragment reads behind the first MFMA of the pair (in front of it, the wait for kf[c] would also wait for them)
          __builtin_amdgcn_sched_barrier(0);
          if (c < 2) kf[(c + 2) % 3] = k_frag(kst, ktn, c + 2);
          vf[c & 1][c >> 1] = v_frag(vst, c & 1, kt, c >> 1);
        }
        GN_PIN();
      }
    }
    // ---- slots 8..19: V^T P^T and denominators
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {        // ol[0], ol[1], o[0][0], o[1][0], o[0][1], o[1][1]
        const int qi = i & 1, m = 6 * u + i;
        if (u == 1 && i == 0 && !(ABL & 256)) kf[0] = k_frag(kst2, kt2, 0);      // fragments of the sub-tile after the next one
        if (u == 1 && i == 2 && !(ABL & 256)) kf[1] = k_frag(kst2, kt2, 1);
        if (i < 2) ol[qi] = mfma16<F16>(ones, pf[qi][u], ol[qi]);
        else o[qi][(i - 2) >> 1] = mfma16<F16>(vf[(i - 2) >> 1][u], pf[qi][u], o[qi][(i - 2) >> 1]);
        if (m < 6) {
          // probabilities of the second 16 keys: 10 steps over 6 slots
#pragma unroll
          for (int st = (10 * m) / 6; st < (10 * (m + 1)) / 6; ++st) if (!(ABL & 32)) e_step(SC, 2 + st / 5, st % 5);
        } else {
          // maximum search of the next sub-tile: 2 x 3 steps over the first four slots -- the test's compare is two MFMAs old when the
          // branch behind the sub-tile reads it
          const int mm = m - 6;
          if (ABL & 16) { }
          else if (mm < 2) m_step(SN, mm, 0, next_key0, mask);
          else if (mm == 2) { m_step(SN, 0, 1, next_key0, mask); m_step(SN, 1, 1, next_key0, mask); }
          else if (mm == 3) { m_step(SN, 0, 2, next_key0, mask); m_step(SN, 1, 2, next_key0, mask); }
        }
        GN_PIN();
      }
    }
    // The score accumulators stay allocated to the end of the sub-tile even when nothing reads them (the scores behind the last tile): to the
    // compiler an MFMA written in assembly has delivered its result when it is issued, so it gave the dead accumulator's registers to the V^T
    // fragment read in the next instruction -- and the MFMA, 8 passes later, wrote its result over the fragment (one 32-wide output tile of a wave
    // without its last 16 keys in about one launch of twelve; tools/probes/mfma_war.hip shows the opposite order, overwriting an operand behind the
    // MFMA that reads it, is safe)
    asm volatile("" ::"v"(SN[0]), "v"(SN[1]));
    if (live) rescale(); else grow = false;
  };

  int k0 = 0, k1 = 1, k2 = 2, k3 = 3;   // K ring stages of tiles t, t + 1, t + 2, t + 3
  int v0 = 0, v1 = 1, v2 = 2;           // V^T ring stages of tiles t, t + 1, t + 2
  auto tile = [&](int t, auto mask_a, auto mask_b) __attribute__((always_inline)) {
    // K(t+1) and V^T(t) have landed once everything but the newest DMA group ({K(t+2), V^T(t+1)}) is complete; lgkmcnt(0): this wave's fragment
    // reads of the stages refilled below have returned; the barrier publishes all waves' shares and proves those stages are no longer being read
    if (ABL & 4) { }
    else if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!(ABL & 2)) {
      if (t + 3 < ntiles) GN_DMA_K(k3, t + 3);        // the stage K(t-1) was read from (its second half one sub-tile ago)
      if (t + 2 < ntiles) GN_DMA_V(v2, t + 2);        // the stage V^T(t-1) was read from
    }
    const unsigned ks0 = 2 * kRing * k0, ks1 = 2 * kRing * k1, vs0 = 2 * kRing * (NDK + v0);
    GN_PIN();
    // sub-tile 2 t: scores of keys 32..63 of tile t; prefetch the first fragments of keys 0..31 of tile t + 1
    subtile(Sa, Sb, ks0, 1, ks1, 0, vs0, 0, t * KT + 32, mask_a, true);
    // sub-tile 2 t + 1: scores of keys 0..31 of tile t + 1 (stale bytes behind the last tile: computed, never used)
    subtile(Sb, Sa, ks1, 0, ks1, 1, vs0, 1, t * KT + 64, mask_b, t + 1 < ntiles);
    { const int s_ = k0; k0 = k1; k1 = k2; k2 = k3; k3 = s_; }
    { const int s_ = v0; v0 = v1; v1 = v2; v2 = s_; }
  };
  // only the last tile can hold keys >= nkv: its first half is searched in the second sub-tile of tile ntiles - 2, its second half in the first
  // sub-tile of tile ntiles - 1 (the masked variants compare every key index with nkv: correct for full tiles as well)
#pragma unroll 1
  for (int t = 0; t + 2 < ntiles; ++t) tile(t, std::false_type{}, std::false_type{});
  if (ntiles >= 2) tile(ntiles - 2, std::false_type{}, std::true_type{});
  if (ntiles >= 1) tile(ntiles - 1, std::true_type{}, std::false_type{});
  stamp(3);
  stamp(4);
#undef GN_DMA_K
#undef GN_DMA_V

  // ---------------------------------------------------------------- rows out
  if (a.outp != nullptr) {   // hm16 rows (x = xh + xm) for the block tail: head h of a row is 256 contiguous bytes (4 groups of 16 high + 16 residual terms)
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    unsigned char* const slab = reinterpret_cast<unsigned char*>(smem) + 2 * (NDK + NDV) * kRing + wave * 32 * kSlabPitch;
    // the guard only looks at rows of VALID queries: with work lists the q rows of a padding-only 128-token tile inside a partly valid 256-query block
    // are whatever an earlier call left there (k_qkv skips such tiles) -- after a call that overflowed, possibly inf -- and must not trip later calls
    const int nq = a.nvalid[bs];
    float amax = 0.f;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      const float l = ol[qi][0];
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      float amq = 0.f;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4v w = {o[qi][d][4 * g + 0] * inv, o[qi][d][4 * g + 1] * inv, o[qi][d][4 * g + 2] * inv, o[qi][d][4 * g + 3] * inv};
          ovf_track(amq, w.x, w.y); ovf_track(amq, w.z, w.w);
          const f16x4 hv = __builtin_convertvector(w, f16x4);
          const f16x4 mv = __builtin_convertvector(w - __builtin_convertvector(hv, f32x4v), f16x4);
          unsigned char* pp = slab + ql * kSlabPitch + (2 * d + (g >> 1)) * 64 + (8 * (g & 1) + 4 * hh) * 2;   // dims d * 32 + 8 g + 4 hh ..
          *reinterpret_cast<f16x4*>(pp) = hv;
          *reinterpret_cast<f16x4*>(pp + 32) = mv;
        }
      if (q0 + 32 * qi + ql < nq) amax = fmaxf(amax, amq);
      // (the same wave wrote the slab: the LDS operations of a wave complete in order)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = 4 * j + (lane >> 4), ch = lane & 15;
        const uint4 val = *reinterpret_cast<const uint4*>(slab + row * kSlabPitch + ch * 16);
        uint16_t* dst = a.outp + hm16_off((size_t)bs * a.npad + q0 + 32 * qi + row, a.ldo, h * 64) + ch * 8;
        *reinterpret_cast<uint4*>(dst) = val;
      }
    }
    ovf_commit(a.ovf, amax);
  } else {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      const float l = ol[qi][0];
      const float inv = l > 0.f ? 1.0f / l : 0.f;
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
  // the rings and the slab are free for the next item once every wave is here (and its LDS reads have returned)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (!(ABL & 8)) continue;
  if (ABL & 8) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(5);
    if (lane == 0) {
      long long* dst = reinterpret_cast<long long*>(a.part) + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * NW + wave) * 8;
      for (int k = 0; k < 6; ++k) dst[k] = ts[k];
      dst[6] = ntiles; dst[7] = 0;
    }
  }
  }   // work
}
#undef GN_PIN
}  // namespace

// grids of whole 256-query blocks; false = not applicable, the caller launches k_attn16_v5
bool launch_attention_pw(const AttnArgs& a, int ablate, hipStream_t s) {
  if (a.npad % 256 != 0 || a.qb == nullptr) return false;
  const int nitems = a.npad / 256 * kHeads * a.BS;
  if (nitems < 8) return false;
  const int ncu = a.ncu >= 8 ? a.ncu : std::max(device_cu_count(), 8);
  // one workgroup per CU (LDS and registers leave room for one) walks the work list; a multiple of the 8 XCDs.  The phase-stamp variants keep one
  // workgroup per item (tools/attn_pw_ab.py reads one record per item).
  const bool stamps = ablate == 3 || ablate >= 100;
  const dim3 grid((unsigned)((stamps || nitems < ncu) ? (nitems & ~7) : (ncu & ~7))), block(256);
  if (a.half_fmt) {
    if (ablate == 1) hipLaunchKernelGGL((k_attn_pw<true, 1>), grid, block, 0, s, a, nitems);
    else if (ablate == 2) hipLaunchKernelGGL((k_attn_pw<true, 2>), grid, block, 0, s, a, nitems);
    else if (stamps) {
      if (g_attn_stamps == nullptr) return false;
      AttnArgs b = a; b.part = reinterpret_cast<float*>(g_attn_stamps); b.tiles = nullptr;
#define GN_PW_ABL(x) case x: hipLaunchKernelGGL((k_attn_pw<true, (x) | 8>), grid, block, 0, s, b, nitems); break;
      switch (ablate >= 100 ? ablate - 100 : 0) {
        GN_PW_ABL(0) GN_PW_ABL(1) GN_PW_ABL(2) GN_PW_ABL(4) GN_PW_ABL(6) GN_PW_ABL(16) GN_PW_ABL(32) GN_PW_ABL(48) GN_PW_ABL(64) GN_PW_ABL(256) GN_PW_ABL(524288) GN_PW_ABL(524320)
        default: return false;
      }
#undef GN_PW_ABL
    }
    else hipLaunchKernelGGL((k_attn_pw<true, 0>), grid, block, 0, s, a, nitems);
    g_last_kernel = "k_attn_pw<true, 0>";
  } else {
    hipLaunchKernelGGL((k_attn_pw<false, 0>), grid, block, 0, s, a, nitems);
    g_last_kernel = "k_attn_pw<false, 0>";
  }
  return true;
}

}  // namespace gn
