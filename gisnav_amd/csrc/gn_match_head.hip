// Match head: sigmoid-log double softmax, mutual arg-max, threshold, ordered compaction, and the
// matched-point gather + DEM lift that follows it in PoseNode.
//
// Stands in for kornia's `sigmoid_log_double_softmax` + `filter_matches` + the LightGlueMatcher
// return packing (call site ros/gisnav/gisnav/core/pose_node.py:285-287), the gathers at
// pose_node.py:289-297 and `_compute_3d_points` (core/_shared.py:95-102).
//
//   P[i][j] = ((S[i][j] - rmax_i) - rlog_i) + ((S[i][j] - cmax_j) - clog_j) + (ls0_i + ls1_j)
//
// evaluated in exactly the association order of the reference expression
// `log_softmax(sim, 2) + log_softmax(sim^T, 2)^T + certainties`, so that the arg-max decision sees the
// same rounding structure; ties resolve to the LOWEST index like torch.max.  The (N+1)x(M+1) matrix
// with the dustbin row/column is never materialised: filter_matches only looks at [:-1, :-1].
//
// The similarity matrix S comes from the MFMA GEMM (gn_gemm.hip) and stays L2/Infinity-Cache
// resident (4 MB per pair at 1024 keypoints); the passes here are coalesced row/column sweeps.
#include <type_traits>

#include "gn_common.h"

namespace gn {

namespace {
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

// grid (npad/4, B): one wave per row
__global__ __launch_bounds__(256) void k_row_stats(HeadArgs a) {
  const int lane = threadIdx.x & 63, b = blockIdx.y;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n0 = a.nvalid[2 * b], n1 = a.nvalid[2 * b + 1];
  if (i >= n0) return;
  const float* row = a.sim + ((size_t)b * a.npad + i) * a.npad;
  float m = -INFINITY;
  for (int j = lane; j < n1; j += 64) m = fmaxf(m, row[j]);
  m = wave_max(m);
  float s = 0.f;
  for (int j = lane; j < n1; j += 64) s += expf(row[j] - m);
  s = wave_sum(s);
  if (lane == 0) { a.rowmax[(size_t)b * a.npad + i] = m; a.rowlog[(size_t)b * a.npad + i] = logf(s); }
}

// grid (npad/64, B): kColGroups row groups x 64 columns (16 waves per workgroup keep the dependent row loop short)
constexpr int kColGroups = 16;
__global__ __launch_bounds__(64 * kColGroups) void k_col_stats(HeadArgs a) {
  __shared__ float red[kColGroups][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6, b = blockIdx.y;
  const int j = blockIdx.x * 64 + c;
  const int n0 = a.nvalid[2 * b], n1 = a.nvalid[2 * b + 1];
  const float* base = a.sim + (size_t)b * a.npad * a.npad + j;
  const bool act = j < n1;
  float m = -INFINITY;
  if (act) for (int i = g; i < n0; i += kColGroups) m = fmaxf(m, base[(size_t)i * a.npad]);
  red[g][c] = m;
  __syncthreads();
  m = red[0][c];
#pragma unroll
  for (int k = 1; k < kColGroups; ++k) m = fmaxf(m, red[k][c]);
  __syncthreads();
  float s = 0.f;
  if (act) for (int i = g; i < n0; i += kColGroups) s += expf(base[(size_t)i * a.npad] - m);
  red[g][c] = s;
  __syncthreads();
  if (g == 0 && act) {
    float t[kColGroups];
#pragma unroll
    for (int k = 0; k < kColGroups; ++k) t[k] = red[k][c];
#pragma unroll
    for (int w = 1; w < kColGroups; w <<= 1)          // fixed pairwise tree: deterministic
#pragma unroll
      for (int k = 0; k < kColGroups; k += 2 * w) t[k] = t[k] + t[k + w];
    a.colmax[(size_t)b * a.npad + j] = m;
    a.collog[(size_t)b * a.npad + j] = logf(t[0]);
  }
}

__device__ inline float score_at(float s, float rm, float rl, float cm, float cl, float li, float lj) {
  return ((s - rm) - rl) + ((s - cm) - cl) + (li + lj);
}

// grid (npad/4, B): row arg-max, first index on ties
__global__ __launch_bounds__(256) void k_row_argmax(HeadArgs a) {
  const int lane = threadIdx.x & 63, b = blockIdx.y;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n0 = a.nvalid[2 * b], n1 = a.nvalid[2 * b + 1];
  if (i >= n0) return;
  const size_t ro = (size_t)b * a.npad;
  const float* row = a.sim + (ro + i) * a.npad;
  const float rm = a.rowmax[ro + i], rl = a.rowlog[ro + i];
  const float li = a.ls[(size_t)(2 * b) * a.npad + i];
  const float* lsj = a.ls + (size_t)(2 * b + 1) * a.npad;
  float best = -INFINITY; int bj = 0x7fffffff;
  for (int j = lane; j < n1; j += 64) {
    const float p = score_at(row[j], rm, rl, a.colmax[ro + j], a.collog[ro + j], li, lsj[j]);
    if (p > best || bj == 0x7fffffff) { best = p; bj = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o); const int oj = __shfl_xor(bj, o);
    if (oj != 0x7fffffff && (bj == 0x7fffffff || ob > best || (ob == best && oj < bj))) { best = ob; bj = oj; }
  }
  if (lane == 0) { a.m0[ro + i] = bj; a.max0[ro + i] = best; }
}

// grid (npad/64, B): column arg-max, first (lowest i) index on ties
__global__ __launch_bounds__(64 * kColGroups) void k_col_argmax(HeadArgs a) {
  __shared__ float rv[kColGroups][64];
  __shared__ int ri[kColGroups][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6, b = blockIdx.y;
  const int j = blockIdx.x * 64 + c;
  const int n0 = a.nvalid[2 * b], n1 = a.nvalid[2 * b + 1];
  const size_t ro = (size_t)b * a.npad;
  float best = -INFINITY; int bi = 0x7fffffff;
  if (j < n1) {
    const float* base = a.sim + ro * a.npad + j;
    const float cm = a.colmax[ro + j], cl = a.collog[ro + j];
    const float lj = a.ls[(size_t)(2 * b + 1) * a.npad + j];
    const float* lsi = a.ls + (size_t)(2 * b) * a.npad;
    for (int i = g; i < n0; i += kColGroups) {
      const float p = score_at(base[(size_t)i * a.npad], a.rowmax[ro + i], a.rowlog[ro + i], cm, cl, lsi[i], lj);
      if (p > best || bi == 0x7fffffff) { best = p; bi = i; }
    }
  }
  rv[g][c] = best; ri[g][c] = bi;
  __syncthreads();
  if (g == 0 && j < n1) {
#pragma unroll
    for (int k = 1; k < kColGroups; ++k) {
      const float ob = rv[k][c]; const int oi = ri[k][c];
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
    }
    a.m1[ro + j] = bi;
  }
}

// grid (B): mutual check + threshold + order-preserving compaction
__global__ __launch_bounds__(256) void k_compact(HeadArgs a) {
  __shared__ int wcount[4];
  __shared__ int base_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = a.nvalid[2 * b], n1 = a.nvalid[2 * b + 1];
  const size_t ro = (size_t)b * a.npad;
  if (tid == 0) base_s = 0;
  __syncthreads();
  // (the five-pass developer form keeps no runner-ups: with a certificate requested every pair that has matches to decide reports "uncertain")
  if (n0 < 2 || n1 < 2 || (a.ovf != nullptr && *a.ovf != 0u)) {  // kornia LightGlueMatcher._no_match; or the f16x2 domain guard tripped (gn_common.h)
    if (tid == 0) { a.n_match[b] = 0; if (a.uncert) a.uncert[b] = (n0 < 2 || n1 < 2) ? 0 : 2; if (a.uncert_alt) a.uncert_alt[b] = 0; }
    return;
  }
  if (tid == 0 && a.uncert) a.uncert[b] = 1;
  if (tid == 0 && a.uncert_alt) a.uncert_alt[b] = 1;
  for (int i0 = 0; i0 < n0; i0 += 256) {
    const int i = i0 + tid;
    bool valid = false; int j = 0; float sc = 0.f;
    if (i < n0) {
      j = a.m0[ro + i];
      sc = expf(a.max0[ro + i]);
      valid = (a.m1[ro + j] == i) && (sc > a.threshold);
    }
    const unsigned long long bal = __ballot(valid);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wcount[w];
    if (valid) {
      const size_t k = (size_t)b * a.kmax + off + before;
      a.idx[2 * k] = i; a.idx[2 * k + 1] = j;
      a.score[k] = sc;
    }
    __syncthreads();
    if (tid == 0) base_s += wcount[0] + wcount[1] + wcount[2] + wcount[3];
    __syncthreads();
  }
  if (tid == 0) a.n_match[b] = base_s;
}

// ------------------------------------------------------------------------------------------------
// Fused match head: the similarity matrix is never written.  Two sweeps over S = md0 . md1^T, each recomputing the tiles on the
// matrix cores (the GEMM is 0.5 GFLOP per pair; writing + four times re-reading 4 MB per pair cost more than computing it twice):
//   k_head_fused<., 1>  row and column soft-max statistics (running max / sum of exponentials), finished per pair by the LAST workgroup
//   k_head_fused<., 2>  score_at() in the reference's association order, row and column arg-max partials; the last workgroup of a
//                       pair reduces them, does the mutual check, the threshold and the ordered compaction.
// grid (npad / 128, B, S), 8 waves: a workgroup owns 128 rows of image 0 and a contiguous 1 / S of the 64-column tiles of image 1
// (S > 1 only for small batches, to occupy the chip); wave (wq, wc) owns rows 32 wq .. and columns 32 wc .. of every tile.  The row
// operand lives in registers for the whole sweep (32 x 16-byte fragments per lane); column tiles (64 rows x 1 KB) are double-buffered
// in LDS, staged by LDS-DMA while the previous tile is being worked on.  Per tile a wave issues its MFMAs, then does the VALU work on
// the accumulators; the two waves of a SIMD run the SAME phase at the same time (tools/probes/overlap.hip: an MFMA wave and a VALU
// wave sharing a SIMD slow each other down; two waves per SIMD still hide each other's latencies).  The sweeps are VALU-bound
// (~45 operations per score element); measured history in DESIGN.md section 4.
// F32 = false: hm16 rows, three v_mfma_f32_32x32x16_f16 per k-step (the f16x2 mode);  F32 = true: f32 rows, v_mfma_f32_32x32x2_f32.
// Both row formats are 1 KB per keypoint and use the same 16-byte fragment addressing.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int kHeadKT = 64 * 128;        // bytes of one 32-wide k-slot of a 64-row column tile
constexpr int kHeadTile = 8 * kHeadKT;   // 64 KB
constexpr int kHeadRows = 128;           // rows of image 0 per workgroup
constexpr int kHeadMaxSplit = 8;         // column splits (grid.z) the partial buffers are sized for
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ int hswz(int row) { return (row ^ (row >> 3)) & 7; }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
// (m, s) <- (m, s) (+) (om, os): running maximum and sum of exp(. - maximum); -inf maxima are empty sets
__device__ __forceinline__ void lse_merge(float& m, float& s, float om, float os) {
  const float mn = fmaxf(m, om);
  const float ref = mn == -INFINITY ? 0.f : mn;
  s = s * ex2((m - ref) * kLog2e) + os * ex2((om - ref) * kLog2e);
  m = mn;
}
// (m, s) <- (m, s) (+) {v}: one element joins.  One exponential: of the two factors exp(m - max) and exp(v - max) one is exactly 1.
// Empty sets carry m = kLseEmpty (a large negative FINITE number), masked elements arrive as v = -inf: neither makes a NaN
// (m - v = +inf -> e = 0 -> s + 0), so there is no validity select in the chain.
constexpr float kLseEmpty = -1.0e30f;
__device__ __forceinline__ void lse_push(float& m, float& s, float v) {
  const float e = ex2(-fabsf(m - v) * kLog2e);
  s = v > m ? __builtin_fmaf(s, e, 1.f) : s + e;
  m = fmaxf(m, v);
}
// Results that ANOTHER workgroup of the same launch reads (the last workgroup of a pair): device-coherent stores / loads
// (sc1: written through / read past the per-XCD L2).  A __threadfence() would write back and invalidate the whole L2 of the XCD
// under every other workgroup's feet (measured: 36 k cycles per workgroup at batch 32).
template <typename T> __device__ __forceinline__ void st_dev(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T ld_dev(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool arg_better(float ov, int oi, float v, int i) {   // (ov, oi) beats (v, i): larger value, lower index on ties
  return oi != 0x7fffffff && (i == 0x7fffffff || ov > v || (ov == v && oi < i));
}

// One k-step of one 32 x 32 tile on two INDEPENDENT accumulation chains (a dependent MFMA waits for the whole latency of the one
// before it): hm16: chain 0 = A_m B_h + A_h B_m (the small terms), chain 1 = A_h B_h;  f32: the eight MFMAs alternate.
struct Acc2 { f32x16 c[2]; };
template <bool F32, bool FIRST, int PART>   // FIRST: the chains start from zero (inline-constant C operand);  PART 0 / 1: first / second half of the step
__device__ __forceinline__ void head_mma(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1, Acc2& acc) {
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (F32) {
    const f32x4 fa = __builtin_bit_cast(f32x4, PART == 0 ? a0 : a1), fb = __builtin_bit_cast(f32x4, PART == 0 ? b0 : b1);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc.c[e & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], (FIRST && PART == 0 && e < 2) ? z : acc.c[e & 1], 0, 0, 0);
  } else {   // a0 / b0: high terms, a1 / b1: residual terms
    if constexpr (PART == 0) {
      acc.c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b0), FIRST ? z : acc.c[0], 0, 0, 0);
      acc.c[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), FIRST ? z : acc.c[1], 0, 0, 0);
    } else {
      acc.c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b1), acc.c[0], 0, 0, 0);
    }
  }
}

template <bool F32, int SWEEP, int ABL = 0>   // ABL, timing-only ablations (wrong results): 1 no DMA in the loop, 2 no VALU work, 4 no MFMA, 8 no LDS fragment reads
__global__ __launch_bounds__(512) void k_head_fused(HeadArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * kHeadTile];
  __shared__ int s_last;
  __shared__ float xch[3][3][4][64];   // [tile % 3][value][row quarter][column of the tile]: column partials of the four row quarters (value 2: sweep 2's runner-up)
  __shared__ f32x4 rowc[kHeadRows];    // SWEEP 2: (rowmax, rowlog, logsigmoid matchability, -) of the workgroup's rows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave & 3, wc = wave >> 2;       // row quarter (32 rows), column half (32 columns of every tile)
  const int hh = lane >> 5, ql = lane & 31;
  // XCD-aware placement: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the row blocks and
  // column splits of ONE pair -- which all stream the same 1 MB of image-1 descriptors -- are renumbered to share an XCD.
  const int S = gridDim.z, np = a.npad;
  int b, rb, sp;
  {
    const int per = gridDim.x * S, nwg = per * gridDim.y;
    const int L = blockIdx.x + gridDim.x * (blockIdx.z + S * blockIdx.y);
    int v = L;
    if ((nwg & 7) == 0) v = (L & 7) * (nwg >> 3) + (L >> 3);
    b = v / per; const int w_ = v - b * per; sp = w_ / gridDim.x; rb = w_ - sp * gridDim.x;
  }
  const int n0 = a.nvalid[2 * b], n1 = a.nvalid[2 * b + 1];
  const bool nomatch = n0 < 2 || n1 < 2;                 // kornia LightGlueMatcher._no_match
  const int nact = nomatch ? 1 : (n0 + kHeadRows - 1) / kHeadRows;   // row blocks of this pair that hold valid rows (block 0 always reports)
  if (rb >= nact) return;
  const int i0 = rb * kHeadRows;
  const int ntile_all = nomatch ? 0 : (n1 + 63) / 64;
  const int tbeg = ntile_all * sp / S, tend = ntile_all * (sp + 1) / S, ntile = tend - tbeg;   // this workgroup's column tiles
  const size_t ro = (size_t)b * np;
  const int npart = np / kHeadRows;                      // column partials per pair: one per row block
  const unsigned char* const A0 = reinterpret_cast<const unsigned char*>(a.md) + ((size_t)(2 * b) * np + i0) * 1024;
  const unsigned char* const B0 = reinterpret_cast<const unsigned char*>(a.md) + ((size_t)(2 * b + 1) * np + 64 * tbeg) * 1024;
  long long* const ts = a.dbg_ts ? a.dbg_ts + ((((size_t)(SWEEP - 1) * a.B + b) * (np / kHeadRows) + rb) * kHeadMaxSplit + sp) * 8 : nullptr;   // developer: phase stamps
  auto stamp = [&](int k) __attribute__((always_inline)) { if (ts && tid == 0) ts[k] = (long long)__builtin_amdgcn_s_memtime(); };
  stamp(0);

  // row operand: 32 fragments of 16 bytes (unit u = 32 bytes of the row: lane half hh takes bytes 16 hh ..)
  uint4 af[32];
  {
    const unsigned char* ap = A0 + (size_t)(32 * wq + ql) * 1024 + hh * 16;
#pragma unroll
    for (int u = 0; u < 32; ++u) af[u] = *reinterpret_cast<const uint4*>(ap + u * 32);
  }
  if (SWEEP == 2 && tid < kHeadRows) {
    const int i = i0 + tid;
    const f32x4 rc = {a.rowmax[ro + i], a.rowlog[ro + i], a.ls[(size_t)(2 * b) * np + i], 0.f};
    rowc[tid] = rc;
  }
  // column-tile staging by LDS-DMA (global_load_lds_dwordx4: one instruction moves 1 KB = 8 rows x 128 B straight into LDS, no
  // registers, nothing to wait for until the end of the tile).  Instruction q = 0..7 of wave w fills slot q, row group w; lane L
  // lands at position L & 7 of row 8 w + (L >> 3), so it fetches the chunk that belongs there.
  const unsigned char* ssrc;
  {
    const int row = 8 * wave + (lane >> 3);
    ssrc = B0 + (size_t)row * 1024 + (((lane & 7) ^ hswz(row)) * 16);
  }
  // The DMA is issued through inline assembly ON PURPOSE: hipcc's wait-count pass makes every ds_read that follows a
  // __builtin_amdgcn_global_load_lds wait for vmcnt(0) (it cannot prove that the tile being read is not the tile being filled), which
  // puts the full DMA latency in front of every k-step.  The waits that matter are written by hand: vmcnt(0) + barrier at the end
  // of every tile.  (The compiler's own vmcnt accounting only ever over-waits because of the in-flight operations it does not see.)
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  auto stage_tile = [&](int t) __attribute__((always_inline)) {
    const unsigned lb = __builtin_amdgcn_readfirstlane(lds0 + (t & 1) * kHeadTile + wave * 1024);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned char* g = ssrc + (size_t)t * 65536 + q * 128;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lb + q * kHeadKT), "v"(g) : "memory");
    }
  };
  // column fragments of step st (16 per tile): slot st >> 1, chunks 4 (st & 1) + hh and 4 (st & 1) + 2 + hh of row 32 wc + ql
  const int brow = (32 * wc + ql) * 128, bsw = hswz(32 * wc + ql);
  uint4 fb[2][2];   // [buffer][term]
  auto read_b = [&](int buf, const unsigned char* tile, int st) __attribute__((always_inline)) {
    const unsigned char* base = tile + (st >> 1) * kHeadKT + brow;
    fb[buf][0] = *reinterpret_cast<const uint4*>(base + (((4 * (st & 1) + hh) ^ bsw) * 16));
    fb[buf][1] = *reinterpret_cast<const uint4*>(base + (((4 * (st & 1) + 2 + hh) ^ bsw) * 16));
  };

  // rows of this lane: register r of a tile <-> local row 32 wq + 4 hh + (r & 3) + 8 (r >> 2)
  const int lrow0 = 32 * wq + 4 * hh, irow0 = i0 + lrow0;
  unsigned rvalid = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) rvalid |= (irow0 + (r & 3) + 8 * (r >> 2) < n0 ? 1u : 0u) << r;
  float ra[16], rbv[16];        // SWEEP 1: running row max / sum;  SWEEP 2: row best score (ra) / best column (rj) / runner-up score (rbv)
  int rj[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { ra[r] = SWEEP == 1 ? kLseEmpty : -INFINITY; rbv[r] = SWEEP == 1 ? 0.f : -INFINITY; rj[r] = 0x7fffffff; }
  if (ntile > 0) stage_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  stamp(1);

  Acc2 acc;
  // the MFMAs of tile t
  auto mma_tile = [&](int t) __attribute__((always_inline)) {
    if (ABL & 4) return;
    const unsigned char* const cur = smem + (t & 1) * kHeadTile;
    read_b(0, cur, 0);
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      if (st + 1 < 16 && !(ABL & 8)) read_b((st + 1) & 1, cur, st + 1);
      if (st == 0) head_mma<F32, true, 0>(af[0], af[1], fb[0][0], fb[0][1], acc);
      else head_mma<F32, false, 0>(af[2 * st], af[2 * st + 1], fb[st & 1][0], fb[st & 1][1], acc);
      head_mma<F32, false, 1>(af[2 * st], af[2 * st + 1], fb[st & 1][0], fb[st & 1][1], acc);
    }
  };
  // the VALU work of tile t on the accumulators: row statistics / arg-max in registers, the column's over this lane's 16 rows
  auto epi_tile = [&](int t) __attribute__((always_inline)) {
    if (ABL & 2) return;
    const int j = 64 * (tbeg + t) + 32 * wc + ql;        // this lane's column
    const bool cvalid = j < n1;
    float cm = 0.f, cl = 0.f, lj = 0.f;
    if (SWEEP == 2) { cm = a.colmax[ro + j]; cl = a.collog[ro + j]; lj = a.ls[(size_t)(2 * b + 1) * np + j]; }
    float c0[2], c1[2]; int ci[2];                        // two interleaved column chains (even / odd registers)
#pragma unroll
    for (int e = 0; e < 2; ++e) { c0[e] = SWEEP == 1 ? kLseEmpty : -INFINITY; c1[e] = SWEEP == 1 ? 0.f : -INFINITY; ci[e] = 0x7fffffff; }   // sweep 2: c1 = the chain's runner-up
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc.c[0][r] + acc.c[1][r];           // small terms + main term
      const bool rv = (rvalid >> r) & 1u;
      if (SWEEP == 1) {
        const float vr = cvalid ? v : -INFINITY;
        lse_push(ra[r], rbv[r], vr);
        lse_push(c0[r & 1], c1[r & 1], rv ? vr : -INFINITY);
      } else {
        const f32x4 rc = rowc[lrow0 + (r & 3) + 8 * (r >> 2)];
        const float p = score_at(v, rc[0], rc[1], cm, cl, rc[2], lj);
        // runner-up of the row / of the column chain: max(second, min(best so far, this one)) -- two instructions per direction, no index
        // (an exact tie leaves runner-up == best: gap 0, which the certificate treats as undecided)
        const float pr = cvalid ? p : -INFINITY, pc = (cvalid & rv) ? p : -INFINITY;
        rbv[r] = fmaxf(rbv[r], fminf(ra[r], pr));
        c1[r & 1] = fmaxf(c1[r & 1], fminf(c0[r & 1], pc));
        const bool take = cvalid & ((p > ra[r]) | (rj[r] == 0x7fffffff));
        ra[r] = take ? p : ra[r]; rj[r] = take ? j : rj[r];
        const bool takec = cvalid & rv & ((p > c0[r & 1]) | (ci[r & 1] == 0x7fffffff));
        c0[r & 1] = takec ? p : c0[r & 1]; ci[r & 1] = takec ? irow0 + (r & 3) + 8 * (r >> 2) : ci[r & 1];
      }
    }
    // merge the two chains, the two row halves of the wave; the four row quarters meet in LDS
    float* const xm = &xch[t % 3][0][wq][32 * wc + ql];
    float* const xs = &xch[t % 3][1][wq][32 * wc + ql];
    if (SWEEP == 1) {
      lse_merge(c0[0], c1[0], c0[1], c1[1]);
      const float om = __shfl_xor(c0[0], 32), os = __shfl_xor(c1[0], 32);
      lse_merge(c0[0], c1[0], om, os);
      if (hh == 0) { *xm = c0[0]; *xs = c1[0]; }
    } else {
      float* const x2 = &xch[t % 3][2][wq][32 * wc + ql];
      c1[0] = fmaxf(fmaxf(c1[0], c1[1]), fminf(c0[0], c0[1]));           // runner-up of the union: the larger runner-up, or the smaller best
      if (arg_better(c0[1], ci[1], c0[0], ci[0])) { c0[0] = c0[1]; ci[0] = ci[1]; }
      const float ov = __shfl_xor(c0[0], 32); const int oi = __shfl_xor(ci[0], 32); const float o2 = __shfl_xor(c1[0], 32);
      c1[0] = fmaxf(fmaxf(c1[0], o2), fminf(c0[0], ov));
      if (arg_better(ov, oi, c0[0], ci[0])) { c0[0] = ov; ci[0] = oi; }
      if (hh == 0) { *xm = c0[0]; *xs = __int_as_float(ci[0]); *x2 = c1[0]; }
    }
  };
  // one partial per (row block, column): wave 0 merges the four row quarters' values of tile tt (in increasing row order)
  auto flush_cols = [&](int tt) __attribute__((always_inline)) {
    if (wave != 0 || tt < 0) return;
    const int par = tt % 3, j = 64 * (tbeg + tt) + lane;
    const size_t po = ((size_t)b * npart + rb) * np + j;
    if (SWEEP == 1) {
      float m = xch[par][0][0][lane], sm = xch[par][1][0][lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) lse_merge(m, sm, xch[par][0][w][lane], xch[par][1][w][lane]);
      if (j < np) { st_dev(a.cpart_m + po, m); st_dev(a.cpart_s + po, sm); }
    } else {
      float bv = xch[par][0][0][lane]; int bi = __float_as_int(xch[par][1][0][lane]); float b2 = xch[par][2][0][lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float ov = xch[par][0][w][lane]; const int oi = __float_as_int(xch[par][1][w][lane]);
        b2 = fmaxf(fmaxf(b2, xch[par][2][w][lane]), fminf(bv, ov));
        if (arg_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
      }
      if (j < np) { st_dev(a.cpart_m + po, bv); st_dev(a.cpart_i + po, bi); st_dev(a.cpart_s + po, b2); }   // (cpart_s: sweep 1's sums were consumed by sweep 1's last workgroup)
    }
  };
  // Both waves of a SIMD run the SAME phase at the same time (MFMAs of tile t, then its VALU work): tools/probes/overlap.hip measures
  // that an MFMA wave and a VALU wave sharing a SIMD take LONGER than one after the other (500 vs 317 cycles per probe body), while
  // two MFMA waves or two VALU waves share it well -- the opposite-phase arrangement tried first was the worst choice available.
  for (int t = 0; t < ntile; ++t) {
    if (t + 1 < ntile && !(ABL & 1)) stage_tile(t + 1);
    flush_cols(t - 1);                // complete since the barrier that ended iteration t - 1
    mma_tile(t);
    epi_tile(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                  // tile t + 1 is staged; every wave is done reading tile t
  }
  if (ntile > 0) flush_cols(ntile - 1);
  stamp(2);

  // ---- rows of this workgroup over its column range: merge the 64 per-lane partials of every row (32 lanes x 2 column halves)
  // through LDS, one partial per (row, column split) -- rpart_a / rpart_b [B][kHeadMaxSplit][npad]
  __syncthreads();
  float* const red0 = reinterpret_cast<float*>(smem);                 // [128 rows][65]
  float* const red1 = red0 + kHeadRows * 65;
  float* const red2 = red1 + kHeadRows * 65;                           // sweep 2: runner-ups
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = lrow0 + (r & 3) + 8 * (r >> 2), slot = 32 * wc + ql;
    red0[row * 65 + slot] = ra[r];
    red1[row * 65 + slot] = SWEEP == 1 ? rbv[r] : __int_as_float(rj[r]);
    if (SWEEP == 2) red2[row * 65 + slot] = rbv[r];
  }
  __syncthreads();
  {
    const int row = tid >> 2, part = tid & 3, i = i0 + row;
    const size_t pr = ((size_t)b * kHeadMaxSplit + sp) * np + i;
    if (SWEEP == 1) {
      float m = -INFINITY, sm = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) lse_merge(m, sm, red0[row * 65 + 16 * part + k], red1[row * 65 + 16 * part + k]);
#pragma unroll
      for (int o = 1; o < 4; o <<= 1) { const float om = __shfl_xor(m, o), os = __shfl_xor(sm, o); lse_merge(m, sm, om, os); }
      if (part == 0 && i < np) {
        if (S == 1) { if (i < n0) { a.rowmax[ro + i] = m; a.rowlog[ro + i] = logf(sm); } }    // single column split: the row is complete
        else { st_dev(a.rpart_a + pr, m); st_dev(a.rpart_b + pr, sm); }
      }
    } else {
      float bv = -INFINITY; int bj = 0x7fffffff; float b2 = -INFINITY;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float ov = red0[row * 65 + 16 * part + k]; const int oj = __float_as_int(red1[row * 65 + 16 * part + k]);
        b2 = fmaxf(fmaxf(b2, red2[row * 65 + 16 * part + k]), fminf(bv, ov));
        if (arg_better(ov, oj, bv, bj)) { bv = ov; bj = oj; }
      }
#pragma unroll
      for (int o = 1; o < 4; o <<= 1) {
        const float ov = __shfl_xor(bv, o); const int oj = __shfl_xor(bj, o); const float o2 = __shfl_xor(b2, o);
        b2 = fmaxf(fmaxf(b2, o2), fminf(bv, ov));
        if (arg_better(ov, oj, bv, bj)) { bv = ov; bj = oj; }
      }
      if (part == 0 && i < np) {
        if (S == 1) { if (i < n0) { st_dev(a.m0 + ro + i, bj); st_dev(a.max0 + ro + i, bv); st_dev(a.max0b + ro + i, b2); } }
        else { st_dev(a.rpart_a + pr, bv); st_dev(a.rpart_b + pr, __int_as_float(bj)); st_dev(a.rpart_c + pr, b2); }
      }
    }
  }
  stamp(3);

  // ---- the last workgroup of the pair finishes the rows and the columns (and, in sweep 2, the matches)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's coherent stores have been acknowledged
  __syncthreads();
  if (tid == 0) {
    const unsigned prev = atomicAdd(&a.tickets[2 * b + (SWEEP - 1)], 1u);
    s_last = prev == (unsigned)(nact * S) - 1u;
    if (s_last) a.tickets[2 * b + (SWEEP - 1)] = 0u;   // ready for the next call on this stream
  }
  __syncthreads();
  stamp(4);
  if (!s_last) return;
  const int nparts = nact;
  // one thread per row / column; the partials are requested together (independent loads: one memory latency per batch) and merged
  // in increasing (column split / row) order
  int* const m1s = reinterpret_cast<int*>(smem) + 64;     // sweep 2: the column arg-max of the pair stays in LDS for the mutual check
  for (int i = tid; S > 1 && i < n0; i += 512) {
    float pa[kHeadMaxSplit], pb[kHeadMaxSplit], pc[kHeadMaxSplit];
#pragma unroll
    for (int e = 0; e < kHeadMaxSplit; ++e) {
      const size_t pr = ((size_t)b * kHeadMaxSplit + min(e, S - 1)) * np + i;
      pa[e] = ld_dev(a.rpart_a + pr); pb[e] = ld_dev(a.rpart_b + pr);
      pc[e] = SWEEP == 2 ? ld_dev(a.rpart_c + pr) : 0.f;
    }
    if (SWEEP == 1) {
      float m = -INFINITY, sm = 0.f;
#pragma unroll
      for (int e = 0; e < kHeadMaxSplit; ++e) if (e < S) lse_merge(m, sm, pa[e], pb[e]);
      a.rowmax[ro + i] = m; a.rowlog[ro + i] = logf(sm);
    } else {
      float bv = -INFINITY; int bj = 0x7fffffff; float b2 = -INFINITY;
#pragma unroll
      for (int e = 0; e < kHeadMaxSplit; ++e) {
        if (e >= S) continue;
        b2 = fmaxf(fmaxf(b2, pc[e]), fminf(bv, pa[e]));
        if (arg_better(pa[e], __float_as_int(pb[e]), bv, bj)) { bv = pa[e]; bj = __float_as_int(pb[e]); }
      }
      st_dev(a.m0 + ro + i, bj); st_dev(a.max0 + ro + i, bv); st_dev(a.max0b + ro + i, b2);
    }
  }
  // The margin certificate (gn_set_certify; kornia consumes `match_indices` as exact integers, pose_node.py:285-297).  With eps a bound on
  // |P - P_exact| over the entries that take part in a decision and L = log(filter_threshold), the pair's match list is the exact
  // arithmetic's list whenever (a) every row whose best score is >= L - eps leads its runner-up by more than 2 eps, (b) the same for every
  // column, and (c) no row's best score lies within eps of L: an exact-arithmetic match (i, j) has P[i][j] > L - eps, so by (a) / (b) it is
  // the row's and the column's arg-max here too, and by (c) it passes the threshold here; conversely a match here keeps both arg-maxima (the
  // gaps exceed twice the error) and stays above L.  (For filter_threshold >= 0.5 -- PoseNode's value -- (a) and (b) follow from (c): a
  // score above 0.5 e^eps leaves less than 0.5 e^-eps for every other entry of its row and column.)  Anything else sets `unc`.
  const bool cert = a.uncert != nullptr && a.cert_eps >= 0.f;
  const bool cert2 = cert && a.uncert_alt != nullptr && a.cert_eps_alt >= 0.f;      // the same three tests for a second eps (HeadArgs::uncert_alt)
  const float Lth = a.threshold > 0.f ? logf(a.threshold) : -INFINITY;
  int unc = 0, unc2 = 0;
  for (int j = tid; j < n1; j += 512) {
    float m = -INFINITY, sm = 0.f; int bi = 0x7fffffff; float m2 = -INFINITY;
    for (int p0 = 0; p0 < nparts; p0 += 16) {     // partials in increasing row order: a strict comparison keeps the lowest row on ties
      float pm[16], ps[16], p2[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const size_t po = ((size_t)b * npart + min(p0 + e, nparts - 1)) * np + j;
        pm[e] = ld_dev(a.cpart_m + po);
        ps[e] = SWEEP == 1 ? ld_dev(a.cpart_s + po) : __int_as_float(ld_dev(a.cpart_i + po));
        p2[e] = SWEEP == 2 ? ld_dev(a.cpart_s + po) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (p0 + e >= nparts) continue;
        if (SWEEP == 1) lse_merge(m, sm, pm[e], ps[e]);
        else {
          m2 = fmaxf(fmaxf(m2, p2[e]), fminf(m, pm[e]));
          if (arg_better(pm[e], __float_as_int(ps[e]), m, bi)) { m = pm[e]; bi = __float_as_int(ps[e]); }
        }
      }
    }
    if (SWEEP == 1) { a.colmax[ro + j] = m; a.collog[ro + j] = logf(sm); }
    else {
      a.m1[ro + j] = bi; m1s[j] = bi;
      if (cert && m >= Lth - a.cert_eps && !(m - m2 > 2.f * a.cert_eps)) unc = 1;     // (b)
      if (cert2 && m >= Lth - a.cert_eps_alt && !(m - m2 > 2.f * a.cert_eps_alt)) unc2 = 1;
    }
  }
  stamp(5);
  if (SWEEP == 1) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // mutual check + threshold + order-preserving compaction (k_compact)
  int* const wcount = reinterpret_cast<int*>(smem);     // (the tile buffers are idle: 64 ints of scratch, then m1s[npad])
  int* const base_s = wcount + 8;
  if (tid == 0) *base_s = 0;
  __syncthreads();
  if (nomatch || (a.ovf != nullptr && *a.ovf != 0u)) {
    if (tid == 0) { a.n_match[b] = 0; if (a.uncert) a.uncert[b] = nomatch ? 0 : 2; if (a.uncert_alt) a.uncert_alt[b] = 0; }   // 2: an activation left the fp16 range -- nothing of this call can be certified
    return;
  }
  for (int ib = 0; ib < n0; ib += 512) {
    const int i = ib + tid;
    bool valid = false; int j = 0; float sc = 0.f;
    if (i < n0) {
      j = ld_dev(a.m0 + ro + i);
      const float best = ld_dev(a.max0 + ro + i);
      sc = expf(best);
      valid = (m1s[j] == i) && (sc > a.threshold);
      if (cert) {
        const float second = ld_dev(a.max0b + ro + i);
        if (best >= Lth - a.cert_eps && !(best - second > 2.f * a.cert_eps)) unc = 1;   // (a)
        if (fabsf(best - Lth) <= a.cert_eps) unc = 1;                                    // (c)
        if (cert2 && ((best >= Lth - a.cert_eps_alt && !(best - second > 2.f * a.cert_eps_alt)) || fabsf(best - Lth) <= a.cert_eps_alt)) unc2 = 1;
      }
    }
    const unsigned long long bal = __ballot(valid);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = *base_s;
    for (int w = 0; w < wave; ++w) off += wcount[w];
    if (valid) {
      const size_t k = (size_t)b * a.kmax + off + before;
      a.idx[2 * k] = i; a.idx[2 * k + 1] = j;
      a.score[k] = sc;
    }
    __syncthreads();
    if (tid == 0) { int tot = 0; for (int w = 0; w < 8; ++w) tot += wcount[w]; *base_s += tot; }
    __syncthreads();
  }
  if (tid == 0) a.n_match[b] = *base_s;
  if (a.uncert) {
    const int any = __syncthreads_or(unc);
    if (tid == 0) a.uncert[b] = (cert && any) ? 1 : 0;
    if (a.uncert_alt) {
      const int any2 = __syncthreads_or(unc2);
      if (tid == 0) a.uncert_alt[b] = (cert2 && any2) ? 1 : 0;
    }
  }
  stamp(6);
}

// grid (kmax/256, B)
__global__ __launch_bounds__(256) void k_gather(GatherArgs a) {
  const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.n_match[b]) return;
  const size_t o = (size_t)b * a.kmax + k;
  const int iq = (int)a.idx[2 * o], ir = (int)a.idx[2 * o + 1];
  const int w = a.kpt_format == GN_KPT_LAF ? 6 : a.kpt_format == GN_KPT_RECORD ? kRecordFloats : 4;
  const int xo = a.kpt_format == GN_KPT_LAF ? 2 : 0, yo = a.kpt_format == GN_KPT_LAF ? 5 : 1;
  const float* kq = a.kpt_q + ((size_t)b * a.stride_q + iq) * w;
  const float* kr = a.kpt_r + ((size_t)b * a.stride_r + ir) * w;
  a.mkp_q[2 * o] = kq[xo]; a.mkp_q[2 * o + 1] = kq[yo];
  const float xr = kr[xo], yr = kr[yo];
  float z = 0.f;
  if (a.dem != nullptr) {  // x, y = floor(mkp_ref).astype(int); z = elevation[y, x]
    int xi = (int)floorf(xr), yi = (int)floorf(yr);
    xi = min(max(xi, 0), a.W - 1); yi = min(max(yi, 0), a.H - 1);
    z = (float)a.dem[((size_t)b * a.H + yi) * a.W + xi];
  }
  a.obj[3 * o] = xr; a.obj[3 * o + 1] = yr; a.obj[3 * o + 2] = z;
}
}  // namespace

int g_head_ablate = 0;   // developer knob 18: timing-only ablations of the first sweep (wrong results)
void launch_match_head_fused(const HeadArgs& a, hipStream_t s) {
  // small batches: split the column tiles of a pair over S workgroups per row block until the grid covers the chip
  const int nrb = a.npad / kHeadRows, ntile = a.npad / 64;
  int S = 1;
  while (S < kHeadMaxSplit && 2 * S <= ntile && nrb * a.B * S < 256) S *= 2;
  const dim3 grid(nrb, a.B, S), block(512);
  if (a.md_f32) {
    hipLaunchKernelGGL((k_head_fused<true, 1>), grid, block, 0, s, a);
    hipLaunchKernelGGL((k_head_fused<true, 2>), grid, block, 0, s, a);
    g_last_kernel = "k_head_fused<true, 2>";
  } else if (g_head_ablate) {
    switch (g_head_ablate) {
      case 1: hipLaunchKernelGGL((k_head_fused<false, 1, 1>), grid, block, 0, s, a); break;
      case 2: hipLaunchKernelGGL((k_head_fused<false, 1, 2>), grid, block, 0, s, a); break;
      case 4: hipLaunchKernelGGL((k_head_fused<false, 1, 4>), grid, block, 0, s, a); break;
      case 8: hipLaunchKernelGGL((k_head_fused<false, 1, 8>), grid, block, 0, s, a); break;
      case 6: hipLaunchKernelGGL((k_head_fused<false, 1, 6>), grid, block, 0, s, a); break;
      default: hipLaunchKernelGGL((k_head_fused<false, 1, 14>), grid, block, 0, s, a); break;
    }
    hipLaunchKernelGGL((k_head_fused<false, 2>), grid, block, 0, s, a);
  } else {
    hipLaunchKernelGGL((k_head_fused<false, 1>), grid, block, 0, s, a);
    hipLaunchKernelGGL((k_head_fused<false, 2>), grid, block, 0, s, a);
    g_last_kernel = "k_head_fused<false, 2>";
  }
}

void launch_match_head(const HeadArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_row_stats, dim3(a.npad / 4, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_col_stats, dim3(a.npad / 64, a.B), dim3(64 * kColGroups), 0, s, a);
  hipLaunchKernelGGL(k_row_argmax, dim3(a.npad / 4, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_col_argmax, dim3(a.npad / 64, a.B), dim3(64 * kColGroups), 0, s, a);
  hipLaunchKernelGGL(k_compact, dim3(a.B), dim3(256), 0, s, a);
}

void launch_gather(const GatherArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_gather, dim3((a.kmax + 255) / 256, a.B), dim3(256), 0, s, a);
}

}  // namespace gn
