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
  if (n0 < 2 || n1 < 2 || (a.ovf != nullptr && *a.ovf != 0u)) {  // kornia LightGlueMatcher._no_match; or the f16x2 domain guard tripped (gn_common.h)
    if (tid == 0) a.n_match[b] = 0;
    return;
  }
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

// grid (kmax/256, B)
__global__ __launch_bounds__(256) void k_gather(GatherArgs a) {
  const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.n_match[b]) return;
  const size_t o = (size_t)b * a.kmax + k;
  const int iq = (int)a.idx[2 * o], ir = (int)a.idx[2 * o + 1];
  const int w = a.kpt_format == GN_KPT_LAF ? 6 : 4;
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
