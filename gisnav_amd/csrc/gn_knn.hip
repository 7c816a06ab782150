// Brute-force 2-nearest-neighbour descriptor matching + Lowe ratio test: the visual-odometry matcher of GISNav's
// TwistNode, i.e. `cv2.BFMatcher(crossCheck=False).knnMatch(desc_qry, desc_ref, k=2)` followed
// by `m.distance < 0.7 * n.distance` -- ros/gisnav/gisnav/core/twist_node.py:95,248-267.
//
// OpenCV's batchDistance computes dist = sqrt(sum (a - b)^2) in f32 and keeps the K best per query by strict-less
// insertion in train-index order (ties: lower train index first).  cv2.SIFT descriptors are integer-valued
// (0..255) f32, for which every partial sum below is an exact integer < 2^24, so
//     d^2 = |q|^2 + |r|^2 - 2 q.r
// evaluated in f32 is bit-identical to OpenCV's direct form regardless of summation order; the q.r panel comes from
// the exact-f32 MFMA GEMM (k_gemm_f32_v3, batched over pairs) into the similarity buffer.  For non-integer
// descriptors the result agrees with OpenCV to f32 rounding of d^2 (documented, not bit-exact).
#include "gn_common.h"

namespace gn {

namespace {
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// one wave per token slot: copy the descriptor (zeros in padding) and its squared norm
__global__ __launch_bounds__(256) void k_vo_pack(VoArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bs = blockIdx.y, b = bs >> 1, side = bs & 1;
  const int i = blockIdx.x * 4 + wave;
  const int n = side ? a.n_r[b] : a.n_q[b];
  const int stride = side ? a.stride_r : a.stride_q;
  const size_t tok = (size_t)bs * a.npad + i;
  float d0 = 0.f, d1 = 0.f;
  if (i < n) {
    const float* din = (side ? a.desc_r : a.desc_q) + ((size_t)b * stride + i) * kInDim;
    d0 = din[lane]; d1 = din[lane + 64];
  }
  a.desc[tok * kInDim + lane] = d0;
  a.desc[tok * kInDim + lane + 64] = d1;
  const float s = wave_sum(d0 * d0 + d1 * d1);
  if (lane == 0) a.norm2[tok] = s;
  if (i == 0 && lane == 0) a.nvalid[bs] = n;
}

struct Top2 { float d1, d2; int j1, j2; };
// insert candidate (d, j) into a sorted pair; candidates arrive in increasing j inside one lane, and across lanes
// ties are broken explicitly by the lower train index -- the order OpenCV's strict-less insertion produces
__device__ __forceinline__ void top2_insert(Top2& t, float d, int j) {
  if (d < t.d1 || (d == t.d1 && j < t.j1)) { t.d2 = t.d1; t.j2 = t.j1; t.d1 = d; t.j1 = j; }
  else if (d < t.d2 || (d == t.d2 && j < t.j2)) { t.d2 = d; t.j2 = j; }
}

// one wave per query keypoint
__global__ __launch_bounds__(256) void k_knn2(VoArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, i = blockIdx.x * 4 + wave;
  const int nq = a.nvalid[2 * b], nr = a.nvalid[2 * b + 1];
  if (i >= nq) return;
  const size_t tq = (size_t)(2 * b) * a.npad + i, tr0 = (size_t)(2 * b + 1) * a.npad;
  const float n2q = a.norm2[tq];
  const float* srow = a.sim + ((size_t)b * a.npad + i) * a.npad;
  Top2 t;
  t.d1 = t.d2 = INFINITY; t.j1 = t.j2 = 0x7fffffff;
  for (int j = lane; j < nr; j += 64) {
    const float d = fmaxf((n2q + a.norm2[tr0 + j]) - 2.0f * srow[j], 0.f);
    top2_insert(t, d, j);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float od1 = __shfl_xor(t.d1, off), od2 = __shfl_xor(t.d2, off);
    const int oj1 = __shfl_xor(t.j1, off), oj2 = __shfl_xor(t.j2, off);
    top2_insert(t, od1, oj1);
    top2_insert(t, od2, oj2);
  }
  if (lane == 0) {
    const size_t o = ((size_t)b * a.npad + i) * 2;
    const bool two = nr >= 2;
    const float m = sqrtf(t.d1), n = two ? sqrtf(t.d2) : 0.f;    // DMatch.distance of the best and the second best
    a.nn_idx[o] = nr >= 1 ? t.j1 : -1; a.nn_idx[o + 1] = two ? t.j2 : -1;
    a.nn_dist[o] = m; a.nn_dist[o + 1] = n;
    // Lowe's test exactly as the reference evaluates it: Python floats (f64) `m.distance < 0.7 * n.distance`
    a.good[(size_t)b * a.npad + i] = (two && (double)m < a.ratio * (double)n) ? 1 : 0;
  }
}

// ordered compaction of the queries that passed the ratio test (one workgroup per pair)
__global__ __launch_bounds__(256) void k_vo_compact(VoArgs a) {
  __shared__ int wcount[4];
  __shared__ int base_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = a.nvalid[2 * b], nr = a.nvalid[2 * b + 1];
  if (tid == 0) base_s = 0;
  __syncthreads();
  if (nr < 2) {   // knnMatch(k=2) yields fewer than two neighbours: the reference cannot unpack `for m, n in matches`
    if (tid == 0) a.n_good[b] = 0;
    return;
  }
  for (int i0 = 0; i0 < nq; i0 += 256) {
    const int i = i0 + tid;
    const bool valid = i < nq && a.good[(size_t)b * a.npad + i] != 0;
    const unsigned long long bal = __ballot(valid);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wcount[w];
    if (valid) {
      const size_t k = (size_t)b * a.kmax + off + before;
      const size_t o = ((size_t)b * a.npad + i) * 2;
      a.idx[2 * k] = i; a.idx[2 * k + 1] = a.nn_idx[o];
      a.dist[k] = a.nn_dist[o];
    }
    __syncthreads();
    if (tid == 0) base_s += wcount[0] + wcount[1] + wcount[2] + wcount[3];
    __syncthreads();
  }
  if (tid == 0) a.n_good[b] = base_s;
}
}  // namespace

void launch_vo_pack(const VoArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_vo_pack, dim3(a.npad / 4, a.B * 2), dim3(256), 0, s, a);
}
void launch_vo_knn2(const VoArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_knn2, dim3(a.npad / 4, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_vo_compact, dim3(a.B), dim3(256), 0, s, a);
}

}  // namespace gn
