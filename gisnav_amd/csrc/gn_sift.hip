// SIFT detector + descriptor on the device: stands in for `cv2.SIFT_create().detectAndCompute(img, None)`
// as GISNav calls it for the reference tile (ros/gisnav/gisnav/core/pose_node.py:122,230-232) and for every camera frame
// (core/twist_node.py:93,227-245).  Follows OpenCV 4.x sift.dispatch.cpp / sift.simd.hpp with the defaults (3 octave
// layers, contrast 0.04, edge 10, sigma 1.6, first octave -1, float descriptors), restated in oracle/sift.py.
//
// Compiled with -ffp-contract=off: every float operation below is a separate IEEE multiply / add / divide in the order the
// oracle writes them, so scale space, keypoints and descriptors agree with it bit for bit.  Stages:
//   k_sift_base (u8 -> f32, 2x bilinear) -> separable Gaussian with BORDER_REFLECT_101, both passes and the DoG level in one
//   launch per level (k_blur_fused: LDS tiles, register windows); the first level of an octave samples level 3 of the
//   octave above directly; every octave of <= 4800 pixels is finished by ONE workgroup in LDS (k_sift_tail) ->
//   k_sift_find (26-neighbour extrema, all octaves and layers in one launch) ->
//   k_sift_refine (quadratic fit, contrast / edge tests, orientation histogram; one wave per candidate) ->
//   k_sift_rank + k_sift_dedup_emit (OpenCV's keypoint order, duplicate removal, first-octave rescale) ->
//   k_sift_descriptor (4x4x8 histogram; producer waves evaluate 64 samples at a time, one wave commits them in OpenCV's order).
#include "gn_common.h"

#include <algorithm>
#include <cmath>

namespace gn {

namespace {
constexpr int kBorder = 5, kMaxInterp = 5, kOriBins = 36, kLayers = 3;
constexpr float kFltEps = 1.1920929e-07f;

__device__ __forceinline__ int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
  }
  return p;
}

// createInitialImage's 2x bilinear upsampling (cv::resize INTER_LINEAR): pixel (dy, dx) of the doubled image from the u8 image g [h][w]
__device__ __forceinline__ float sift_base_sample(const uint8_t* g, int h, int w, int dy, int dx) {
  auto taps = [](int d, int n_src, int& s, int& s1, float& a0, float& a1) {
    float f = (float)d * 0.5f - 0.25f;                 // (d + 0.5) / 2 - 0.5, exact in f32
    s = (int)floorf(f);
    f = f - (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s + 1 >= n_src) { f = 0.f; s = n_src - 1; }
    s1 = min(s + 1, n_src - 1);
    a0 = 1.0f - f; a1 = f;
  };
  int sx, sx1, sy, sy1; float a0, a1, b0, b1;
  taps(dx, w, sx, sx1, a0, a1);
  taps(dy, h, sy, sy1, b0, b1);
  const float h0 = (float)g[(size_t)sy * w + sx] * a0 + (float)g[(size_t)sy * w + sx1] * a1;
  const float h1 = (float)g[(size_t)sy1 * w + sx] * a0 + (float)g[(size_t)sy1 * w + sx1] * a1;
  return h0 * b0 + h1 * b1;
}

// stand-alone form (only used when the initial blur is not the stock 11-tap kernel; otherwise k_blur_fused samples on the fly)
__global__ __launch_bounds__(256) void k_sift_base(const uint8_t* g, int h, int w, float* out, long long out_stride, int* counters) {
  const int b = blockIdx.z;
  g += (size_t)b * h * w; out += (long long)b * out_stride;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 3) counters[4 * b + threadIdx.x] = 0;   // candidate / raw / final counts of this image
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dx >= 2 * w || dy >= 2 * h) return;
  out[(size_t)dy * 2 * w + dx] = sift_base_sample(g, h, w, dy, dx);
}

__global__ __launch_bounds__(256) void k_blur_row(const float* in, float* out, int w, int h, const float* k, int n) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const int r = n >> 1;
  const float* row = in + (size_t)y * w;
  float acc = k[0] * row[reflect101(x - r, w)];
  for (int t = 1; t < n; ++t) acc = acc + k[t] * row[reflect101(x + t - r, w)];
  out[(size_t)y * w + x] = acc;
}

// optional fused DoG: dog = blurred - prev (prev = the level this blur started from), saves one launch per level
__global__ __launch_bounds__(256) void k_blur_col(const float* in, float* out, int w, int h, const float* k, int n, const float* prev, float* dog) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const int r = n >> 1;
  float acc = k[r] * in[(size_t)y * w + x];
  for (int t = 1; t <= r; ++t)
    acc = acc + k[r + t] * (in[(size_t)reflect101(y + t, h) * w + x] + in[(size_t)reflect101(y - t, h) * w + x]);
  out[(size_t)y * w + x] = acc;
  if (dog != nullptr) dog[(size_t)y * w + x] = acc - prev[(size_t)y * w + x];
}

// ---- both passes of one Gaussian level in ONE launch: a workgroup owns a 64 x 32 output tile, stages the input tile
// (+ R columns/rows of BORDER_REFLECT_101 halo) in LDS, row-filters the 32 + 2R rows it needs into a second LDS tile and
// column-filters those.  Every output is the same expression, in the same order, as k_blur_row followed by k_blur_col;
// a thread produces 4 adjacent outputs of a row (8 of a column) from one register window, so an LDS word is read once per
// 4 (8) outputs instead of once per tap.  N (taps) is a template parameter: the pyramid only has 11/13/17/21/27.
// The first level of an octave reads its input straight from level 3 of the octave above (in_step = 2), so the
// half-size base image is never materialised.
constexpr int kFtW = 64, kFtH = 32;   // (64-row tiles -- 1.4x instead of 1.8x halo work in the row pass -- measured 10 % SLOWER on batches: half the blocks per CU)
// BASE: `in` is the u8 camera image [h / 2][w / 2] (images stride_in BYTES apart) and the tile is sampled from its 2x bilinear
// upsampling on the fly -- the initial blur of createInitialImage without ever storing the doubled image; `counters` are zeroed.
template <int N, bool BASE = false, int TH = kFtH>
__global__ __launch_bounds__(256) void k_blur_fused(const float* in, float* out, int w, int h, const float* k, float* dog, int in_step, int in_w, long long stride_in, long long stride_out,
                                                    int* counters = nullptr) {
  const uint8_t* gray = reinterpret_cast<const uint8_t*>(in) + (long long)blockIdx.z * stride_in;
  if (BASE && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 3) counters[4 * blockIdx.z + threadIdx.x] = 0;   // candidate / raw / final counts of this image
  in += (long long)blockIdx.z * stride_in; out += (long long)blockIdx.z * stride_out;   // image blockIdx.z of the batch
  if (dog != nullptr) dog += (long long)blockIdx.z * stride_out;
  constexpr int R = N / 2, ROWS = TH + 2 * R, COLS = (kFtW + 2 * R + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float s_in[ROWS * COLS];
  __shared__ __attribute__((aligned(16))) float s_row[ROWS * kFtW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * kFtW, y0 = blockIdx.y * TH;
  float kk[N];
#pragma unroll
  for (int t = 0; t < N; ++t) kk[t] = k[t];
  {   // stage the tile: all global loads of a thread are issued before the first LDS store (one memory latency, not ~20)
    constexpr int CW = kFtW + 2 * R, TOTAL = ROWS * CW, PER = (TOTAL + 255) / 256;
    float stage[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + 256 * i;
      const int ry = idx / CW, rx = idx - ry * CW;
      if (BASE) stage[i] = idx < TOTAL ? sift_base_sample(gray, h / 2, w / 2, reflect101(y0 - R + ry, h), reflect101(x0 - R + rx, w)) : 0.f;
      else stage[i] = idx < TOTAL ? in[(size_t)(reflect101(y0 - R + ry, h) * in_step) * in_w + reflect101(x0 - R + rx, w) * in_step] : 0.f;   // in_step 2: every second pixel of the octave above
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + 256 * i;
      const int ry = idx / CW, rx = idx - ry * CW;
      if (idx < TOTAL) s_in[ry * COLS + rx] = stage[i];
    }
  }
  __syncthreads();
  for (int item = tid; item < ROWS * (kFtW / 4); item += 256) {
    const int ry = item / (kFtW / 4), xq = item % (kFtW / 4);
    float v[N + 3];
    const float4* src = reinterpret_cast<const float4*>(s_in + ry * COLS + 4 * xq);
#pragma unroll
    for (int j = 0; j < (N + 3 + 3) / 4; ++j) {
      const float4 q = src[j];
      if (4 * j < N + 3) v[4 * j] = q.x;
      if (4 * j + 1 < N + 3) v[4 * j + 1] = q.y;
      if (4 * j + 2 < N + 3) v[4 * j + 2] = q.z;
      if (4 * j + 3 < N + 3) v[4 * j + 3] = q.w;
    }
    float o4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float acc = kk[0] * v[q];
#pragma unroll
      for (int t = 1; t < N; ++t) acc = acc + kk[t] * v[q + t];
      o4[q] = acc;
    }
    *reinterpret_cast<float4*>(s_row + ry * kFtW + 4 * xq) = make_float4(o4[0], o4[1], o4[2], o4[3]);
  }
  __syncthreads();
#pragma unroll 1
  for (int pass = 0; pass < TH / 32; ++pass) {
    const int x = x0 + lane, ly0 = pass * 32 + wave * 8;  // 4 waves x 8 rows per pass of 32 rows
    float v[8 + 2 * R];
#pragma unroll
    for (int j = 0; j < 8 + 2 * R; ++j) v[j] = s_row[(ly0 + j) * kFtW + lane];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float acc = kk[R] * v[q + R];
#pragma unroll
      for (int t = 1; t <= R; ++t) acc = acc + kk[R + t] * (v[q + R + t] + v[q + R - t]);
      const int y = y0 + ly0 + q;
      if (x < w && y < h) {
        out[(size_t)y * w + x] = acc;
        if (dog != nullptr) dog[(size_t)y * w + x] = acc - s_in[(ly0 + q + R) * COLS + lane + R];
      }
    }
  }
}

// ---- the tail of the pyramid (every octave of at most kTailPx pixels, sides <= kTailSide) in ONE launch of one workgroup:
// the levels live in LDS -- two x-padded buffers (current / next level) and one y-padded buffer (row-filtered) whose
// halos hold the BORDER_REFLECT_101 samples, so the tap loops are branch-free and fully unrolled -- and only the results
// (Gaussian levels, DoG levels) go to global memory.  Same expressions, same order, as k_blur_row / k_blur_col.
constexpr int kTailPx = 4800, kTailSide = 128, kTailPad = 13, kTailBuf = kTailPx + 2 * kTailPad * kTailSide;
struct SiftBlurPlan { int off[6], n[6]; };

template <int N>
__device__ __forceinline__ void tail_level(const float* cur, float* tmp, float* nxt, int w, int h, const float* kg, float* g_out, float* d_out) {
  constexpr int R = N / 2, P = kTailPad;
  const int tid = threadIdx.x, wp = w + 2 * P, px = w * h;
  float kk[N];
#pragma unroll
  for (int t = 0; t < N; ++t) kk[t] = kg[t];
  for (int i = tid; i < px; i += 1024) {
    const int y = i / w, x = i - y * w;
    const float* row = cur + y * wp + P + x - R;
    float acc = kk[0] * row[0];
#pragma unroll
    for (int t = 1; t < N; ++t) acc = acc + kk[t] * row[t];
    tmp[(y + P) * w + x] = acc;
  }
  __syncthreads();
  for (int i = tid; i < 2 * R * w; i += 1024) {           // reflected rows above and below
    const int j = i / w, x = i - j * w;
    const int y = j < R ? -1 - j : h + (j - R);
    tmp[(y + P) * w + x] = tmp[(reflect101(y, h) + P) * w + x];
  }
  __syncthreads();
  for (int i = tid; i < px; i += 1024) {
    const int y = i / w, x = i - y * w;
    const float* col = tmp + (y + P) * w + x;
    float acc = kk[R] * col[0];
#pragma unroll
    for (int t = 1; t <= R; ++t) acc = acc + kk[R + t] * (col[t * w] + col[-t * w]);
    nxt[y * wp + P + x] = acc;
    g_out[i] = acc;
    d_out[i] = acc - cur[y * wp + P + x];
  }
  __syncthreads();
}

// reflected columns left and right of an x-padded level (the widest kernel needs kTailPad of them)
__device__ __forceinline__ void tail_pad_x(float* buf, int w, int h) {
  constexpr int P = kTailPad;
  const int wp = w + 2 * P;
  for (int i = threadIdx.x; i < 2 * P * h; i += 1024) {
    const int y = i / (2 * P), j = i - y * (2 * P);
    const int x = j < P ? -1 - j : w + (j - P);
    buf[y * wp + P + x] = buf[y * wp + P + reflect101(x, w)];
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void k_sift_tail(SiftPyramid py, int o_first, const float* dk, SiftBlurPlan plan) {
  extern __shared__ __attribute__((aligned(16))) float s_tail[];
  float* cur = s_tail; float* nxt = s_tail + kTailBuf; float* tmp = s_tail + 2 * kTailBuf;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;                            // one workgroup per image
  for (int o = o_first; o < py.n_oct; ++o) {
    const SiftOctave& oc = py.oct[o];
    const long long boff = (long long)b * oc.stride;   // image b of the batch (a by-value copy of the octave record would go to scratch)
    const int w = oc.w, h = oc.h, px = w * h, wp = w + 2 * kTailPad;
    {   // base level: every second pixel of level 3 of the octave above (written by an earlier launch or by this workgroup)
      const float* src = py.oct[o - 1].gauss[3] + (long long)b * py.oct[o - 1].stride;
      const int ws = py.oct[o - 1].w;
      for (int i = tid; i < px; i += 1024) { const int y = i / w, x = i - y * w; const float v = src[(size_t)(2 * y) * ws + 2 * x]; cur[y * wp + kTailPad + x] = v; oc.gauss[0][boff + i] = v; }
    }
    __syncthreads();
    for (int lvl = 1; lvl < 6; ++lvl) {
      tail_pad_x(cur, w, h);
      const float* kg = dk + plan.off[lvl];
      switch (plan.n[lvl]) {
        case 11: tail_level<11>(cur, tmp, nxt, w, h, kg, oc.gauss[lvl] + boff, oc.dog[lvl - 1] + boff); break;
        case 13: tail_level<13>(cur, tmp, nxt, w, h, kg, oc.gauss[lvl] + boff, oc.dog[lvl - 1] + boff); break;
        case 17: tail_level<17>(cur, tmp, nxt, w, h, kg, oc.gauss[lvl] + boff, oc.dog[lvl - 1] + boff); break;
        case 21: tail_level<21>(cur, tmp, nxt, w, h, kg, oc.gauss[lvl] + boff, oc.dog[lvl - 1] + boff); break;
        default: tail_level<27>(cur, tmp, nxt, w, h, kg, oc.gauss[lvl] + boff, oc.dog[lvl - 1] + boff); break;
      }
      float* t2 = cur; cur = nxt; nxt = t2;
    }
    __syncthreads();                                   // level 3 of this octave is read back from global memory by the next one
  }
}

__global__ __launch_bounds__(256) void k_half_nearest(const float* in, int w, float* out, int w2, int h2) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w2 || y >= h2) return;
  out[(size_t)y * w2 + x] = in[(size_t)(2 * y) * w + 2 * x];
}


// ---- restated math (identical in oracle/sift.py)
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = (float)(0.9997878412794807 * 57.29577951308232), p3 = (float)(-0.3258083974640975 * 57.29577951308232),
              p5 = (float)(0.1555786518463281 * 57.29577951308232), p7 = (float)(-0.04432655554792128 * 57.29577951308232);
  const float ax = fabsf(x), ay = fabsf(y);
  const bool swap = ax < ay;
  const float num = swap ? ax : ay, den = swap ? ay : ax;
  const float c = num / (den + 2.220446049250313e-16f);
  const float c2 = c * c;
  float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  if (swap) a = 90.0f - a;
  if (x < 0) a = 180.0f - a;
  if (y < 0) a = 360.0f - a;
  return a;
}

__device__ __forceinline__ float exp32(float x) {
  const float t = x * 1.4426950408889634f;
  const float n = rintf(t);
  const float f = t - n;
  float p = 0.00015403530393381608f;
  p = p * f + 0.0013333558146428443f;
  p = p * f + 0.009618129107628477f;
  p = p * f + 0.05550410866482158f;
  p = p * f + 0.2402265069591007f;
  p = p * f + 0.6931471805599453f;
  p = p * f + 1.0f;
  return ldexpf(p, (int)n);
}

// Matx33f::solve(b, DECOMP_LU): partial pivoting, float; false when singular
__device__ bool lu_solve3(float A[3][3], float x[3]) {
  for (int i = 0; i < 3; ++i) {
    int k = i;
    for (int j = i + 1; j < 3; ++j) if (fabsf(A[j][i]) > fabsf(A[k][i])) k = j;
    if (fabsf(A[k][i]) < kFltEps) return false;
    if (k != i) {
      for (int c = 0; c < 3; ++c) { const float tmp = A[i][c]; A[i][c] = A[k][c]; A[k][c] = tmp; }
      const float tb = x[i]; x[i] = x[k]; x[k] = tb;
    }
    const float d = -1.0f / A[i][i];
    for (int j = i + 1; j < 3; ++j) {
      const float alpha = A[j][i] * d;
      for (int c = i + 1; c < 3; ++c) A[j][c] = A[j][c] + alpha * A[i][c];
      x[j] = x[j] + alpha * x[i];
    }
  }
  for (int i = 2; i >= 0; --i) {
    float s = x[i];
    for (int c = i + 1; c < 3; ++c) s = s - A[i][c] * x[c];
    x[i] = s / A[i][i];
  }
  return true;
}

// ---- candidates: |DoG| above the threshold and a 26-neighbour extremum
// one launch for all octaves and layers: blockIdx.x runs over the 64 x 32 tiles of every octave back to back
constexpr int kFindRows = 32;
struct SiftFindPlan { int first_tile[kSiftMaxOctaves + 1]; int tiles_x[kSiftMaxOctaves]; };
__global__ __launch_bounds__(256) void k_sift_find(SiftPyramid py, SiftFindPlan plan, float threshold, int4* cand, int* n_cand, int max_cand) {
  const int b = blockIdx.z;
  cand += (size_t)b * max_cand; n_cand += 4 * b;
  int octave = 0;
  while (octave + 1 < py.n_oct && (int)blockIdx.x >= plan.first_tile[octave + 1]) ++octave;
  const SiftOctave& oc = py.oct[octave];
  const long long boff = (long long)b * oc.stride;
  const int tile = blockIdx.x - plan.first_tile[octave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = kBorder + (tile % plan.tiles_x[octave]) * 64 + lane;
  const int rbeg = kBorder + (tile / plan.tiles_x[octave]) * kFindRows + wave * (kFindRows / 4);
  const int rend = min(rbeg + kFindRows / 4, oc.h - kBorder);
  if (c >= oc.w - kBorder || rbeg >= rend) return;
  // a thread walks down a column strip with the 5 x 3 x 3 DoG neighbourhood in registers: 15 loads per row serve the three
  // layers (5 per tested value instead of 1 + 27 from L1 -- this kernel was bound by L1 request rate, not by HBM).
  // (Fetching the left / right neighbours from adjacent lanes with DPP wave shifts instead -- 5 full loads + 10 one-lane edge
  // loads per row -- was 2x SLOWER: the texture path is charged per load instruction, not per lane; four-column strips per
  // thread with one unaligned 16-byte load + two 4-byte loads per row and layer were 1.4x slower as well.)
  const int w = oc.w;
  float win[5][3][3];
  auto load_row = [&](int r, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int l = 0; l < 5; ++l) {
      const float* p = oc.dog[l] + boff + (size_t)r * w + c;
      win[l][slot][0] = p[-1]; win[l][slot][1] = p[0]; win[l][slot][2] = p[1];
    }
  };
  load_row(rbeg - 1, 0);
  load_row(rbeg, 1);
  for (int r = rbeg; r < rend; ++r) {
    load_row(r + 1, 2);
#pragma unroll
    for (int layer = 1; layer <= kLayers; ++layer) {
      const float val = win[layer][1][1];
      if (fabsf(val) > threshold && val != 0.f) {
        bool is_max = val > 0, is_min = val < 0;
#pragma unroll
        for (int l = layer - 1; l <= layer + 1; ++l)
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const float nbv = win[l][dy][dx];
              is_max = is_max && (val >= nbv);
              is_min = is_min && (val <= nbv);
            }
        if (is_max || is_min) {
          const int slot = atomicAdd(n_cand, 1);
          if (slot < max_cand) cand[slot] = make_int4(octave, layer, r, c);
        }
      }
    }
#pragma unroll
    for (int l = 0; l < 5; ++l)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) { win[l][0][dx] = win[l][1][dx]; win[l][1][dx] = win[l][2][dx]; }
  }
}

// ---- adjustLocalExtrema + calcOrientationHist + the peak loop of findScaleSpaceExtrema: one WAVE per candidate.
// The quadratic refinement is scalar work (every lane computes it redundantly); the orientation histogram evaluates 64
// raster positions per step in parallel and commits them in raster order -- bin b of the histogram lives in a register of
// lane b, each sample is broadcast with v_readlane and added by its owner lane -- so the float sums keep OpenCV's order.
__global__ __launch_bounds__(64) void k_sift_refine(SiftPyramid py, const int4* cand, int* counters, int max_cand,
                                                     SiftKeypoint* kp, long long kp_stride, int max_kp) {
  const int lane = threadIdx.x, b = blockIdx.y;
  cand += (size_t)b * max_cand; kp += (long long)b * kp_stride;
  const int* n_cand = counters + 4 * b; int* n_kp = counters + 4 * b + 1;
  const int nc = min(*n_cand, max_cand);
  for (int id = blockIdx.x; id < nc; id += gridDim.x) {
    const int octv = cand[id].x;
    int layer = cand[id].y, r = cand[id].z, c = cand[id].w;
    const SiftOctave& oc = py.oct[octv];
    const long long boff = (long long)b * oc.stride;
    const int rows = oc.h, cols = oc.w;
    const float img_scale = 1.0f / 255.0f, deriv_scale = img_scale * 0.5f, second_scale = img_scale, cross_scale = img_scale * 0.25f;
#define AT(p, rr, cc) (p)[(size_t)(rr) * cols + (cc)]
    float xi = 0.f, xr = 0.f, xc = 0.f;
    int i = 0;
    bool dead = false;                                  // the candidate left the volume or diverged
    for (; i < kMaxInterp; ++i) {
      const float *img = oc.dog[layer] + boff, *prv = oc.dog[layer - 1] + boff, *nxt = oc.dog[layer + 1] + boff;
      float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale, (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                     (AT(nxt, r, c) - AT(prv, r, c)) * deriv_scale};
      const float v2 = AT(img, r, c) * 2.0f;
      const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_scale;
      const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_scale;
      const float dss = (AT(nxt, r, c) + AT(prv, r, c) - v2) * second_scale;
      const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_scale;
      const float dxs = (AT(nxt, r, c + 1) - AT(nxt, r, c - 1) - AT(prv, r, c + 1) + AT(prv, r, c - 1)) * cross_scale;
      const float dys = (AT(nxt, r + 1, c) - AT(nxt, r - 1, c) - AT(prv, r + 1, c) + AT(prv, r - 1, c)) * cross_scale;
      float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
      float X[3] = {dD[0], dD[1], dD[2]};
      if (!lu_solve3(H, X)) { X[0] = X[1] = X[2] = 0.f; }
      xi = -X[2]; xr = -X[1]; xc = -X[0];
      if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
      if ((double)fabsf(xi) > 715827882.0 || (double)fabsf(xr) > 715827882.0 || (double)fabsf(xc) > 715827882.0) { dead = true; break; }
      c += (int)rintf(xc); r += (int)rintf(xr); layer += (int)rintf(xi);
      if (layer < 1 || layer > kLayers || c < kBorder || c >= cols - kBorder || r < kBorder || r >= rows - kBorder) { dead = true; break; }
    }
    if (dead || i >= kMaxInterp) continue;
    float contr;
    {
      const float *img = oc.dog[layer] + boff, *prv = oc.dog[layer - 1] + boff, *nxt = oc.dog[layer + 1] + boff;
      const float d0 = (AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale, d1 = (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                  d2 = (AT(nxt, r, c) - AT(prv, r, c)) * deriv_scale;
      const float t = (d0 * xc + d1 * xr) + d2 * xi;
      contr = AT(img, r, c) * img_scale + t * 0.5f;
      if (fabsf(contr) * (float)kLayers < 0.04f) continue;
      const float v2 = AT(img, r, c) * 2.0f;
      const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_scale;
      const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_scale;
      const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_scale;
      const float tr = dxx + dyy;
      const float det = dxx * dyy - dxy * dxy;
      if (det <= 0.f || (tr * tr) * 10.0f >= (11.0f * 11.0f) * det) continue;
    }
    const float scale = (float)(1 << octv);
    const float kx = ((float)c + xc) * scale, ky = ((float)r + xr) * scale;
    const int octave = octv + (layer << 8) + ((int)rintf((xi + 0.5f) * 255.0f) << 16);
    const float e = ((float)layer + xi) / (float)kLayers;
    const float size = ((1.6f * (float)pow(2.0, (double)e)) * scale) * 2.0f;
    const float response = fabsf(contr);

    // orientation histogram on the Gaussian level of the refined layer
    const float scl_octv = (size * 0.5f) / scale;
    const int radius = (int)rintf(4.5f * scl_octv);
    const float sigma_w = 1.5f * scl_octv;
    const float expf_scale = -1.0f / (2.0f * (sigma_w * sigma_w));
    const float* g = oc.gauss[layer] + boff;
    const int side = 2 * radius + 1, total = side * side;
    float acc = 0.f;                                    // temphist[lane] for lane < 36
    for (int base = 0; base < total; base += 64) {
      const int p = base + lane;
      int mybin = -1; float myval = 0.f;
      if (p < total) {
        const int ii = p / side - radius, jj = p % side - radius;
        const int y = r + ii, x = c + jj;
        if (!(y <= 0 || y >= rows - 1 || x <= 0 || x >= cols - 1)) {
          const float dx = AT(g, y, x + 1) - AT(g, y, x - 1);
          const float dy = AT(g, y - 1, x) - AT(g, y + 1, x);
          const float W = exp32((float)(ii * ii + jj * jj) * expf_scale);
          const float ori = fast_atan2_deg(dy, dx);
          const float mag = sqrtf(dx * dx + dy * dy);
          int bin = (int)rintf((float)(kOriBins / 360.0) * ori);
          if (bin >= kOriBins) bin -= kOriBins;
          if (bin < 0) bin += kOriBins;
          mybin = bin; myval = W * mag;
        }
      }
#pragma unroll
      for (int sidx = 0; sidx < 64; ++sidx) {            // commit in raster order
        const int b = __builtin_amdgcn_readlane(mybin, sidx);
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myval), sidx));
        if (lane == b) acc = acc + v;
      }
    }
    // smoothing, maximum, peaks: lane b < 36 owns bin b
    const int bl = lane < kOriBins ? lane : 0;
    const float m2 = __shfl(acc, (bl + kOriBins - 2) % kOriBins), m1 = __shfl(acc, (bl + kOriBins - 1) % kOriBins);
    const float p1 = __shfl(acc, (bl + 1) % kOriBins), p2 = __shfl(acc, (bl + 2) % kOriBins);
    const float hj = ((m2 + p2) * (float)(1.0 / 16.0) + (m1 + p1) * (float)(4.0 / 16.0)) + acc * (float)(6.0 / 16.0);
    float omax = lane < kOriBins ? hj : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) omax = fmaxf(omax, __shfl_xor(omax, off));
    const float mag_thr = omax * 0.8f;
    const float hl = __shfl(hj, bl > 0 ? bl - 1 : kOriBins - 1), hr = __shfl(hj, bl < kOriBins - 1 ? bl + 1 : 0);
    if (lane < kOriBins && hj > hl && hj > hr && hj >= mag_thr) {
      float bin = (float)lane + (0.5f * (hl - hr)) / ((hl - 2.0f * hj) + hr);
      bin = bin < 0 ? (float)kOriBins + bin : (bin >= (float)kOriBins ? bin - (float)kOriBins : bin);
      float ang = 360.0f - (float)(360.0 / kOriBins) * bin;
      if (fabsf(ang - 360.0f) < kFltEps) ang = 0.f;
      const int slot = atomicAdd(n_kp, 1);
      if (slot < max_kp) { SiftKeypoint k; k.x = kx; k.y = ky; k.size = size; k.angle = ang; k.response = response; k.octave = octave; kp[slot] = k; }
    }
  }
#undef AT
}

// ---- calcSIFTDescriptor: one workgroup of 1 + kDescProducers waves per keypoint.  The 4x4x8 histogram must receive its
// contributions in OpenCV's raster order (float sums do not commute), but everything else about a sample is independent:
// each PRODUCER wave evaluates 64 consecutive raster positions at once (rotation, Gaussian weight, gradient, trilinear
// shares), drops the positions outside the rotated window (ballot/popcount, order kept) and leaves (LDS address, share)
// pairs in LDS.  The COMMITTER wave walks the batches of the previous round in raster order and adds the shares with LDS
// float atomics (ds_add_f32: an exact IEEE f32 add, subnormals included -- tools/probes/lds_fadd.hip): one instruction per
// position, its eight lanes adding the eight shares, which always hit eight different bins.  DS instructions of one wave
// execute in issue order, so every bin sees exactly the serial sequence of additions.
constexpr int kDescProducers = 7;

__global__ __launch_bounds__(64 * (kDescProducers + 1)) void k_sift_descriptor(SiftPyramid py, const SiftKeypoint* kp, long long kp_stride, const int* counters, int max_n, float* desc, long long out_stride) {
  const int b = blockIdx.y;
  kp += (long long)b * kp_stride; desc += (long long)b * out_stride * 128;
  const int n = min(counters[4 * b + 2], max_n);
  constexpr int d = 4, nb = 8, HL = (d + 2) * (d + 2) * (nb + 2), NP = kDescProducers, NT = 64 * (NP + 1);
  __shared__ float hist[HL + 8];                     // [HL..HL+7]: sink for the padding lanes of a partial group of eight
  __shared__ int2 s_pair[2][NP][64 * 8];             // [round parity][producer]: (LDS address of the bin, share bits), 8 per live position
  __shared__ int s_live[2][NP];
  __shared__ float s_dst[128];
  __shared__ float s_nrm;
  const int id = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (id >= n) return;
  const SiftKeypoint k = kp[id];
  int o = k.octave & 255; const int layer = (k.octave >> 8) & 255;
  o = o < 128 ? o : (-128 | o);
  const float kscale = o >= 0 ? 1.0f / (float)(1 << o) : (float)(1 << -o);
  const SiftOctave& oc = py.oct[o + 1];
  const int rows = oc.h, cols = oc.w;
  const float* img = oc.gauss[layer] + (long long)b * oc.stride;
  const float size = k.size * kscale;
  const float ptx = k.x * kscale, pty = k.y * kscale;
  float ori = 360.0f - k.angle;
  if (fabsf(ori - 360.0f) < kFltEps) ori = 0.f;
  const float scl = size * 0.5f;
  const int px = (int)rintf(ptx), pyi = (int)rintf(pty);
  float cos_t = (float)cos((double)ori * (3.141592653589793 / 180.0)), sin_t = (float)sin((double)ori * (3.141592653589793 / 180.0));
  const float bins_per_deg = (float)(nb / 360.0);
  const float exp_scale = -1.0f / (float)(d * d * 0.5);
  const float hist_width = 3.0f * scl;
  int radius = (int)rintf(((hist_width * 1.4142135623730951f) * (float)(d + 1)) * 0.5f);
  radius = min(radius, (int)sqrt((double)cols * cols + (double)rows * rows));
  cos_t = cos_t / hist_width; sin_t = sin_t / hist_width;
  for (int q = tid; q < HL + 8; q += NT) hist[q] = 0.f;
  const uint32_t hist_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)hist;   // LDS byte address of bin 0
  const int side = 2 * radius + 1, total = side * side;
  const int n_rounds = ((total + 63) / 64 + NP - 1) / NP;
  __syncthreads();
  for (int round = 0; round <= n_rounds; ++round) {
    if (wave > 0 && round < n_rounds) {
      // ---- producer: raster positions [base, base + 64)
      const int w = wave - 1;
      const int p = (round * NP + w) * 64 + lane;
      const int ii = p / side - radius, jj = p % side - radius;
      const float fi = (float)ii, fj = (float)jj;
      const float c_rot = fj * cos_t - fi * sin_t;
      const float r_rot = fj * sin_t + fi * cos_t;
      float rbin = r_rot + (float)(d / 2) - 0.5f;
      float cbin = c_rot + (float)(d / 2) - 0.5f;
      const int r = pyi + ii, c = px + jj;
      int idx = -1;
      float v[8];
      if (p < total && rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1) {
        const float dx = img[(size_t)r * cols + c + 1] - img[(size_t)r * cols + c - 1];
        const float dy = img[(size_t)(r - 1) * cols + c] - img[(size_t)(r + 1) * cols + c];
        const float W = exp32((c_rot * c_rot + r_rot * r_rot) * exp_scale);
        const float Ori = fast_atan2_deg(dy, dx);
        const float Mag = sqrtf(dx * dx + dy * dy);
        float obin = (Ori - ori) * bins_per_deg;
        const float mag = Mag * W;
        const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
        int o0 = (int)floorf(obin);
        rbin = rbin - (float)r0; cbin = cbin - (float)c0; obin = obin - (float)o0;
        if (o0 < 0) o0 += nb;
        if (o0 >= nb) o0 -= nb;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
        const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        v[7] = v_rc11 * obin; v[6] = v_rc11 - v[7];
        v[5] = v_rc10 * obin; v[4] = v_rc10 - v[5];
        v[3] = v_rc01 * obin; v[2] = v_rc01 - v[3];
        v[1] = v_rc00 * obin; v[0] = v_rc00 - v[1];
        idx = ((r0 + 1) * (d + 2) + c0 + 1) * (nb + 2) + o0;
      }
      const unsigned long long live_mask = __ballot(idx >= 0);
      const int rank = __popcll(live_mask & ((1ull << lane) - 1ull));
      if (idx >= 0) {
#pragma unroll
        for (int l = 0; l < 8; ++l) {
          const int boff = (l & 1) + ((l >> 1) & 1) * (nb + 2) + (l >> 2) * (d + 2) * (nb + 2);
          s_pair[round & 1][w][rank * 8 + l] = make_int2((int)(hist_lds + 4u * (uint32_t)(idx + boff)), __float_as_int(v[l]));
        }
      }
      if (lane == 0) s_live[round & 1][w] = __popcll(live_mask);
    } else if (wave == 0 && round > 0) {
      // ---- committer: the batches of the previous round, in raster order.  Lane L holds share (L & 7) of live position
      // 8 t + (L >> 3); eight ds_add_f32, each with only the eight lanes of ONE position enabled (exec set by hand: a
      // compiler-generated branch per position costs more than the add itself)
      const int par = (round - 1) & 1;
      for (int w = 0; w < NP; ++w) {
        const int n_live = s_live[par][w];
        const uint32_t pair_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int2*)&s_pair[par][w][lane];
        unsigned long long pr = n_live > 0 ? *reinterpret_cast<const unsigned long long*>(&s_pair[par][w][lane]) : 0ull;
        for (int t = 0; t * 8 < n_live; ++t) {
          const bool live = t * 8 + (lane >> 3) < n_live;
          const uint32_t addr = live ? (uint32_t)pr : hist_lds + 4u * (uint32_t)(HL + (lane & 7));
          const float val = live ? __int_as_float((int)(pr >> 32)) : 0.f;
          const uint32_t next_lds = pair_lds + (uint32_t)(min(t + 1, 7) * 64 * sizeof(int2));
          unsigned long long saved, nxt;
          // the pairs of the next group are fetched ahead of this group's adds (DS ops complete in order: once at most the
          // eight adds are outstanding the read has landed), so the atomic unit never waits for a read round trip.
          // (Measured, tools/probes/lds_atomic_rate.hip: the LDS atomic unit of a CU retires ~1 lane per 3 clocks, shared by
          // all its waves -- a second committer wave, or narrowing exec with v_cmpx to skip border-cell shares, buys nothing.)
          asm volatile(
              "ds_read_b64 %[nx], %[na]\n\t"
              "s_mov_b64 %[sv], exec\n\t"
              "s_mov_b32 exec_hi, 0\n\t"
              "s_mov_b32 exec_lo, 0xff\n\t"        "ds_add_f32 %[a], %[v]\n\t"
              "s_mov_b32 exec_lo, 0xff00\n\t"      "ds_add_f32 %[a], %[v]\n\t"
              "s_mov_b32 exec_lo, 0xff0000\n\t"    "ds_add_f32 %[a], %[v]\n\t"
              "s_mov_b32 exec_lo, 0xff000000\n\t"  "ds_add_f32 %[a], %[v]\n\t"
              "s_mov_b32 exec_lo, 0\n\t"
              "s_mov_b32 exec_hi, 0xff\n\t"        "ds_add_f32 %[a], %[v]\n\t"
              "s_mov_b32 exec_hi, 0xff00\n\t"      "ds_add_f32 %[a], %[v]\n\t"
              "s_mov_b32 exec_hi, 0xff0000\n\t"    "ds_add_f32 %[a], %[v]\n\t"
              "s_mov_b32 exec_hi, 0xff000000\n\t"  "ds_add_f32 %[a], %[v]\n\t"
              "s_mov_b64 exec, %[sv]\n\t"
              "s_waitcnt lgkmcnt(8)\n\t"
              : [sv] "=&s"(saved), [nx] "=&v"(nxt)
              : [a] "v"(addr), [v] "v"(val), [na] "v"(next_lds)
              : "memory");
          pr = nxt;
        }
      }
    }
    __syncthreads();
  }
  if (tid < d * d) {
    const int i = tid / d, j = tid % d;
    const int idx = ((i + 1) * (d + 2) + (j + 1)) * (nb + 2);
    hist[idx] = hist[idx] + hist[idx + nb];
    hist[idx + 1] = hist[idx + 1] + hist[idx + nb + 1];
    for (int q = 0; q < nb; ++q) s_dst[(i * d + j) * nb + q] = hist[idx + q];
  }
  __syncthreads();
  if (tid == 0) {                                     // the two norms are sequential float sums in OpenCV
    float nrm2 = 0.f;
#pragma unroll 4
    for (int q = 0; q < 128; q += 8) {                // (loads batched eight at a time; the sums stay strictly sequential)
      float e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = s_dst[q + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) nrm2 = nrm2 + e[u] * e[u];
    }
    const float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0.f;
#pragma unroll 4
    for (int q = 0; q < 128; q += 8) {
      float e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = fminf(s_dst[q + u], thr);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s_dst[q + u] = e[u]; nrm2 = nrm2 + e[u] * e[u]; }
    }
    s_nrm = 512.0f / fmaxf(sqrtf(nrm2), kFltEps);
  }
  __syncthreads();
  float* dst = desc + (size_t)id * 128;
  if (tid < 128) dst[tid] = fminf(fmaxf(rintf(s_dst[tid] * s_nrm), 0.f), 255.f);   // saturate_cast<uchar>
}
// Throughput variant for batches: ONE wave per keypoint, plain LDS read-add-write by lanes 0..7 per position instead of
// LDS atomics.  A single wave walks its positions ~130 clocks apart (latency of the read-modify-write chain), but 20+
// such waves share a CU and their chains interleave, while the LDS float-atomic unit retires only ~1 lane per 3 clocks for
// the whole CU -- so with thousands of keypoints in flight this is several times the throughput of the atomic committer,
// and with one image the atomic version has the lower latency.  Same arithmetic, same order, same bits.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_sift_descriptor_tp(SiftPyramid py, const SiftKeypoint* kp, long long kp_stride, const int* counters, int max_n, float* desc, long long out_stride) {
  const int b = blockIdx.y;
  kp += (long long)b * kp_stride; desc += (long long)b * out_stride * 128;
  const int n = min(counters[4 * b + 2], max_n);
  constexpr int d = 4, nb = 8, HL = (d + 2) * (d + 2) * (nb + 2), NT = 64;
  __shared__ float hist[HL + 8];
  __shared__ float s_val1[64 * 8];                  // shares, 8 per live position, raster order ...
  __shared__ unsigned short s_bin1[64 * 8];         // ... and their bins (5 KB of LDS per keypoint in all: 32 waves fit a CU)
  __shared__ float s_dst[128];
  __shared__ float s_nrm;
  const int id = blockIdx.x, tid = threadIdx.x, lane = tid;
  if (id >= n) return;
  const SiftKeypoint k = kp[id];
  int o = k.octave & 255; const int layer = (k.octave >> 8) & 255;
  o = o < 128 ? o : (-128 | o);
  const float kscale = o >= 0 ? 1.0f / (float)(1 << o) : (float)(1 << -o);
  const SiftOctave& oc = py.oct[o + 1];
  const int rows = oc.h, cols = oc.w;
  const float* img = oc.gauss[layer] + (long long)b * oc.stride;
  const float size = k.size * kscale;
  const float ptx = k.x * kscale, pty = k.y * kscale;
  float ori = 360.0f - k.angle;
  if (fabsf(ori - 360.0f) < kFltEps) ori = 0.f;
  const float scl = size * 0.5f;
  const int px = (int)rintf(ptx), pyi = (int)rintf(pty);
  float cos_t = (float)cos((double)ori * (3.141592653589793 / 180.0)), sin_t = (float)sin((double)ori * (3.141592653589793 / 180.0));
  const float bins_per_deg = (float)(nb / 360.0);
  const float exp_scale = -1.0f / (float)(d * d * 0.5);
  const float hist_width = 3.0f * scl;
  int radius = (int)rintf(((hist_width * 1.4142135623730951f) * (float)(d + 1)) * 0.5f);
  radius = min(radius, (int)sqrt((double)cols * cols + (double)rows * rows));
  cos_t = cos_t / hist_width; sin_t = sin_t / hist_width;
  for (int q = tid; q < HL + 8; q += NT) hist[q] = 0.f;
  const int side = 2 * radius + 1, total = side * side;
  __syncthreads();
  for (int base = 0; base < total; base += 64) {
    {
      const int p = base + lane;
      const int ii = p / side - radius, jj = p % side - radius;
      const float fi = (float)ii, fj = (float)jj;
      const float c_rot = fj * cos_t - fi * sin_t;
      const float r_rot = fj * sin_t + fi * cos_t;
      float rbin = r_rot + (float)(d / 2) - 0.5f;
      float cbin = c_rot + (float)(d / 2) - 0.5f;
      const int r = pyi + ii, c = px + jj;
      int idx = -1;
      float v[8];
      if (p < total && rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1) {
        const float dx = img[(size_t)r * cols + c + 1] - img[(size_t)r * cols + c - 1];
        const float dy = img[(size_t)(r - 1) * cols + c] - img[(size_t)(r + 1) * cols + c];
        const float W = exp32((c_rot * c_rot + r_rot * r_rot) * exp_scale);
        const float Ori = fast_atan2_deg(dy, dx);
        const float Mag = sqrtf(dx * dx + dy * dy);
        float obin = (Ori - ori) * bins_per_deg;
        const float mag = Mag * W;
        const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
        int o0 = (int)floorf(obin);
        rbin = rbin - (float)r0; cbin = cbin - (float)c0; obin = obin - (float)o0;
        if (o0 < 0) o0 += nb;
        if (o0 >= nb) o0 -= nb;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
        const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        v[7] = v_rc11 * obin; v[6] = v_rc11 - v[7];
        v[5] = v_rc10 * obin; v[4] = v_rc10 - v[5];
        v[3] = v_rc01 * obin; v[2] = v_rc01 - v[3];
        v[1] = v_rc00 * obin; v[0] = v_rc00 - v[1];
        idx = ((r0 + 1) * (d + 2) + c0 + 1) * (nb + 2) + o0;
      }
      const unsigned long long live_mask = __ballot(idx >= 0);
      const int n_live = __popcll(live_mask);
      const int rank = __popcll(live_mask & ((1ull << lane) - 1ull));
      if (idx >= 0) {
#pragma unroll
        for (int l = 0; l < 8; ++l) {
          const int boff = (l & 1) + ((l >> 1) & 1) * (nb + 2) + (l >> 2) * (d + 2) * (nb + 2);
          s_val1[rank * 8 + l] = v[l]; s_bin1[rank * 8 + l] = (unsigned short)(idx + boff);
        }
      }
      asm volatile("" ::: "memory");                 // one wave: DS ops execute in issue order, a compiler fence is all it takes
      // commit in raster order; the eight shares of a position hit eight different bins
      if (lane < 8) {
        int s0 = 0;
        for (; s0 + 4 <= n_live; s0 += 4) {          // shares fetched four positions ahead of the dependent read-add-write chain
          const int b0 = s_bin1[(s0 + 0) * 8 + lane], b1 = s_bin1[(s0 + 1) * 8 + lane], b2 = s_bin1[(s0 + 2) * 8 + lane], b3 = s_bin1[(s0 + 3) * 8 + lane];
          const float v0 = s_val1[(s0 + 0) * 8 + lane], v1 = s_val1[(s0 + 1) * 8 + lane], v2 = s_val1[(s0 + 2) * 8 + lane], v3 = s_val1[(s0 + 3) * 8 + lane];
          hist[b0] = hist[b0] + v0;
          hist[b1] = hist[b1] + v1;
          hist[b2] = hist[b2] + v2;
          hist[b3] = hist[b3] + v3;
        }
        for (; s0 < n_live; ++s0) { const int b0 = s_bin1[s0 * 8 + lane]; hist[b0] = hist[b0] + s_val1[s0 * 8 + lane]; }
      }
      asm volatile("" ::: "memory");
    }
  }
  __syncthreads();
  if (tid < d * d) {
    const int i = tid / d, j = tid % d;
    const int idx = ((i + 1) * (d + 2) + (j + 1)) * (nb + 2);
    hist[idx] = hist[idx] + hist[idx + nb];
    hist[idx + 1] = hist[idx + 1] + hist[idx + nb + 1];
    for (int q = 0; q < nb; ++q) s_dst[(i * d + j) * nb + q] = hist[idx + q];
  }
  __syncthreads();
  if (tid == 0) {                                     // the two norms are sequential float sums in OpenCV
    float nrm2 = 0.f;
#pragma unroll 4
    for (int q = 0; q < 128; q += 8) {                // (loads batched eight at a time; the sums stay strictly sequential)
      float e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = s_dst[q + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) nrm2 = nrm2 + e[u] * e[u];
    }
    const float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0.f;
#pragma unroll 4
    for (int q = 0; q < 128; q += 8) {
      float e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = fminf(s_dst[q + u], thr);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s_dst[q + u] = e[u]; nrm2 = nrm2 + e[u] * e[u]; }
    }
    s_nrm = 512.0f / fmaxf(sqrtf(nrm2), kFltEps);
  }
  __syncthreads();
  float* dst = desc + (size_t)id * 128;
  for (int q = tid; q < 128; q += 64) dst[q] = fminf(fmaxf(rintf(s_dst[q] * s_nrm), 0.f), 255.f);   // saturate_cast<uchar>
}

// ---- KeyPointsFilter::removeDuplicatedSorted + the first-octave rescale, on the device: the raw keypoints (appended in
// arbitrary order by the atomics) are put in OpenCV's KeyPoint_LessThan order (x, y, size descending, angle, response
// descending, octave descending) by a rank sort, then one workgroup drops repeats of (x, y, size, angle), rescales and
// writes the final list -- the order is therefore deterministic whatever the atomics did.
__device__ __forceinline__ bool kp_less(const SiftKeypoint& a, const SiftKeypoint& b) {
  if (a.x != b.x) return a.x < b.x;
  if (a.y != b.y) return a.y < b.y;
  if (a.size != b.size) return a.size > b.size;
  if (a.angle != b.angle) return a.angle < b.angle;
  if (a.response != b.response) return a.response > b.response;
  return a.octave > b.octave;
}

// rank sort: thread i counts the keypoints that precede its own (ties between identical keypoints broken by the raw
// index, so ranks are a permutation) and stores it at that position.  O(n^2) comparisons spread over n / 64 workgroups,
// the list streamed through LDS in tiles of 256 (x as a separate float4-readable array; the full record is only touched
// when x ties) -- no barrier per sorting stage as in a bitonic network.
__global__ __launch_bounds__(256) void k_sift_rank(const SiftKeypoint* kp, long long kp_stride, const int* counters, int max_raw, SiftKeypoint* sorted) {
  kp += (long long)blockIdx.y * kp_stride; sorted += (long long)blockIdx.y * kp_stride;
  const int* n_raw_p = counters + 4 * blockIdx.y + 1;
  __shared__ SiftKeypoint s_tile[256];
  __shared__ __attribute__((aligned(16))) float s_x[256];
  __shared__ int s_rank[4][64];
  const int n = min(*n_raw_p, max_raw);
  if ((int)blockIdx.x * 64 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * 64 + lane;                // the four waves share 64 keypoints and split every tile of 256 four ways
  SiftKeypoint me;
  me.x = __int_as_float(0x7fc00000); me.y = 0.f; me.size = 0.f; me.angle = 0.f; me.response = 0.f; me.octave = 0;   // NaN: precedes / follows nothing
  if (i < n) me = kp[i];
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    __syncthreads();
    if (j0 + tid < n) { const SiftKeypoint t = kp[j0 + tid]; s_tile[tid] = t; s_x[tid] = t.x; }
    else s_x[tid] = __int_as_float(0x7fc00000);
    __syncthreads();
    const float4* xs = reinterpret_cast<const float4*>(s_x + 64 * wave);
    auto visit = [&](float xj, int jl) {
      if (xj < me.x) ++rank;
      else if (xj == me.x && j0 + jl != i) {          // x ties beyond the keypoint itself: several orientations of one extremum
        const SiftKeypoint o = s_tile[jl];
        if (kp_less(o, me) || (!kp_less(me, o) && j0 + jl < i)) ++rank;
      }
    };
#pragma unroll 1
    for (int g = 0; g < 16; ++g) {                     // kept rolled: the kernel is launched cold, code size is latency
      const float4 q = xs[g];                          // same address for every lane: an LDS broadcast
      const int jl = 64 * wave + 4 * g;
      visit(q.x, jl); visit(q.y, jl + 1); visit(q.z, jl + 2); visit(q.w, jl + 3);
    }
  }
  s_rank[wave][lane] = rank;
  __syncthreads();
  if (wave == 0 && i < n) sorted[s_rank[0][lane] + s_rank[1][lane] + s_rank[2][lane] + s_rank[3][lane]] = me;
}

// removeDuplicatedSorted + first-octave rescale + the output arrays, one workgroup over the sorted list.
// More distinct keypoints than the caller's buffers hold (max_out): the call degrades like cv2's `nfeatures` cap
// (KeyPointsFilter::retainBest) instead of failing -- the max_out keypoints of largest response survive (equal responses: the
// earlier one in KeyPoint_LessThan order), still listed in that order; counters[4b + 3] reports how many there were.  The
// response threshold is found by a 4 x 8-bit radix select over the responses' bit patterns (responses are |contrast| >= 0, so
// unsigned order = float order).
__global__ __launch_bounds__(1024) void k_sift_dedup_emit(const SiftKeypoint* kp, SiftKeypoint* out, long long kp_stride, int* counters, int max_raw,
                                                            int max_out, float* kpt_xysa, float* response, int32_t* octave, long long out_stride) {
  const int b = blockIdx.x;
  kp += (long long)b * kp_stride; out += (long long)b * kp_stride;
  kpt_xysa += (long long)b * out_stride * 4;
  if (response) response += (long long)b * out_stride;
  if (octave) octave += (long long)b * out_stride;
  const int* n_raw_p = counters + 4 * b + 1; int* n_out = counters + 4 * b + 2;
  __shared__ int s_wcount[16], s_wcount_eq[16];
  __shared__ int s_base, s_base_eq, s_total, s_need;
  __shared__ unsigned int s_prefix;
  __shared__ int s_hist[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(*n_raw_p, max_raw);
  auto keep_at = [&](int i, SiftKeypoint& q) -> bool {
    if (i >= n) return false;
    q = kp[i];
    if (i == 0) return true;
    const SiftKeypoint p = kp[i - 1];
    return !(p.x == q.x && p.y == q.y && p.size == q.size && p.angle == q.angle);
  };
  // ---- how many distinct keypoints are there?
  if (tid == 0) s_total = 0;
  __syncthreads();
  {
    int mine = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) { SiftKeypoint q; mine += keep_at(i0 + tid, q) ? 1 : 0; }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0 && mine) atomicAdd(&s_total, mine);
  }
  __syncthreads();
  const int total = s_total;
  const bool overflow = total > max_out;
  if (overflow) {   // radix select: bit pattern of the max_out-th largest response, and how many keypoints of exactly that response fit
    if (tid == 0) { s_prefix = 0u; s_need = max_out; }
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) s_hist[tid] = 0;
      __syncthreads();
      const unsigned int prefix = s_prefix;
      for (int i0 = 0; i0 < n; i0 += 1024) {
        SiftKeypoint q;
        if (keep_at(i0 + tid, q)) {
          const unsigned int bits = __float_as_uint(q.response);
          if (shift == 24 || ((bits ^ prefix) >> (shift + 8)) == 0u) atomicAdd(&s_hist[(bits >> shift) & 255u], 1);
        }
      }
      __syncthreads();
      if (tid == 0) {
        int need = s_need, bin = 255;
        for (; bin > 0; --bin) { if (s_hist[bin] >= need) break; need -= s_hist[bin]; }
        s_need = need; s_prefix = prefix | ((unsigned int)bin << shift);
      }
      __syncthreads();
    }
  }
  const unsigned int tbits = overflow ? s_prefix : 0u;
  const int eq_budget = overflow ? s_need : 0;
  if (tid == 0) { s_base = 0; s_base_eq = 0; }
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    SiftKeypoint q;
    bool keep = keep_at(i, q);
    if (overflow) {   // uniform branch: the barriers inside are reached by every thread
      const unsigned int bits = keep ? __float_as_uint(q.response) : 0u;
      const bool eq = keep && bits == tbits;
      const unsigned long long bal_eq = __ballot(eq);
      if (lane == 0) s_wcount_eq[wave] = __popcll(bal_eq);
      __syncthreads();
      int eq_pos = s_base_eq + __popcll(bal_eq & ((1ull << lane) - 1ull));
      for (int wv = 0; wv < wave; ++wv) eq_pos += s_wcount_eq[wv];
      keep = keep && (bits > tbits || (eq && eq_pos < eq_budget));
      __syncthreads();
      if (tid == 0) { int t = 0; for (int wv = 0; wv < 16; ++wv) t += s_wcount_eq[wv]; s_base_eq += t; }
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) s_wcount[wave] = __popcll(bal);
    __syncthreads();
    int pos = s_base + __popcll(bal & ((1ull << lane) - 1ull));
    for (int wv = 0; wv < wave; ++wv) pos += s_wcount[wv];
    if (keep && pos < max_out) {
      q.octave = (q.octave & ~255) | ((q.octave + (-1 & 255)) & 255);        // firstOctave = -1
      q.x = q.x * 0.5f; q.y = q.y * 0.5f; q.size = q.size * 0.5f;
      out[pos] = q;
      kpt_xysa[4 * pos] = q.x; kpt_xysa[4 * pos + 1] = q.y; kpt_xysa[4 * pos + 2] = q.size; kpt_xysa[4 * pos + 3] = q.angle;
      if (response) response[pos] = q.response;
      if (octave) octave[pos] = q.octave;
    }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int wv = 0; wv < 16; ++wv) t += s_wcount[wv]; s_base += t; }
    __syncthreads();
  }
  if (tid == 0) { *n_out = min(s_base, max_out); counters[4 * b + 3] = total; }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host driver
void sift_gaussian_kernel(double sigma, std::vector<float>& k) {
  const int n = (int)std::nearbyint(sigma * 8 + 1) | 1;
  std::vector<double> kd(n);
  const double scale2x = -0.5 / (sigma * sigma);
  double total = 0.0;
  for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; kd[i] = std::exp(scale2x * x * x); total += kd[i]; }
  const double inv = 1.0 / total;
  k.resize(n);
  for (int i = 0; i < n; ++i) k[i] = (float)(kd[i] * inv);
}

static inline dim3 grid2d(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }

static inline bool sift_fused_taps(int n) { return n == 11 || n == 13 || n == 17 || n == 21 || n == 27; }
void sift_blur(int B, long long stride_in, long long stride_out, const float* in, float* tmp, float* out, int w, int h, const float* dk, int n, hipStream_t s,
               float* dog, int in_step, int in_w, float* half_scratch) {
  if (in_w <= 0) in_w = w;
  const dim3 g((w + kFtW - 1) / kFtW, (h + kFtH - 1) / kFtH, B);
  switch (n) {
    case 11: hipLaunchKernelGGL(k_blur_fused<11>, g, dim3(256), 0, s, in, out, w, h, dk, dog, in_step, in_w, stride_in, stride_out, (int*)nullptr); return;
    case 13: hipLaunchKernelGGL(k_blur_fused<13>, g, dim3(256), 0, s, in, out, w, h, dk, dog, in_step, in_w, stride_in, stride_out, (int*)nullptr); return;
    case 17: hipLaunchKernelGGL(k_blur_fused<17>, g, dim3(256), 0, s, in, out, w, h, dk, dog, in_step, in_w, stride_in, stride_out, (int*)nullptr); return;
    case 21: hipLaunchKernelGGL(k_blur_fused<21>, g, dim3(256), 0, s, in, out, w, h, dk, dog, in_step, in_w, stride_in, stride_out, (int*)nullptr); return;
    case 27: hipLaunchKernelGGL(k_blur_fused<27>, g, dim3(256), 0, s, in, out, w, h, dk, dog, in_step, in_w, stride_in, stride_out, (int*)nullptr); return;
    default: break;
  }
  // other sigma: image by image, materialise the half-size image if needed, then two plain passes
  for (int b = 0; b < B; ++b) {
    const float* src = in + (long long)b * stride_in;
    float* dst = out + (long long)b * stride_out;
    float* dg = dog ? dog + (long long)b * stride_out : nullptr;
    if (in_step == 2) {
      float* hs = half_scratch + (long long)b * stride_out;
      hipLaunchKernelGGL(k_half_nearest, grid2d(w, h), dim3(256), 0, s, src, in_w, hs, w, h);
      src = hs;
    }
    hipLaunchKernelGGL(k_blur_row, grid2d(w, h), dim3(256), 0, s, src, tmp, w, h, dk, n);
    hipLaunchKernelGGL(k_blur_col, grid2d(w, h), dim3(256), 0, s, tmp, dst, w, h, dk, n, src, dg);
  }
}
// octaves [o_first, n_oct) in one launch; o_first = sift_tail_first(py) (>= 1; n_oct when the kernel sizes are not the stock ones)
int sift_tail_first(const SiftPyramid& py, const int* ksize) {
  for (int i = 1; i < 6; ++i) if (!sift_fused_taps(ksize[i])) return py.n_oct;
  int o = 1;
  while (o < py.n_oct && (py.oct[o].w * py.oct[o].h > kTailPx || py.oct[o].w > kTailSide || py.oct[o].h > kTailSide)) ++o;
  return o;
}
void sift_tail(const SiftPyramid& py, int B, int o_first, const float* dk, const int* koff, const int* ksize, hipStream_t s) {
  if (o_first >= py.n_oct) return;
  SiftBlurPlan plan;
  for (int i = 0; i < 6; ++i) { plan.off[i] = koff[i]; plan.n[i] = ksize[i]; }
  constexpr int lds = 3 * kTailBuf * (int)sizeof(float);
  static const bool attr_set = (hipFuncSetAttribute(reinterpret_cast<const void*>(k_sift_tail), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess);
  (void)attr_set;
  hipLaunchKernelGGL(k_sift_tail, dim3(B), dim3(1024), lds, s, py, o_first, dk, plan);
}
// createInitialImage for B images: gray [B][h][w] u8 -> level 0 of octave 0 (2h x 2w, blurred to sigma 1.6); zeroes counters[4b + 0..2].
// Stock kernel (11 taps): one launch, the doubled image is sampled on the fly; otherwise k_sift_base into `scratch` + the generic blur.
void sift_base_blur(const uint8_t* gray, int B, int h, int w, float* scratch, float* tmp, float* out, long long out_stride, const float* dk, int n, int* counters, hipStream_t s) {
  if (n == 11) {
    const dim3 g((2 * w + kFtW - 1) / kFtW, (2 * h + kFtH - 1) / kFtH, B);
    hipLaunchKernelGGL((k_blur_fused<11, true>), g, dim3(256), 0, s, reinterpret_cast<const float*>(gray), out, 2 * w, 2 * h, dk, (float*)nullptr, 1, 2 * w,
                       (long long)h * w, out_stride, counters);
    return;
  }
  dim3 g = grid2d(2 * w, 2 * h); g.z = B;
  hipLaunchKernelGGL(k_sift_base, g, dim3(256), 0, s, gray, h, w, scratch, out_stride, counters);
  sift_blur(B, out_stride, out_stride, scratch, tmp, out, 2 * w, 2 * h, dk, n, s);
}
void sift_find(const SiftPyramid& py, int B, float threshold, int4* cand, int* n_cand, int max_cand, hipStream_t s) {
  SiftFindPlan plan;
  int total = 0;
  for (int o = 0; o < kSiftMaxOctaves; ++o) {
    plan.first_tile[o] = total; plan.tiles_x[o] = 1;
    if (o >= py.n_oct) continue;
    const SiftOctave& oc = py.oct[o];
    if (oc.h <= 2 * kBorder || oc.w <= 2 * kBorder) continue;
    const int gx = (oc.w - 2 * kBorder + 63) / 64, gy = (oc.h - 2 * kBorder + kFindRows - 1) / kFindRows;
    plan.tiles_x[o] = gx;
    total += gx * gy;
  }
  plan.first_tile[kSiftMaxOctaves] = total;
  if (total > 0) hipLaunchKernelGGL(k_sift_find, dim3(total, 1, B), dim3(256), 0, s, py, plan, threshold, cand, n_cand, max_cand);
}
void sift_refine(const SiftPyramid& py, int B, const int4* cand, int* counters, int max_cand, SiftKeypoint* kp, long long kp_stride, int max_raw, hipStream_t s) {
  hipLaunchKernelGGL(k_sift_refine, dim3(std::min(max_cand, 4096), B), dim3(64), 0, s, py, cand, counters, max_cand, kp, kp_stride, max_raw);
}
void sift_descriptors(const SiftPyramid& py, int B, const SiftKeypoint* kp_final, long long kp_stride, const int* counters, int max_n, float* desc, long long out_stride, hipStream_t s) {
  // blocks beyond an image's keypoint count exit at once.  One image: latency matters (producer waves + atomic committer);
  // a batch: throughput matters (one wave per keypoint, plain LDS read-add-write)
  if (B >= 4) { hipLaunchKernelGGL(k_sift_descriptor_tp, dim3(max_n, B), dim3(64), 0, s, py, kp_final, kp_stride, counters, max_n, desc, out_stride); return; }
  hipLaunchKernelGGL(k_sift_descriptor, dim3(max_n, B), dim3(64 * (kDescProducers + 1)), 0, s, py, kp_final, kp_stride, counters, max_n, desc, out_stride);
}

// kp: per image [max_raw raw | max_raw sorted | final list], kp_stride records apart
void sift_sort_dedup(int B, SiftKeypoint* kp, long long kp_stride, int* counters, int max_raw, int max_out,
                     float* kpt_xysa, float* response, int32_t* octave, long long out_stride, hipStream_t s) {
  hipLaunchKernelGGL(k_sift_rank, dim3((max_raw + 63) / 64, B), dim3(256), 0, s, kp, kp_stride, counters, max_raw, kp + max_raw);
  hipLaunchKernelGGL(k_sift_dedup_emit, dim3(B), dim3(1024), 0, s, kp + max_raw, kp + 2 * max_raw, kp_stride, counters, max_raw, max_out, kpt_xysa, response, octave, out_stride);
}

}  // namespace gn
