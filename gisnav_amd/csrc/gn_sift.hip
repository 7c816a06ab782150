// SIFT detector + descriptor on the device: stands in for `cv2.SIFT_create().detectAndCompute(img, None)`
// as GISNav calls it for the reference tile (ros/gisnav/gisnav/core/pose_node.py:122,230-232) and for every camera frame
// (core/twist_node.py:93,227-245).  Follows OpenCV 4.x sift.dispatch.cpp / sift.simd.hpp with the defaults (3 octave
// layers, contrast 0.04, edge 10, sigma 1.6, first octave -1, float descriptors), restated in oracle/sift.py.
//
// Compiled with -ffp-contract=off: every float operation below is a separate IEEE multiply / add / divide in the order the
// oracle writes them, so scale space, keypoints and descriptors agree with it bit for bit.  Stages:
//   k_sift_base (u8 -> f32, 2x bilinear) -> separable Gaussian (k_blur_row / k_blur_col, BORDER_REFLECT_101) ->
//   k_half_nearest between octaves -> k_sub (DoG) -> k_sift_find (26-neighbour extrema) ->
//   k_sift_refine (quadratic fit, contrast / edge tests, orientation histogram; one wave per candidate) ->
//   host: sort, duplicate removal, first-octave rescale (KeyPointsFilter::removeDuplicatedSorted) ->
//   k_sift_descriptor (4x4x8 histogram; one wave per keypoint: samples evaluated 64 at a time, committed in OpenCV's order).
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

__global__ __launch_bounds__(256) void k_sift_base(const uint8_t* g, int h, int w, float* out) {
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dx >= 2 * w || dy >= 2 * h) return;
  auto taps = [](int d, int n_src, int& s, int& s1, float& a0, float& a1) {
    float f = (float)(((double)d + 0.5) * 0.5 - 0.5);
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
  out[(size_t)dy * 2 * w + dx] = h0 * b0 + h1 * b1;
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

__global__ __launch_bounds__(256) void k_half_nearest(const float* in, int w, float* out, int w2, int h2) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w2 || y >= h2) return;
  out[(size_t)y * w2 + x] = in[(size_t)(2 * y) * w + 2 * x];
}

__global__ __launch_bounds__(256) void k_sub(const float* a, const float* b, float* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] - b[i];
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
__global__ __launch_bounds__(256) void k_sift_find(SiftOctave oc, int octave, int layer, float threshold, int4* cand, int* n_cand, int max_cand) {
  const int c = kBorder + blockIdx.x * 64 + (threadIdx.x & 63), r = kBorder + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (c >= oc.w - kBorder || r >= oc.h - kBorder) return;
  const float* img = oc.dog[layer];
  const float val = img[(size_t)r * oc.w + c];
  if (!(fabsf(val) > threshold) || val == 0.f) return;
  bool is_max = val > 0, is_min = val < 0;
  for (int l = layer - 1; l <= layer + 1; ++l) {
    const float* p = oc.dog[l];
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const float nb = p[(size_t)(r + dy) * oc.w + c + dx];
        is_max = is_max && (val >= nb);
        is_min = is_min && (val <= nb);
      }
  }
  if (!(is_max || is_min)) return;
  const int slot = atomicAdd(n_cand, 1);
  if (slot < max_cand) cand[slot] = make_int4(octave, layer, r, c);
}

// ---- adjustLocalExtrema + calcOrientationHist + the peak loop of findScaleSpaceExtrema: one WAVE per candidate.
// The quadratic refinement is scalar work (every lane computes it redundantly); the orientation histogram evaluates 64
// raster positions per step in parallel and commits them in raster order -- bin b of the histogram lives in a register of
// lane b, each sample is broadcast with v_readlane and added by its owner lane -- so the float sums keep OpenCV's order.
__global__ __launch_bounds__(64) void k_sift_refine(SiftPyramid py, const int4* cand, const int* n_cand, int max_cand,
                                                     SiftKeypoint* kp, int* n_kp, int max_kp) {
  const int lane = threadIdx.x;
  const int nc = min(*n_cand, max_cand);
  for (int id = blockIdx.x; id < nc; id += gridDim.x) {
    const int octv = cand[id].x;
    int layer = cand[id].y, r = cand[id].z, c = cand[id].w;
    const SiftOctave& oc = py.oct[octv];
    const int rows = oc.h, cols = oc.w;
    const float img_scale = 1.0f / 255.0f, deriv_scale = img_scale * 0.5f, second_scale = img_scale, cross_scale = img_scale * 0.25f;
#define AT(p, rr, cc) (p)[(size_t)(rr) * cols + (cc)]
    float xi = 0.f, xr = 0.f, xc = 0.f;
    int i = 0;
    bool dead = false;                                  // the candidate left the volume or diverged
    for (; i < kMaxInterp; ++i) {
      const float *img = oc.dog[layer], *prv = oc.dog[layer - 1], *nxt = oc.dog[layer + 1];
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
      const float *img = oc.dog[layer], *prv = oc.dog[layer - 1], *nxt = oc.dog[layer + 1];
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
    const float* g = oc.gauss[layer];
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

// ---- calcSIFTDescriptor: one WAVE per keypoint.  The 4x4x8 histogram must receive its contributions in OpenCV's raster
// order (float sums do not commute), but everything else about a sample is independent: the 64 lanes evaluate 64
// consecutive raster positions at once (rotation, Gaussian weight, gradient, trilinear shares) into LDS, then the
// samples are committed one at a time -- lanes 0..7 add the eight shares of one sample, which always hit eight different
// bins -- so the accumulation order is exactly the serial one.
__global__ __launch_bounds__(64) void k_sift_descriptor(SiftPyramid py, const SiftKeypoint* kp, int n, float* desc) {
  constexpr int d = 4, nb = 8, HL = (d + 2) * (d + 2) * (nb + 2);
  __shared__ float hist[HL];
  __shared__ int s_idx[64];
  __shared__ float s_val[8][64];
  __shared__ float s_dst[128];
  __shared__ float s_nrm;
  const int id = blockIdx.x, lane = threadIdx.x;
  if (id >= n) return;
  const SiftKeypoint k = kp[id];
  int o = k.octave & 255; const int layer = (k.octave >> 8) & 255;
  o = o < 128 ? o : (-128 | o);
  const float kscale = o >= 0 ? 1.0f / (float)(1 << o) : (float)(1 << -o);
  const SiftOctave& oc = py.oct[o + 1];
  const int rows = oc.h, cols = oc.w;
  const float* img = oc.gauss[layer];
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
  for (int q = lane; q < HL; q += 64) hist[q] = 0.f;
  const int side = 2 * radius + 1, total = side * side;
  const int off = lane == 0 ? 0 : lane == 1 ? 1 : lane == 2 ? (nb + 2) : lane == 3 ? (nb + 3) : lane == 4 ? (d + 2) * (nb + 2)
                  : lane == 5 ? (d + 2) * (nb + 2) + 1 : lane == 6 ? (d + 3) * (nb + 2) : (d + 3) * (nb + 2) + 1;
  __syncthreads();
  for (int base = 0; base < total; base += 64) {
    const int p = base + lane;
    int idx = -1;
    if (p < total) {
      const int ii = p / side - radius, jj = p % side - radius;
      const float fi = (float)ii, fj = (float)jj;
      const float c_rot = fj * cos_t - fi * sin_t;
      const float r_rot = fj * sin_t + fi * cos_t;
      float rbin = r_rot + (float)(d / 2) - 0.5f;
      float cbin = c_rot + (float)(d / 2) - 0.5f;
      const int r = pyi + ii, c = px + jj;
      if (rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1) {
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
        const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
        const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
        const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
        const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
        idx = ((r0 + 1) * (d + 2) + c0 + 1) * (nb + 2) + o0;
        s_val[0][lane] = v000; s_val[1][lane] = v001; s_val[2][lane] = v010; s_val[3][lane] = v011;
        s_val[4][lane] = v100; s_val[5][lane] = v101; s_val[6][lane] = v110; s_val[7][lane] = v111;
      }
    }
    s_idx[lane] = idx;
    __syncthreads();
    for (int sidx = 0; sidx < 64; ++sidx) {           // commit in raster order; the eight shares of a sample hit distinct bins
      const int ib = s_idx[sidx];
      if (ib < 0) continue;
      if (lane < 8) hist[ib + off] = hist[ib + off] + s_val[lane][sidx];
    }
    __syncthreads();
  }
  if (lane < d * d) {
    const int i = lane / d, j = lane % d;
    const int idx = ((i + 1) * (d + 2) + (j + 1)) * (nb + 2);
    hist[idx] = hist[idx] + hist[idx + nb];
    hist[idx + 1] = hist[idx + 1] + hist[idx + nb + 1];
    for (int q = 0; q < nb; ++q) s_dst[(i * d + j) * nb + q] = hist[idx + q];
  }
  __syncthreads();
  if (lane == 0) {                                    // the two norms are sequential float sums in OpenCV
    float nrm2 = 0.f;
    for (int q = 0; q < 128; ++q) nrm2 = nrm2 + s_dst[q] * s_dst[q];
    const float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0.f;
    for (int q = 0; q < 128; ++q) { const float v = fminf(s_dst[q], thr); s_dst[q] = v; nrm2 = nrm2 + v * v; }
    s_nrm = 512.0f / fmaxf(sqrtf(nrm2), kFltEps);
  }
  __syncthreads();
  float* dst = desc + (size_t)id * 128;
  for (int q = lane; q < 128; q += 64) dst[q] = fminf(fmaxf(rintf(s_dst[q] * s_nrm), 0.f), 255.f);   // saturate_cast<uchar>
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

void sift_blur(const float* in, float* tmp, float* out, int w, int h, const float* dk, int n, hipStream_t s, float* dog) {
  hipLaunchKernelGGL(k_blur_row, grid2d(w, h), dim3(256), 0, s, in, tmp, w, h, dk, n);
  hipLaunchKernelGGL(k_blur_col, grid2d(w, h), dim3(256), 0, s, tmp, out, w, h, dk, n, in, dog);
}
void sift_base(const uint8_t* gray, int h, int w, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_sift_base, grid2d(2 * w, 2 * h), dim3(256), 0, s, gray, h, w, out);
}
void sift_half(const float* in, int w, float* out, int w2, int h2, hipStream_t s) {
  hipLaunchKernelGGL(k_half_nearest, grid2d(w2, h2), dim3(256), 0, s, in, w, out, w2, h2);
}
void sift_sub(const float* a, const float* b, float* out, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_sub, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, out, n);
}
void sift_find(const SiftPyramid& py, float threshold, int4* cand, int* n_cand, int max_cand, hipStream_t s) {
  for (int o = 0; o < py.n_oct; ++o) {
    const SiftOctave& oc = py.oct[o];
    if (oc.h <= 2 * kBorder || oc.w <= 2 * kBorder) continue;
    for (int l = 1; l <= kLayers; ++l)
      hipLaunchKernelGGL(k_sift_find, grid2d(oc.w - 2 * kBorder, oc.h - 2 * kBorder), dim3(256), 0, s, oc, o, l, threshold, cand, n_cand, max_cand);
  }
}
void sift_refine(const SiftPyramid& py, const int4* cand, const int* n_cand, int max_cand, SiftKeypoint* kp, int* n_kp, int max_kp, hipStream_t s) {
  hipLaunchKernelGGL(k_sift_refine, dim3(std::min(max_cand, 4096)), dim3(64), 0, s, py, cand, n_cand, max_cand, kp, n_kp, max_kp);
}
void sift_descriptors(const SiftPyramid& py, const SiftKeypoint* kp, int n, float* desc, float* /*unused*/, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_sift_descriptor, dim3(n), dim3(64), 0, s, py, kp, n, desc);
}

// KeyPointsFilter::removeDuplicatedSorted + the first-octave rescale of SIFT_Impl::detectAndCompute
void sift_sort_dedup(std::vector<SiftKeypoint>& k) {
  std::stable_sort(k.begin(), k.end(), [](const SiftKeypoint& a, const SiftKeypoint& b) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.size != b.size) return a.size > b.size;
    if (a.angle != b.angle) return a.angle < b.angle;
    if (a.response != b.response) return a.response > b.response;
    return a.octave > b.octave;
  });
  std::vector<SiftKeypoint> out;
  for (const SiftKeypoint& q : k) {
    if (!out.empty() && out.back().x == q.x && out.back().y == q.y && out.back().size == q.size && out.back().angle == q.angle) continue;
    out.push_back(q);
  }
  for (SiftKeypoint& q : out) {
    q.octave = (q.octave & ~255) | ((q.octave + (-1 & 255)) & 255);
    q.x = q.x * 0.5f; q.y = q.y * 0.5f; q.size = q.size * 0.5f;
  }
  k.swap(out);
}

}  // namespace gn
