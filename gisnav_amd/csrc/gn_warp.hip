// Reference-raster preparation of GISNav's StereoNode: BGR -> gray, stack with the DEM, rotate
// about the centre and centre-crop to the camera frame -- ros/gisnav/gisnav/core/stereo_node.py:229-262, 292-335
// (`cv2.cvtColor(BGR2GRAY)`, `cv2.getRotationMatrix2D`, `cv2.warpAffine`, numpy crop).
//
// One thread per OUTPUT (cropped) pixel; only the cropped window of the rotated image is ever computed.  The
// arithmetic is OpenCV's fixed-point path restated exactly (oracle/stereo_warp.py): source coordinates in 1/32 pixel
// from `cvRound` of f64 products (no FMA contraction: __dmul_rn / __dadd_rn), 15-bit bilinear weights, taps outside the
// source read 0, (sum + 2^14) >> 15.  HBM-bound byte work: 2 B written and <= 16 B gathered per pixel.
#include "gn_common.h"

namespace gn {

namespace {
__device__ __forceinline__ int gray_of(const uint8_t* p) {   // RGB2Gray<uchar>: B, G, R interleaved
  return (p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14;
}

template <bool FUSED_GRAY>
__global__ __launch_bounds__(256) void k_rotate_crop(WarpArgs a) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= a.crop_w || y >= a.crop_h) return;
  const double xf = (double)(x + a.dx), yf = (double)(y + a.dy);
  // WarpAffineInvoker: adelta / bdelta / X0 / Y0, AB_BITS = 10, round_delta = 16
  const long long ad = (long long)rint(__dmul_rn(__dmul_rn(a.M[0], xf), 1024.0));
  const long long bd = (long long)rint(__dmul_rn(__dmul_rn(a.M[3], xf), 1024.0));
  const long long X0 = (long long)rint(__dmul_rn(__dadd_rn(__dmul_rn(a.M[1], yf), a.M[2]), 1024.0)) + 16;
  const long long Y0 = (long long)rint(__dmul_rn(__dadd_rn(__dmul_rn(a.M[4], yf), a.M[5]), 1024.0)) + 16;
  const long long X = (X0 + ad) >> 5, Y = (Y0 + bd) >> 5;
  long long sxl = X >> 5, syl = Y >> 5;
  sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);     // saturate_cast<short>
  syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
  const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
  int w[4] = {(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32};
  if (fx == 0 && fy == 0) { w[0] = 32767; w[3] = 1; }            // BilinearTab_i[0]: 32768 saturates to short, compensated on tap 3
  int acc0 = 0, acc1 = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int xx = sx + (k & 1), yy = sy + (k >> 1);
    if (xx >= 0 && xx < a.W && yy >= 0 && yy < a.H) {            // BORDER_CONSTANT, value 0
      const size_t p = (size_t)yy * a.W + xx;
      if (FUSED_GRAY) { acc0 += gray_of(a.src0 + 3 * p) * w[k]; acc1 += a.src1[p] * w[k]; }
      else { acc0 += a.src0[2 * p] * w[k]; acc1 += a.src0[2 * p + 1] * w[k]; }
    }
  }
  const int v0 = min(max((acc0 + (1 << 14)) >> 15, 0), 255), v1 = min(max((acc1 + (1 << 14)) >> 15, 0), 255);
  const size_t o = (size_t)y * a.crop_w + x;
  if (a.out1 != nullptr) { a.out0[o] = (uint8_t)v0; a.out1[o] = (uint8_t)v1; }
  else { a.out0[2 * o] = (uint8_t)v0; a.out0[2 * o + 1] = (uint8_t)v1; }
}
}  // namespace

void launch_rotate_crop(const WarpArgs& a, bool fused_gray, hipStream_t s) {
  dim3 grid((a.crop_w + 63) / 64, (a.crop_h + 3) / 4), block(256);
  if (fused_gray) hipLaunchKernelGGL(k_rotate_crop<true>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(k_rotate_crop<false>, grid, block, 0, s, a);
}

}  // namespace gn
