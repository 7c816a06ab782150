// Batched PnP + RANSAC + Rodrigues on gfx950: k_pnp_hyp = one single-wave workgroup per (pair, RANSAC hypothesis) -- the cv::RNG stream does not depend on
// the models, so the <= 10 hypotheses of a pair run concurrently --, k_pnp_refine = one wave per pair (best-model replay, initial guess, LM, Rodrigues).
//
// Stands in for `cv2.solvePnPRansac(obj, img, K, zeros(4,1), useExtrinsicGuess=False,
// iterationsCount=10)` + `cv2.Rodrigues` as called by `compute_pose`
// (ros/gisnav/gisnav/core/_shared.py:104-117).  Algorithm restated from OpenCV 4.x calib3d
// (SURVEY.md Appendix B): cv::RNG(-1) multiply-with-carry stream, 5-point subsets with redraw on
// duplicates, EPnP minimal solver on normalised image points, float32 squared reprojection error
// tested `<= 64.0f`, the `good > max(maxGood, 4)` update with RANSACUpdateNumIters, then
// SOLVEPNP_ITERATIVE on the inliers: planar-homography or 12x12 DLT initialisation followed by
// CvLevMarq (<= 20 iterations, eps = FLT_EPSILON) on (rvec, tvec), all in float64.
//
// Mapping to the wavefront:
//   * RANSAC control flow, RNG and the small dense algebra are wave-uniform; register-resident
//     3x3 / 6x6 kernels are fully unrolled, the 9x9 / 12x12 symmetric eigenproblems run as a
//     lane-parallel cyclic Jacobi on LDS-resident matrices (lane k owns row/column k);
//   * hypothesis scoring strides the K correspondences over the 64 lanes and counts inliers with
//     __ballot + popcount; the DLT / homography normal matrices and the Levenberg-Marquardt
//     J^T J, J^T e, |e|^2 are per-lane partial sums combined with a butterfly of wave shuffles.
// Deviations (documented in DESIGN.md): where
// OpenCV's SVD leaves a basis implementation-defined (rank-deficient 3x3 alignment) the third
// singular pair is completed as u0 x u1, v0 x v1.
#include "gn_common.h"

namespace gn {

namespace {

constexpr double kDblEps = 2.220446049250313e-16;
constexpr double kFltEps = 1.1920928955078125e-07;
constexpr double kDblMin = 2.2250738585072014e-308;

// wave-wide sum, the same value in every lane.  DPP row shifts + row broadcasts (20 VALU instructions, no LDS round trips: the
// ds_bpermute butterfly this replaces cost six dependent LDS latencies per value).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const unsigned hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);   // lanes without a source add +0.0
}
__device__ inline double wsum(double v) {
  v = dpp_add<0x111, 0xf>(v);   // row_shr:1
  v = dpp_add<0x112, 0xf>(v);   // row_shr:2
  v = dpp_add<0x114, 0xf>(v);   // row_shr:4
  v = dpp_add<0x118, 0xf>(v);   // row_shr:8  -> lane 15 of every row of 16 holds the row sum
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3 -> lane 63 holds the total
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, 63), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// 1 / sqrt(x) and 1 / x for normal, well-scaled x: hardware seed + two Newton steps (no range scaling, no special cases)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);      // y (1 + e/2 + 3 e^2 / 8)
  e = __builtin_fma(-x * y, y, 1.0);
  return __builtin_fma(y * e, 0.5, y);
}
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
}
// Every PnP kernel runs single-wave workgroups, so a workgroup barrier is a wave barrier: it orders the
// LDS traffic exchanged between lanes and costs next to nothing.
__device__ inline void wave_sync() { __syncthreads(); }

struct Shared {
  double A[144];
  double V[144];
  int ord[12];
  double L[60];
  double rho[6];
  double alphas[20];
  double us[10];
  double pws[15];
  double cws[12];
  double cam[4];   // fu, fv, uc, vc as OpenCV's epnp constructor takes them from the camera matrix: us = x fu + uc (epnp::init_points)
};

// ------------------------------------------------------------------------------------------------
// register-resident 3x3 symmetric eigen solver (wave-uniform), eigenvalues DESCENDING, V columns.
template <int P, int Q>
__device__ inline void rot3(double a[3][3], double v[3][3]) {
  const double apq = a[P][Q];
  if (fabs(apq) < 1e-300) return;
  const double tau = (a[Q][Q] - a[P][P]) / (2.0 * apq);
  const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
  const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double akp = a[k][P], akq = a[k][Q];
    a[k][P] = c * akp - s * akq; a[k][Q] = s * akp + c * akq;
    const double vkp = v[k][P], vkq = v[k][Q];
    v[k][P] = c * vkp - s * vkq; v[k][Q] = s * vkp + c * vkq;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double apk = a[P][k], aqk = a[Q][k];
    a[P][k] = c * apk - s * aqk; a[Q][k] = s * apk + c * aqk;
  }
}

template <int I, int J>
__device__ inline void cswap3(double w[3], double v[3][3]) {
  if (w[I] < w[J]) {
    const double t = w[I]; w[I] = w[J]; w[J] = t;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double u = v[k][I]; v[k][I] = v[k][J]; v[k][J] = u; }
  }
}

__device__ inline void eig3(const double S[3][3], double w[3], double v[3][3]) {
  double a[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { a[i][j] = S[i][j]; v[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double dg = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-32 * dg || off == 0.0) break;
    rot3<0, 1>(a, v); rot3<0, 2>(a, v); rot3<1, 2>(a, v);
  }
  w[0] = a[0][0]; w[1] = a[1][1]; w[2] = a[2][2];
  cswap3<0, 1>(w, v); cswap3<0, 2>(w, v); cswap3<1, 2>(w, v);
}

__device__ inline void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ inline double det3(const double m[3][3]) {
  return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
         m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

// R = U V^T of the SVD of M (closest orthogonal matrix); singular pair 2 completed by cross
// products when M is rank deficient.
__device__ inline void polar3(const double M[3][3], double R[3][3]) {
  double S[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) S[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
  double w[3], v[3][3];
  eig3(S, w, v);
  double u[3][3];  // u[i] = i-th left singular vector
  double vv[3][3];  // vv[i] = i-th right singular vector
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) vv[i][k] = v[k][i];
  const double s0 = sqrt(fmax(w[0], 0.0));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double si = sqrt(fmax(w[i], 0.0));
    const double inv = si > 0 ? 1.0 / si : 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) u[i][k] = (M[k][0] * vv[i][0] + M[k][1] * vv[i][1] + M[k][2] * vv[i][2]) * inv;
  }
  // re-orthonormalise u1 against u0 (guards nearly equal singular values)
  {
    double n0 = sqrt(u[0][0] * u[0][0] + u[0][1] * u[0][1] + u[0][2] * u[0][2]);
    n0 = n0 > 0 ? 1.0 / n0 : 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) u[0][k] *= n0;
    const double d = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) u[1][k] -= d * u[0][k];
    double n1 = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
    n1 = n1 > 0 ? 1.0 / n1 : 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) u[1][k] *= n1;
  }
  const double s2 = sqrt(fmax(w[2], 0.0));
  if (s2 > 1e-9 * s0) {
    const double inv = 1.0 / s2;
#pragma unroll
    for (int k = 0; k < 3; ++k) u[2][k] = (M[k][0] * vv[2][0] + M[k][1] * vv[2][1] + M[k][2] * vv[2][2]) * inv;
  } else {
    cross3(u[0], u[1], u[2]);
    double t[3];
    cross3(vv[0], vv[1], t);
    vv[2][0] = t[0]; vv[2][1] = t[1]; vv[2][2] = t[2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i][j] = u[0][i] * vv[0][j] + u[1][i] * vv[1][j] + u[2][i] * vv[2][j];
}

// ------------------------------------------------------------------------------------------------
// cvRodrigues2
__device__ inline void rodrigues_v2m(const double r[3], double R[3][3], double J[3][9], bool want_j) {
  const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (theta < kDblEps) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[i][j] = (i == j) ? 1.0 : 0.0;
    if (want_j) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 9; ++k) J[i][k] = 0.0;
      J[0][5] = -1; J[0][7] = 1; J[1][2] = 1; J[1][6] = -1; J[2][1] = -1; J[2][3] = 1;
    }
    return;
  }
  const double c = cos(theta), s = sin(theta), c1 = 1.0 - c, itheta = 1.0 / theta;
  const double rx = r[0] * itheta, ry = r[1] * itheta, rz = r[2] * itheta;
  const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
  const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
  const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k / 3][k % 3] = c * eye[k] + c1 * rrt[k] + s * r_x[k];
  if (!want_j) return;
  const double drrt[3][9] = {{rx + rx, ry, rz, ry, 0, 0, rz, 0, 0},
                             {0, rx, 0, rx, ry + ry, rz, 0, rz, 0},
                             {0, 0, rx, 0, 0, ry, rx, ry, rz + rz}};
  const double d_r_x[3][9] = {{0, 0, 0, 0, 0, -1, 0, 1, 0}, {0, 0, 1, 0, 0, 0, -1, 0, 0}, {0, -1, 0, 1, 0, 0, 0, 0, 0}};
  const double rv[3] = {rx, ry, rz};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double ri = rv[i];
    const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
    const double a3 = (c - s * itheta) * ri, a4 = s * itheta;
#pragma unroll
    for (int k = 0; k < 9; ++k) J[i][k] = a0 * eye[k] + a1 * rrt[k] + a2 * drrt[i][k] + a3 * r_x[k] + a4 * d_r_x[i][k];
  }
}

__device__ inline void rodrigues_m2v(const double Rin[3][3], double r[3]) {
  double R[3][3];
  polar3(Rin, R);
  r[0] = R[2][1] - R[1][2]; r[1] = R[0][2] - R[2][0]; r[2] = R[1][0] - R[0][1];
  const double s = sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25);
  double c = (R[0][0] + R[1][1] + R[2][2] - 1) * 0.5;
  c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
  double theta = acos(c);
  if (s < 1e-5) {
    if (c > 0) { r[0] = r[1] = r[2] = 0.0; return; }
    double t = (R[0][0] + 1) * 0.5;
    r[0] = sqrt(fmax(t, 0.0));
    t = (R[1][1] + 1) * 0.5;
    r[1] = sqrt(fmax(t, 0.0)) * (R[0][1] < 0 ? -1.0 : 1.0);
    t = (R[2][2] + 1) * 0.5;
    r[2] = sqrt(fmax(t, 0.0)) * (R[0][2] < 0 ? -1.0 : 1.0);
    if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && ((R[1][2] > 0) != (r[1] * r[2] > 0))) r[2] = -r[2];
    theta /= sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    r[0] *= theta; r[1] *= theta; r[2] *= theta;
  } else {
    const double vth = theta / (2 * s);
    r[0] *= vth; r[1] *= vth; r[2] *= vth;
  }
}

// ------------------------------------------------------------------------------------------------
// SPD solve (LDL^T, no pivoting), fully unrolled; a vanishing pivot zeroes that unknown.
template <int K>
__device__ inline void spd_solve(const double N[K][K], const double b[K], double x[K]) {
  double L[K][K], d[K], inv[K];
  double scale = 0.0;
#pragma unroll
  for (int i = 0; i < K; ++i) scale = fmax(scale, fabs(N[i][i]));
#pragma unroll
  for (int j = 0; j < K; ++j) {
    double dj = N[j][j];
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k < j) dj -= L[j][k] * L[j][k] * d[k];
    const bool good = dj > 1e-15 * scale;
    d[j] = good ? dj : 0.0;
    inv[j] = good ? 1.0 / dj : 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i)
      if (i > j) {
        double v = N[i][j];
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (k < j) v -= L[i][k] * L[j][k] * d[k];
        L[i][j] = v * inv[j];
      }
  }
  double y[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k < i) v -= L[i][k] * y[k];
    y[i] = v;
  }
#pragma unroll
  for (int i = K - 1; i >= 0; --i) {
    double v = y[i] * inv[i];
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k > i) v -= L[k][i] * x[k];
    x[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Register-resident Jacobi for an N x N symmetric matrix (N = 9, 12), PARALLEL ordering: a sweep is M - 1 rounds (M = N rounded
// up to even) of M / 2 rotations on disjoint index pairs (round-robin tournament); the rotations of one round commute.
//   lane k      (k < N) holds ROW k of A,   lane 16 + k holds ROW k of V   -- both in the same N registers `row`
//   column phase (A <- A J, V <- V J): element pairs (k, p), (k, q) sit in ONE lane; the schedule is static, so after unrolling the
//                rounds p and q are register names and the six (c, s) are wave-uniform (v_readlane from the pair's lane)
//   row phase    (A <- J^T A): lane p and lane q exchange their rows through ds_bpermute; the V lanes pass through (c = 1, s = 0)
// No LDS traffic and no barriers inside a sweep (the LDS version this replaces spent ~1800 cycles per round, this one ~600).
// A comes from LDS and the results go back to LDS in the layout the callers read: eigenvalues on the diagonal of A, eigenvectors
// in the COLUMNS of V, ord[] = column indices by ASCENDING eigenvalue.  Called by the whole (single-wave) block.
__device__ __forceinline__ double bcast_lane(double v, int src_lane /* compile-time constant */) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, src_lane), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src_lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
struct RoundRobin {   // pair i of round r over m players: 0 <= p < q < m
  int p, q;
  constexpr RoundRobin(int m, int r, int i) : p(0), q(0) {
    int a_ = i == 0 ? r % (m - 1) : (r + i) % (m - 1);
    int b_ = i == 0 ? m - 1 : (r - i + (m - 1)) % (m - 1);
    p = a_ < b_ ? a_ : b_; q = a_ < b_ ? b_ : a_;
  }
};
template <int N>
__device__ int sym_eig_reg(double* A, double* V, int* ord, int lane) {   // returns the number of sweeps run
  constexpr int M = (N + 1) & ~1, H = M / 2;
  const bool isA = lane < N, isV = lane >= 16 && lane < 16 + N;
  const int k = isA ? lane : (isV ? lane - 16 : 0);
  double row[N];
#pragma unroll
  for (int j = 0; j < N; ++j) row[j] = isA ? A[k * N + j] : ((isV && j == k) ? 1.0 : 0.0);
  // per round: the lane holding the partner row (self when idle), and whether this lane is the pair's lower index
  int partner[M - 1];
  unsigned lowmask = 0;
#pragma unroll
  for (int r = 0; r < M - 1; ++r) {
    int pt = lane;
#pragma unroll
    for (int i = 0; i < H; ++i) {
      constexpr int dummy = 0; (void)dummy;
      const RoundRobin pr(M, r, i);
      if (pr.q < N) {
        if (lane == pr.p) { pt = pr.q; lowmask |= 1u << r; }
        if (lane == pr.q) pt = pr.p;
      }
    }
    partner[r] = pt;
  }
  double prev_off = 1.7e308;
  int sweep = 0;
#pragma unroll 1
  for (; sweep < 40; ++sweep) {
    double off = 0.0, dg = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double v2 = row[j] * row[j];
      if (isA) { if (j == k) dg += v2; else off += v2; }
    }
    off = wsum(off); dg = wsum(dg);
    // converged, or at the round-off floor: M^T M of a minimal sample is rank deficient, its null-space block keeps rotating
    // noise (off ~ 1e-30 dg) for a dozen sweeps that change nothing; once off is tiny a sweep that fails to halve it is the last
    if (off <= 1e-30 * dg || off == 0.0 || (off <= 1e-20 * dg && off > 0.5 * prev_off)) break;
    prev_off = off;
#pragma unroll
    for (int r = 0; r < M - 1; ++r) {
      // the rotation of this lane's pair, computed identically by both of its lanes from lane p's copy of a_pq
      double apq = 0.0, diag = 0.0;
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const RoundRobin pr(M, r, i);
        if (pr.q < N) apq = lane == pr.p ? row[pr.q] : apq;
      }
#pragma unroll
      for (int j = 0; j < N; ++j) diag = lane == j ? row[j] : diag;
      const int pt = partner[r];
      const bool low = (lowmask >> r) & 1u;
      const double odiag = __shfl(diag, pt), oapq = __shfl(apq, pt);
      const double app = low ? diag : odiag, aqq = low ? odiag : diag;
      apq = low ? apq : oapq;
      // Jacobi angle without the IEEE division / square-root expansions (6 of them, ~100 dependent instructions, were 2/3 of a
      // round): t = sgn(tau) |o| / (|d| + sqrt(d^2 + o^2)) with d = a_qq - a_pp, o = 2 a_pq, tau = d / o;  c = 1 / sqrt(1 + t^2).
      // v_rsq_f64 / v_rcp_f64 seeds + two Newton steps each: relative error of (c, s) a few ulp, c^2 + s^2 = 1 to ~4e-16.
      const double d = aqq - app, o = apq + apq;
      const double h2 = d * d + o * o;
      // (bitwise & / | on the comparisons: the short-circuit forms compile to exec-mask branches, 17 of them per round)
      const bool tneg = (d != 0.0) & ((d < 0.0) != (o < 0.0));
      const double t = (tneg ? -fabs(o) : fabs(o)) * fast_rcp(fabs(d) + h2 * fast_rsqrt(h2));
      double c = fast_rsqrt(1.0 + t * t), sn = t * c;
      const bool rotate = isA & (pt != lane) & (apq * apq > 1e-36 * fabs(app * aqq)) & (fabs(apq) >= 1e-300);
      c = rotate ? c : 1.0; sn = rotate ? sn : 0.0;
      // column phase, A rows and V rows alike
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const RoundRobin pr(M, r, i);
        if (pr.q < N) {
          const double ci = bcast_lane(c, pr.p), si = bcast_lane(sn, pr.p);
          const double rp = row[pr.p], rq = row[pr.q];
          row[pr.p] = ci * rp - si * rq; row[pr.q] = si * rp + ci * rq;
        }
      }
      // row phase: row_p <- c row_p - s row_q, row_q <- s row_p + c row_q  (V lanes and idle lanes: c = 1, s = 0, partner = self)
      // all N exchanges are issued back to back, then consumed in order (one LDS latency per round; the compiler otherwise
      // pairs every exchange with its own wait: ~15 exposed latencies per round, most of the ~1.9 k cycles a round took)
      const double sg = low ? -sn : sn;
      double other[N];
#pragma unroll
      for (int j = 0; j < N; ++j) other[j] = __shfl(row[j], pt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < N; ++j) row[j] = c * row[j] + sg * other[j];
    }
  }
  wave_sync();
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (isA && j == k) A[k * N + j] = row[j];
    if (isV) V[k * N + j] = row[j];
  }
  wave_sync();
  if (lane == 0) {
    for (int i = 0; i < N; ++i) ord[i] = i;
    for (int i = 1; i < N; ++i) {
      const int oi = ord[i];
      const double wi = A[oi * N + oi];
      int j = i - 1;
      while (j >= 0 && A[ord[j] * N + ord[j]] > wi) { ord[j + 1] = ord[j]; --j; }
      ord[j + 1] = oi;
    }
  }
  wave_sync();
  return sweep;
}

// ------------------------------------------------------------------------------------------------
// The eigenvector of the SMALLEST eigenvalue of an N x N symmetric positive semi-definite matrix (N = 9, 12) -- all that the
// homography / DLT initial guesses of solvePnP(ITERATIVE) take from cv::eigen / cv::SVD of L^T L.  Inverse iteration on the
// Cholesky factors of A + eps I instead of a full Jacobi diagonalisation (~8 k cycles instead of 85 - 125 k): the shift (1e-12
// of the mean diagonal, far below lambda_2 of these data matrices) only makes the factorisation safe when round-off left
// lambda_min <= 0; the eigenvectors of A + eps I are those of A, and each iteration shrinks the other components by
// (lambda_min + eps) / (lambda_2 + eps).  Lane k holds row k (L overwrites it); the iterate is replicated in every lane, and every
// element a step needs from another lane is a v_readlane with compile-time lane and register.  Result: unit vector in column 0
// of V, ord[0] = 0 (the layout sym_eig_reg's callers read).  Called by the whole (single-wave) block.
template <int N>
__device__ void smallest_eigvec_reg(const double* A, double* V, int* ord, int lane) {
  const int k = lane < N ? lane : 0;
  double row[N];
#pragma unroll
  for (int j = 0; j < N; ++j) row[j] = A[k * N + j];
  double tr = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) tr += bcast_lane(row[j], j);
  const double eps = tr * (1e-12 / N);
#pragma unroll
  for (int j = 0; j < N; ++j) row[j] = lane == j ? row[j] + eps : row[j];
  double dinv[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double pj = fmax(bcast_lane(row[j], j), eps * 1e-3);
    const double d = fast_rsqrt(pj);            // 1 / L[j][j]
    dinv[j] = d;
    const double lk = row[j] * d;               // L[k][j] (lanes k >= j; lane j: L[j][j])
    row[j] = lk;
#pragma unroll
    for (int m = j + 1; m < N; ++m) row[m] -= lk * bcast_lane(lk, m);
  }
  double x[N];
#pragma unroll
  for (int j = 0; j < N; ++j) x[j] = 1.0 + 0.125 * j;
#pragma unroll 1
  for (int it = 0; it < 5; ++it) {
#pragma unroll
    for (int j = 0; j < N; ++j) {                // L y = x:  L[j][m] is register m of lane j
      double sacc = x[j];
#pragma unroll
      for (int m = 0; m < j; ++m) sacc -= bcast_lane(row[m], j) * x[m];
      x[j] = sacc * dinv[j];
    }
#pragma unroll
    for (int j = N - 1; j >= 0; --j) {           // L^T z = y:  L[m][j] is register j of lane m
      double sacc = x[j];
#pragma unroll
      for (int m = j + 1; m < N; ++m) sacc -= bcast_lane(row[j], m) * x[m];
      x[j] = sacc * dinv[j];
    }
    double n2 = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) n2 += x[j] * x[j];
    const double inv = fast_rsqrt(n2);
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] *= inv;
  }
  wave_sync();
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < N; ++j) V[j * N] = x[j];
    ord[0] = 0;
  }
  wave_sync();
}

// ------------------------------------------------------------------------------------------------
// EPnP on the 5 correspondences staged in sh.pws / sh.us (normalised image coordinates).
__device__ inline void epnp_Ab(const Shared& sh, const double be[4], double A[6][4], double b[6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double* l = &sh.L[i * 10];
    A[i][0] = 2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3];
    A[i][1] = l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3];
    A[i][2] = l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3];
    A[i][3] = l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3];
    b[i] = sh.rho[i] - (l[0] * be[0] * be[0] + l[1] * be[0] * be[1] + l[2] * be[1] * be[1] + l[3] * be[0] * be[2] +
                        l[4] * be[1] * be[2] + l[5] * be[2] * be[2] + l[6] * be[0] * be[3] + l[7] * be[1] * be[3] +
                        l[8] * be[2] * be[3] + l[9] * be[3] * be[3]);
  }
}

// epnp.cpp's own qr_solve (gauss_newton): Householder QR of a 6 x K system WITHOUT pivoting and without a rank test -- a
// nearly dependent column gives a huge step, exactly as in OpenCV (the candidate then loses on reprojection error).  A column
// that is exactly zero makes OpenCV's routine return early with X untouched: x keeps the caller's previous step.  Fully
// unrolled, wave-uniform.
template <int K>
__device__ inline void lsq6(const double Ain[6][K], const double bin[6], double x[K]) {
  double A[6][K], b[6], rd[K];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    b[i] = bin[i];
#pragma unroll
    for (int j = 0; j < K; ++j) A[i][j] = Ain[i][j];
  }
  bool singular = false;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double sigma = 0.0, eta = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (i >= k) { sigma += A[i][k] * A[i][k]; if (i < 5 || k == 5) eta = fmax(eta, fabs(A[i][k])); }   // OpenCV's scan skips the last row
    if (eta == 0.0) singular = true;
    if (singular) { rd[k] = 1.0; continue; }
    const double akk = A[k][k];
    const double alpha = akk > 0 ? -sqrt(sigma) : sqrt(sigma);
    const double beta = 1.0 / (sigma - akk * alpha);
    A[k][k] = akk - alpha;  // v_k; v_i = A[i][k] for i > k
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (j > k) {
        double sdot = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
          if (i >= k) sdot += A[i][k] * A[i][j];
        sdot *= beta;
#pragma unroll
        for (int i = 0; i < 6; ++i)
          if (i >= k) A[i][j] -= sdot * A[i][k];
      }
    {
      double sdot = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (i >= k) sdot += A[i][k] * b[i];
      sdot *= beta;
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (i >= k) b[i] -= sdot * A[i][k];
    }
    rd[k] = alpha;
  }
  if (singular) return;
#pragma unroll
  for (int k = K - 1; k >= 0; --k) {
    double v = b[k];
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (j > k) v -= A[k][j] * x[j];
    x[k] = v / rd[k];
  }
}

// cvSolve(A, b, x, CV_SVD) for a 6 x K system, as find_betas_approx_{1,2,3} call it: minimum-norm least squares through the
// SVD, singular values <= 2 DBL_EPSILON sum(w) treated as zero (cv::SVD::backSubst).  One-sided (Hestenes) Jacobi: the columns
// of A are rotated pairwise until orthogonal, A V = U diag(w).  Fully unrolled per sweep.  The lanes of the wave may hold DIFFERENT
// systems (the three EPnP candidates run side by side): all lanes sweep until none rotates, a converged lane's sweeps are no-ops.
// Trailing all-zero columns are inert (never rotated, w = 0, x = 0): a 6 x 3 or 6 x 4 system padded to K = 5 gives bit-identical x.
template <int K>
__device__ inline void svd_solve6(const double Ain[6][K], const double bin[6], double x[K]) {
  double U[6][K], V[K][K];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) U[i][j] = Ain[i][j];
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) V[i][j] = i == j ? 1.0 : 0.0;
#pragma unroll 1
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < K - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < K; ++q) {
        double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { al += U[i][p] * U[i][p]; be += U[i][q] * U[i][q]; ga += U[i][p] * U[i][q]; }
        // branch-free (the lanes of a wave may hold different systems): c = 1, s = 0 leaves the columns bit-identical
        // |ga| <= 10 DBL_EPSILON sqrt(al be): the orthogonality test of cv::JacobiSVDImpl_ (a tighter one spins on round-off)
        const bool rot = (ga * ga > 4.93e-30 * (al * be)) & (fabs(ga) > 1e-300);
        rotated = rotated | rot;
        const double d = be - al, o = ga + ga, h2 = d * d + o * o;
        const bool tneg = (d != 0.0) & ((d < 0.0) != (o < 0.0));
        const double t = (tneg ? -fabs(o) : fabs(o)) * fast_rcp(fabs(d) + h2 * fast_rsqrt(h2));
        double c = fast_rsqrt(1.0 + t * t), sn = c * t;
        c = rot ? c : 1.0; sn = rot ? sn : 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { const double up = U[i][p], uq = U[i][q]; U[i][p] = c * up - sn * uq; U[i][q] = sn * up + c * uq; }
#pragma unroll
        for (int i = 0; i < K; ++i) { const double vp = V[i][p], vq = V[i][q]; V[i][p] = c * vp - sn * vq; V[i][q] = sn * vp + c * vq; }
      }
    if (!__any(rotated)) break;
  }
  double w[K], wsum = 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double n2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) n2 += U[i][k] * U[i][k];
    w[k] = sqrt(n2); wsum += w[k];
  }
  const double thr = 2.0 * 2.220446049250313e-16 * wsum;
#pragma unroll
  for (int j = 0; j < K; ++j) x[j] = 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (w[k] > thr) {
      double ub = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) ub += U[i][k] * bin[i];
      const double f = ub / (w[k] * w[k]);
#pragma unroll
      for (int j = 0; j < K; ++j) x[j] += V[j][k] * f;
    }
  }
}

__device__ inline double epnp_Rt(const Shared& sh, const double be[4], double R[3][3], double t[3]) {
  double ccs[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double v = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m) v += be[m] * sh.V[(3 * i + k) * 12 + sh.ord[m]];
      ccs[i][k] = v;
    }
  double pcs[5][3];
#pragma unroll
  for (int p = 0; p < 5; ++p)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double v = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) v += sh.alphas[p * 4 + j] * ccs[j][k];
      pcs[p][k] = v;
    }
  if (pcs[0][2] < 0) {
#pragma unroll
    for (int p = 0; p < 5; ++p)
#pragma unroll
      for (int k = 0; k < 3; ++k) pcs[p][k] = -pcs[p][k];
  }
  double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
#pragma unroll
  for (int p = 0; p < 5; ++p)
#pragma unroll
    for (int k = 0; k < 3; ++k) { pc0[k] += pcs[p][k]; pw0[k] += sh.pws[p * 3 + k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) { pc0[k] *= 0.2; pw0[k] *= 0.2; }
  double abt[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double v = 0;
#pragma unroll
      for (int p = 0; p < 5; ++p) v += (pcs[p][j] - pc0[j]) * (sh.pws[p * 3 + k] - pw0[k]);
      abt[j][k] = v;
    }
  polar3(abt, R);
  if (det3(R) < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = pc0[k] - (R[k][0] * pw0[0] + R[k][1] * pw0[1] + R[k][2] * pw0[2]);
  double err = 0;
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    const double* w = &sh.pws[p * 3];
    const double X = R[0][0] * w[0] + R[0][1] * w[1] + R[0][2] * w[2] + t[0];
    const double Y = R[1][0] * w[0] + R[1][1] * w[1] + R[1][2] * w[2] + t[1];
    const double Z = R[2][0] * w[0] + R[2][1] * w[1] + R[2][2] * w[2] + t[2];
    const double du = sh.us[2 * p] - (sh.cam[2] + sh.cam[0] * X / Z), dv = sh.us[2 * p + 1] - (sh.cam[3] + sh.cam[1] * Y / Z);   // epnp::reprojection_error, pixels
    err += sqrt(du * du + dv * dv);
  }
  return err * 0.2;
}

__device__ bool epnp5(Shared& sh, int lane, double Rb[3][3], double tb[3], double* dbg = nullptr, long long* ts = nullptr) {
  auto stamp = [&](int k) __attribute__((always_inline)) { if (ts) ts[k] = (long long)__builtin_amdgcn_s_memtime(); };
  // control points: centroid + PCA axes
  double c0[3] = {0, 0, 0};
#pragma unroll
  for (int p = 0; p < 5; ++p)
#pragma unroll
    for (int k = 0; k < 3; ++k) c0[k] += sh.pws[p * 3 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) c0[k] *= 0.2;
  double S[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double v = 0;
#pragma unroll
      for (int p = 0; p < 5; ++p) v += (sh.pws[p * 3 + i] - c0[i]) * (sh.pws[p * 3 + j] - c0[j]);
      S[i][j] = v;
    }
  double dc[3], uc[3][3];
  eig3(S, dc, uc);
  // canonical axis signs (same rule as the oracle's canonical_axis_signs): the largest-magnitude
  // component of every principal axis is made positive, first such component on ties
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double a0 = fabs(uc[0][i]), a1 = fabs(uc[1][i]), a2 = fabs(uc[2][i]);
    const double lead = (a0 >= a1 && a0 >= a2) ? uc[0][i] : ((a1 >= a2) ? uc[1][i] : uc[2][i]);
    if (lead < 0) { uc[0][i] = -uc[0][i]; uc[1][i] = -uc[1][i]; uc[2][i] = -uc[2][i]; }
  }
  double kk[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) kk[i] = sqrt(fmax(dc[i], 0.0) / 5.0);
  wave_sync();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) sh.cws[k] = c0[k];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) sh.cws[3 * (i + 1) + k] = c0[k] + kk[i] * uc[k][i];
    // barycentric coordinates through the pseudo-inverse of CC = U diag(kk)
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      double a123 = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double v = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) v += uc[k][i] * (sh.pws[p * 3 + k] - c0[k]);
        v = kk[i] > 1e-15 * kk[0] ? v / kk[i] : 0.0;
        sh.alphas[p * 4 + 1 + i] = v;
        a123 += v;
      }
      sh.alphas[p * 4] = 1.0 - a123;
    }
  }
  wave_sync();
  // M^T M (12 x 12), M is 10 x 12 (epnp::fill_M: the rows carry fu / fv, the image points are in pixels)
  const double cfu = sh.cam[0], cfv = sh.cam[1], cuc = sh.cam[2], cvc = sh.cam[3];
  for (int idx = lane; idx < 144; idx += 64) {
    const int r = idx / 12, c = idx % 12;
    const int jr = r / 3, kr = r % 3, jc = c / 3, kc = c % 3;
    double v = 0;
    for (int p = 0; p < 5; ++p) {
      const double ar = sh.alphas[p * 4 + jr], ac = sh.alphas[p * 4 + jc];
      const double u = cuc - sh.us[2 * p], w = cvc - sh.us[2 * p + 1];
      // row 2p:   [a fu, 0, a (uc - u)]   row 2p+1: [0, a fv, a (vc - v)]
      const double m0r = kr == 0 ? ar * cfu : (kr == 2 ? ar * u : 0.0), m0c = kc == 0 ? ac * cfu : (kc == 2 ? ac * u : 0.0);
      const double m1r = kr == 1 ? ar * cfv : (kr == 2 ? ar * w : 0.0), m1c = kc == 1 ? ac * cfv : (kc == 2 ? ac * w : 0.0);
      v += m0r * m0c + m1r * m1c;
    }
    sh.A[idx] = v;
  }
  wave_sync();
  stamp(1);
  const int eig_sweeps = sym_eig_reg<12>(sh.A, sh.V, sh.ord, lane);
  stamp(2);
  if (ts) ts[8] = eig_sweeps;
  // M is 10 x 12, so the two smallest eigenvectors span an exactly 2-D null space whose basis is an
  // artefact of the eigen-solver.  Fix it deterministically (same rule as the oracle's
  // canonical_nullspace): v0 = normalised projection of e_11 onto the null space, v1 = its in-plane
  // complement with component 10 >= 0.
  {
    const int c0 = sh.ord[0], c1 = sh.ord[1];
    const double na = sh.V[11 * 12 + c0], nb = sh.V[11 * 12 + c1];
    const double nrm = sqrt(na * na + nb * nb);
    double w0 = 0, w1 = 0, w1_10 = 0;
    if (nrm >= 1e-12) {
      const double p0 = sh.V[10 * 12 + c0], p1 = sh.V[10 * 12 + c1];
      w1_10 = (-nb * p0 + na * p1) / nrm;
      if (lane < 12) {
        const double x0 = sh.V[lane * 12 + c0], x1 = sh.V[lane * 12 + c1];
        w0 = (na * x0 + nb * x1) / nrm;
        w1 = (-nb * x0 + na * x1) / nrm;
        if (w1_10 < 0) w1 = -w1;
      }
    }
    wave_sync();
    if (nrm >= 1e-12 && lane < 12) { sh.V[lane * 12 + c0] = w0; sh.V[lane * 12 + c1] = w1; }
    wave_sync();
  }
  // L (6 x 10) and rho
  if (lane < 60) {
    const int j = lane / 10, m = lane % 10;
    const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    const int ia[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3}, ib[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3};
    int a_ = 0, b_ = 0, x_ = 0, y_ = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) if (q == j) { a_ = pa[q]; b_ = pb[q]; }
#pragma unroll
    for (int q = 0; q < 10; ++q) if (q == m) { x_ = ia[q]; y_ = ib[q]; }
    const int cx = sh.ord[x_], cy = sh.ord[y_];
    double d = 0;
    for (int k = 0; k < 3; ++k) {
      const double dx = sh.V[(3 * a_ + k) * 12 + cx] - sh.V[(3 * b_ + k) * 12 + cx];
      const double dy = sh.V[(3 * a_ + k) * 12 + cy] - sh.V[(3 * b_ + k) * 12 + cy];
      d += dx * dy;
    }
    sh.L[lane] = (x_ == y_) ? d : 2.0 * d;
    if (m == 0) {
      double rr = 0;
      for (int k = 0; k < 3; ++k) { const double e = sh.cws[3 * a_ + k] - sh.cws[3 * b_ + k]; rr += e * e; }
      sh.rho[j] = rr;
    }
  }
  wave_sync();

  stamp(3);
  // The three beta approximations (find_betas_approx_1 / _2 / _3), their Gauss-Newton refinements and the candidate poses run SIDE BY
  // SIDE: lane 0, 1, 2 carry candidate 0, 1, 2 (lanes >= 3 repeat candidate 2), one pass through the code instead of three.
  double rho[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) rho[i] = sh.rho[i];
  const int cand = lane < 2 ? lane : 2;
  double be[4] = {0, 0, 0, 0};
  {
    // the 6 x 4 (columns 0 1 3 6 of L), 6 x 3 (0 1 2) and 6 x 5 (0 1 2 3 4) systems, zero-padded to 6 x 5
    double A5[6][5], b5[5];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double* l = &sh.L[i * 10];
      A5[i][0] = l[0]; A5[i][1] = l[1];
      A5[i][2] = cand == 0 ? l[3] : l[2];
      A5[i][3] = cand == 0 ? l[6] : (cand == 2 ? l[3] : 0.0);
      A5[i][4] = cand == 2 ? l[4] : 0.0;
    }
    svd_solve6<5>(A5, rho, b5);
    if (cand == 0) {         // betas10 -> [B11 B12 B13 B14]
      if (b5[0] < 0) { be[0] = sqrt(-b5[0]); be[1] = -b5[1] / be[0]; be[2] = -b5[2] / be[0]; be[3] = -b5[3] / be[0]; }
      else { be[0] = sqrt(b5[0]); be[1] = b5[1] / be[0]; be[2] = b5[2] / be[0]; be[3] = b5[3] / be[0]; }
    } else {                 // [B11 B12 B22] and [B11 B12 B22 B13 B23]
      if (b5[0] < 0) { be[0] = sqrt(-b5[0]); be[1] = b5[2] < 0 ? sqrt(-b5[2]) : 0.0; }
      else { be[0] = sqrt(b5[0]); be[1] = b5[2] > 0 ? sqrt(b5[2]) : 0.0; }
      if (b5[1] < 0) be[0] = -be[0];
      if (cand == 2) be[2] = b5[3] / be[0];
    }
  }
  stamp(4);
  double x[4] = {0, 0, 0, 0};      // the step persists across iterations (qr_solve leaves X untouched on an exactly singular system)
#pragma unroll 1
  for (int it = 0; it < 5; ++it) {  // gauss_newton
    double A[6][4], b[6];
    epnp_Ab(sh, be, A, b);
    lsq6<4>(A, b, x);
#pragma unroll
    for (int k = 0; k < 4; ++k) be[k] += x[k];
  }
  stamp(5);
  double R[3][3], t[3];
  const double err = epnp_Rt(sh, be, R, t);
  stamp(6);
  if (dbg && lane < 3) {
    dbg[12 + lane] = err;
#pragma unroll
    for (int k = 0; k < 4; ++k) dbg[15 + 4 * lane + k] = be[k];
  }
  // the sequential choice `if (err < best) best = cand` over candidates 0, 1, 2 (NaN never wins)
  int bestc = -1; double best_err = 0;
#pragma unroll
  for (int cnd = 0; cnd < 3; ++cnd) {
    const double e = bcast_lane(err, cnd);
    if (e == e && (bestc < 0 || e < best_err)) { bestc = cnd; best_err = e; }
  }
  const bool have = bestc >= 0;
  const int src = have ? bestc : 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) { tb[i] = __shfl(t[i], src);
#pragma unroll
    for (int j = 0; j < 3; ++j) Rb[i][j] = __shfl(R[i][j], src); }
  bool fin = have;
#pragma unroll
  for (int i = 0; i < 3; ++i) { fin = fin && isfinite(tb[i]);
#pragma unroll
    for (int j = 0; j < 3; ++j) fin = fin && isfinite(Rb[i][j]); }
  return fin;
}

// ------------------------------------------------------------------------------------------------
struct Cam { double fx, fy, cx, cy; };

// |e|^2, and optionally J^T J (upper, 21) and J^T e (6), over the n (compacted inlier) points
__device__ inline double lm_accumulate(const float* obj, const float* img, int n, int lane,
                                       const Cam& cam, const double p[6], bool want_j, double JtJ[21], double Jte[6]) {
  double R[3][3], dR[3][9];
  rodrigues_v2m(p, R, dR, want_j);
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  for (int i = lane; i < n; i += 64) {
    const double M0 = obj[3 * i], M1 = obj[3 * i + 1], M2 = obj[3 * i + 2];
    const double X = R[0][0] * M0 + R[0][1] * M1 + R[0][2] * M2 + p[3];
    const double Y = R[1][0] * M0 + R[1][1] * M1 + R[1][2] * M2 + p[4];
    const double Z = R[2][0] * M0 + R[2][1] * M1 + R[2][2] * M2 + p[5];
    const double z = Z != 0 ? 1.0 / Z : 1.0;
    const double x = X * z, y = Y * z;
    const double ex = x * cam.fx + cam.cx - (double)img[2 * i], ey = y * cam.fy + cam.cy - (double)img[2 * i + 1];
    acc[27] += ex * ex + ey * ey;
    if (want_j) {
      double jx[6], jy[6];
      // d(x,y)/dX
      const double ax0 = cam.fx * z, ax2 = -cam.fx * x * z;
      const double ay1 = cam.fy * z, ay2 = -cam.fy * y * z;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double dX = dR[k][0] * M0 + dR[k][1] * M1 + dR[k][2] * M2;
        const double dY = dR[k][3] * M0 + dR[k][4] * M1 + dR[k][5] * M2;
        const double dZ = dR[k][6] * M0 + dR[k][7] * M1 + dR[k][8] * M2;
        jx[k] = ax0 * dX + ax2 * dZ;
        jy[k] = ay1 * dY + ay2 * dZ;
      }
      jx[3] = ax0; jx[4] = 0.0; jx[5] = ax2;
      jy[3] = 0.0; jy[4] = ay1; jy[5] = ay2;
      int q = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = r; c < 6; ++c) { acc[q] += jx[r] * jx[c] + jy[r] * jy[c]; ++q; }
        acc[21 + r] += jx[r] * ex + jy[r] * ey;
      }
    }
  }
  const int lo = want_j ? 0 : 27;
#pragma unroll
  for (int k = 0; k < 28; ++k)
    if (k >= lo) acc[k] = wsum(acc[k]);
  if (want_j) {
#pragma unroll
    for (int k = 0; k < 21; ++k) JtJ[k] = acc[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) Jte[k] = acc[21 + k];
  }
  return acc[27];
}

__device__ inline void lm_step(const double JtJ[21], const double Jte[6], const double prev[6], int lambda_lg10, double out[6]) {
  const double lam = exp(lambda_lg10 * 2.302585092994046);
  double N[6][6];
  int q = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) { N[r][c] = JtJ[q]; N[c][r] = JtJ[q]; ++q; }
#pragma unroll
  for (int r = 0; r < 6; ++r) N[r][r] *= 1.0 + lam;
  double d[6];
  spd_solve<6>(N, Jte, d);
#pragma unroll
  for (int k = 0; k < 6; ++k) out[k] = prev[k] - d[k];
}

// CvLevMarq(6, 2n, TermCriteria(20, FLT_EPSILON)) driven as in cvFindExtrinsicCameraParams2
__device__ void levmarq_pose(const float* obj, const float* img, int n, int lane, const Cam& cam, double p[6]) {
  double prev[6], JtJ[21], Jte[6];
  int lambda_lg10 = -3, iters = 0;
  double prev_err = 0.0;
  double e2 = lm_accumulate(obj, img, n, lane, cam, p, true, JtJ, Jte);
  for (;;) {
#pragma unroll
    for (int k = 0; k < 6; ++k) prev[k] = p[k];
    lm_step(JtJ, Jte, prev, lambda_lg10, p);
    if (iters == 0) prev_err = sqrt(e2);
    double err_norm;
    for (;;) {
      err_norm = sqrt(lm_accumulate(obj, img, n, lane, cam, p, false, JtJ, Jte));
      if (err_norm > prev_err) {
        if (++lambda_lg10 <= 16) { lm_step(JtJ, Jte, prev, lambda_lg10, p); continue; }
      }
      break;
    }
    lambda_lg10 = lambda_lg10 - 1 > -16 ? lambda_lg10 - 1 : -16;
    ++iters;
    double dn = 0, pn = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { dn += (p[k] - prev[k]) * (p[k] - prev[k]); pn += prev[k] * prev[k]; }
    const double rel = sqrt(dn) / (pn > 0 ? sqrt(pn) : 1.0);
    if (iters >= 20 || rel < kFltEps) break;
    prev_err = err_norm;
    e2 = lm_accumulate(obj, img, n, lane, cam, p, true, JtJ, Jte);
  }
}

__device__ inline int ransac_update_iters(double p, double ep, int model_points, int max_iters) {
  p = fmin(fmax(p, 0.0), 1.0); ep = fmin(fmax(ep, 0.0), 1.0);
  double num = fmax(1.0 - p, kDblMin);
  double denom = 1.0 - pow(1.0 - ep, (double)model_points);
  if (denom < kDblMin) return 0;
  num = log(num); denom = log(denom);
  return (denom >= 0 || -num >= max_iters * (-denom)) ? max_iters : (int)rint(num / denom);
}

__device__ inline unsigned rng_next(unsigned long long& st) {
  st = (unsigned long long)(unsigned)st * 4164903690ull + (st >> 32);
  return (unsigned)st;
}

constexpr int kMaxHyp = 16;   // RANSAC hypotheses evaluated concurrently, one wavefront each


// planar-structure initial guess of cvFindExtrinsicCameraParams2 (homography from the model plane)
__device__ __noinline__ bool pnp_init_planar(Shared& sh, const float* obj, const float* img, int ninl, int lane, const Cam& cam, const double mc[3], const double Vc[3][3], double p[6]) {
  bool init_ok = true;
    // planar structure: homography from the model plane to the normalised image
    double Rt[3][3];  // rows = principal axes (V^T)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Rt[i][j] = Vc[j][i];
    if (Rt[0][2] * Rt[0][2] + Rt[1][2] * Rt[1][2] < 1e-10) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Rt[i][j] = (i == j) ? 1.0 : 0.0;
    }
    if (det3(Rt) < 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Rt[i][j] = -Rt[i][j];
    }
    double Tt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) Tt[i] = -(Rt[i][0] * mc[0] + Rt[i][1] * mc[1] + Rt[i][2] * mc[2]);
    // findHomography(method 0) works on float32 copies of both point sets
    auto plane_xy = [&](int i, double& X, double& Y, double& x, double& y) {
      const double M0 = obj[3 * i], M1 = obj[3 * i + 1], M2 = obj[3 * i + 2];
      X = (double)(float)(Rt[0][0] * M0 + Rt[0][1] * M1 + Rt[0][2] * M2 + Tt[0]);
      Y = (double)(float)(Rt[1][0] * M0 + Rt[1][1] * M1 + Rt[1][2] * M2 + Tt[1]);
      x = (double)(float)(((double)img[2 * i] - cam.cx) / cam.fx);
      y = (double)(float)(((double)img[2 * i + 1] - cam.cy) / cam.fy);
    };
    double s4[4] = {0, 0, 0, 0};
    for (int i = lane; i < ninl; i += 64)
      { double X, Y, x, y; plane_xy(i, X, Y, x, y); s4[0] += X; s4[1] += Y; s4[2] += x; s4[3] += y; }
    double cM[2], cm[2];
    cM[0] = wsum(s4[0]) / ninl; cM[1] = wsum(s4[1]) / ninl; cm[0] = wsum(s4[2]) / ninl; cm[1] = wsum(s4[3]) / ninl;
    double d4[4] = {0, 0, 0, 0};
    for (int i = lane; i < ninl; i += 64)
      { double X, Y, x, y; plane_xy(i, X, Y, x, y);
        d4[0] += fabs(X - cM[0]); d4[1] += fabs(Y - cM[1]); d4[2] += fabs(x - cm[0]); d4[3] += fabs(y - cm[1]); }
#pragma unroll
    for (int k = 0; k < 4; ++k) d4[k] = wsum(d4[k]);
    if (fabs(d4[0]) < kDblEps || fabs(d4[1]) < kDblEps || fabs(d4[2]) < kDblEps || fabs(d4[3]) < kDblEps) {
      init_ok = false;
    } else {
      const double sM[2] = {ninl / d4[0], ninl / d4[1]}, sm[2] = {ninl / d4[2], ninl / d4[3]};
      // L^T L blocks: S1 = sum QQ^T, Sx = sum x QQ^T, Sy = sum y QQ^T, Sw = sum (x^2+y^2) QQ^T, Q = [X Y 1]
      double q24[24];
#pragma unroll
      for (int k = 0; k < 24; ++k) q24[k] = 0.0;
      for (int i = lane; i < ninl; i += 64)
        {
          double X, Y, x, y; plane_xy(i, X, Y, x, y);
          x = (x - cm[0]) * sm[0]; y = (y - cm[1]) * sm[1]; X = (X - cM[0]) * sM[0]; Y = (Y - cM[1]) * sM[1];
          const double qq[6] = {X * X, X * Y, X, Y * Y, Y, 1.0};
          const double w = x * x + y * y;
#pragma unroll
          for (int k = 0; k < 6; ++k) { q24[k] += qq[k]; q24[6 + k] += x * qq[k]; q24[12 + k] += y * qq[k]; q24[18 + k] += w * qq[k]; }
        }
#pragma unroll
      for (int k = 0; k < 24; ++k) q24[k] = wsum(q24[k]);
      wave_sync();
      for (int idx = lane; idx < 81; idx += 64) {
        const int r = idx / 9, c = idx % 9, br = r / 3, bc = c / 3, ir = r % 3, ic = c % 3;
        const int lo = ir < ic ? ir : ic, hi = ir < ic ? ic : ir;
        const int sidx = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);  // (0,0)0 (0,1)1 (0,2)2 (1,1)3 (1,2)4 (2,2)5
        double v = 0.0;
        if (br == bc && br < 2) v = q24[sidx];
        else if (br == 2 && bc == 2) v = q24[18 + sidx];
        else if ((br == 0 && bc == 2) || (br == 2 && bc == 0)) v = -q24[6 + sidx];
        else if ((br == 1 && bc == 2) || (br == 2 && bc == 1)) v = -q24[12 + sidx];
        sh.A[idx] = v;
      }
      wave_sync();
      smallest_eigvec_reg<9>(sh.A, sh.V, sh.ord, lane);
      double H0[3][3];
      const int c0 = sh.ord[0];
#pragma unroll
      for (int k = 0; k < 9; ++k) H0[k / 3][k % 3] = sh.V[k * 9 + c0];
      // H = invHnorm * H0 * Hnorm2, then / H[2][2]
      const double ih[3][3] = {{1.0 / sm[0], 0, cm[0]}, {0, 1.0 / sm[1], cm[1]}, {0, 0, 1}};
      const double hn[3][3] = {{sM[0], 0, -cM[0] * sM[0]}, {0, sM[1], -cM[1] * sM[1]}, {0, 0, 1}};
      double T1[3][3], H[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) T1[i][j] = ih[i][0] * H0[0][j] + ih[i][1] * H0[1][j] + ih[i][2] * H0[2][j];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) H[i][j] = T1[i][0] * hn[0][j] + T1[i][1] * hn[1][j] + T1[i][2] * hn[2][j];
      const double h22 = 1.0 / H[2][2];
      bool fin = true;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { H[i][j] *= h22; fin = fin && isfinite(H[i][j]); }
      if (fin && ninl > 4) {
        // findHomography's polish: LMSolver (levmarq.cpp) on HomographyRefineCallback, 8 free parameters (h[8] = 1), at most 10
        // iterations -- oracle/pnp_ransac.py:_homography_refine.  One pass over the inliers per iteration yields everything the
        // solver needs at the trial point (S, J^T J, J^T r, max |r|); they are adopted when the trial is accepted.
        //   J rows of a point, with w = 1 / (h6 X + h7 Y + 1), a = (p, q, w) = (X w, Y w, w):  [a 0 -p xi -q xi], [0 a -p yi -q yi]
        //   sums: 0..5 a a^T (pp pq pw qq qw ww) | 6..10 (pp pq qq pw qw) xi | 11..15 the same with yi | 16..18 (pp pq qq)(xi^2 + yi^2)
        //         19..21 a rx | 22..24 a ry | 25..26 (p q)(xi rx + yi ry) | 27 S
        auto eval = [&](const double h[8], double sm_[28], double& rmax) {
          double acc[28];
#pragma unroll
          for (int k = 0; k < 28; ++k) acc[k] = 0.0;
          double mx = 0.0;
          for (int i = lane; i < ninl; i += 64)
            {
              double X, Y, x, y; plane_xy(i, X, Y, x, y);
              double w = h[6] * X + h[7] * Y + 1.0;
              w = fabs(w) > kDblEps ? 1.0 / w : 0.0;
              const double xi = (h[0] * X + h[1] * Y + h[2]) * w, yi = (h[3] * X + h[4] * Y + h[5]) * w;
              const double rx = xi - x, ry = yi - y, pp_ = X * w, qq_ = Y * w;
              const double q5[5] = {pp_ * pp_, pp_ * qq_, qq_ * qq_, pp_ * w, qq_ * w};
              acc[0] += q5[0]; acc[1] += q5[1]; acc[2] += q5[3]; acc[3] += q5[2]; acc[4] += q5[4]; acc[5] += w * w;
              const double e = xi * xi + yi * yi, g = xi * rx + yi * ry;
#pragma unroll
              for (int k = 0; k < 5; ++k) { acc[6 + k] += q5[k] * xi; acc[11 + k] += q5[k] * yi; }
#pragma unroll
              for (int k = 0; k < 3; ++k) acc[16 + k] += q5[k] * e;
              acc[19] += pp_ * rx; acc[20] += qq_ * rx; acc[21] += w * rx;
              acc[22] += pp_ * ry; acc[23] += qq_ * ry; acc[24] += w * ry;
              acc[25] += pp_ * g; acc[26] += qq_ * g;
              acc[27] += rx * rx + ry * ry;
              mx = fmax(mx, fmax(fabs(rx), fabs(ry)));
            }
#pragma unroll
          for (int k = 0; k < 28; ++k) sm_[k] = wsum(acc[k]);
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
          rmax = mx;
        };
        auto normal_eq = [&](const double sm_[28], double N8[8][8], double v8[8]) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) N8[i][j] = 0.0;
          const double aa[3][3] = {{sm_[0], sm_[1], sm_[2]}, {sm_[1], sm_[3], sm_[4]}, {sm_[2], sm_[4], sm_[5]}};
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) { N8[i][j] = aa[i][j]; N8[3 + i][3 + j] = aa[i][j]; }
          // a (p q)^T: rows p -> (pp pq), q -> (pq qq), w -> (pw qw)
          const int ix[3][2] = {{0, 1}, {1, 2}, {3, 4}};
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              N8[i][6 + j] = N8[6 + j][i] = -sm_[6 + ix[i][j]];
              N8[3 + i][6 + j] = N8[6 + j][3 + i] = -sm_[11 + ix[i][j]];
            }
          N8[6][6] = sm_[16]; N8[6][7] = N8[7][6] = sm_[17]; N8[7][7] = sm_[18];
#pragma unroll
          for (int k = 0; k < 6; ++k) v8[k] = sm_[19 + k];
          v8[6] = -sm_[25]; v8[7] = -sm_[26];
        };
        double x8[8] = {H[0][0], H[0][1], H[0][2], H[1][0], H[1][1], H[1][2], H[2][0], H[2][1]};
        double cur[28], rmax;
        eval(x8, cur, rmax);
        double D8[8];
        { double N8[8][8], v8[8]; normal_eq(cur, N8, v8);
#pragma unroll
          for (int k = 0; k < 8; ++k) D8[k] = N8[k][k]; }
        double lam = 1.0, lc = 0.75;
#pragma unroll 1
        for (int it = 0;;) {
          double N8[8][8], v8[8], d[8], xd[8], trial[28], rmd;
          normal_eq(cur, N8, v8);
          {
            double Ap[8][8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
              for (int j = 0; j < 8; ++j) Ap[i][j] = N8[i][j] + (i == j ? lam * D8[i] : 0.0);
            spd_solve<8>(Ap, v8, d);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) xd[k] = x8[k] - d[k];
          eval(xd, trial, rmd);
          const double S = cur[27], Sd = trial[27];
          double dS = 0.0, tt = 0.0, dmax = 0.0;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            double Ad = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) Ad += N8[i][j] * d[j];
            dS += d[i] * (2.0 * v8[i] - Ad);
            tt += d[i] * v8[i];
            dmax = fmax(dmax, fabs(d[i]));
          }
          const double Rr = (S - Sd) / (fabs(dS) > kDblEps ? dS : 1.0);
          if (Rr > 0.75) {
            lam *= 0.5;
            if (lam < lc) lam = 0.0;
          } else if (Rr < 0.25) {
            double nu = (Sd - S) / (fabs(tt) > kDblEps ? tt : 1.0) + 2.0;
            nu = fmin(fmax(nu, 2.0), 10.0);
            if (lam == 0.0) {
              // max |diag(A^-1)|: lane k < 8 solves A c = e_k (the same instruction stream, eight right-hand sides) and keeps c[k].
              // Not a rare path: near convergence the gain ratio is round-off, and the 224 planar test seeds take it ~9 times per call.
              double ek[8], col[8], ckk = 0.0;
#pragma unroll
              for (int i = 0; i < 8; ++i) ek[i] = i == lane ? 1.0 : 0.0;
              spd_solve<8>(N8, ek, col);
#pragma unroll
              for (int i = 0; i < 8; ++i) ckk = i == lane ? fabs(col[i]) : ckk;
#pragma unroll
              for (int o = 4; o > 0; o >>= 1) ckk = fmax(ckk, __shfl_xor(ckk, o));
              const double maxval = fmax(kDblEps, bcast_lane(ckk, 0));
              lam = lc = 1.0 / maxval;
              nu *= 0.5;
            }
            lam *= nu;
          }
          if (Sd < S) {
#pragma unroll
            for (int k = 0; k < 28; ++k) cur[k] = trial[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) x8[k] = xd[k];
            rmax = rmd;
          }
          ++it;
          if (!(it < 10 && dmax >= kDblEps && rmax >= kDblEps)) break;
        }
        H[0][0] = x8[0]; H[0][1] = x8[1]; H[0][2] = x8[2]; H[1][0] = x8[3]; H[1][1] = x8[4]; H[1][2] = x8[5];
        H[2][0] = x8[6]; H[2][1] = x8[7]; H[2][2] = 1.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) fin = fin && isfinite(H[i][j]);
      }
      if (!fin) init_ok = false;
      else {
        const double h1n = sqrt(H[0][0] * H[0][0] + H[1][0] * H[1][0] + H[2][0] * H[2][0]);
        const double h2n = sqrt(H[0][1] * H[0][1] + H[1][1] * H[1][1] + H[2][1] * H[2][1]);
        const double i1 = 1.0 / fmax(h1n, kDblEps), i2 = 1.0 / fmax(h2n, kDblEps), it3 = 2.0 / fmax(h1n + h2n, kDblEps);
        double h1[3] = {H[0][0] * i1, H[1][0] * i1, H[2][0] * i1}, h2[3] = {H[0][1] * i2, H[1][1] * i2, H[2][1] * i2}, h3[3];
        double tt[3] = {H[0][2] * it3, H[1][2] * it3, H[2][2] * it3};
        cross3(h1, h2, h3);
        double Rh[3][3] = {{h1[0], h2[0], h3[0]}, {h1[1], h2[1], h3[1]}, {h1[2], h2[2], h3[2]}};
        double rv[3], dummy[3][9];
        rodrigues_m2v(Rh, rv);
        rodrigues_v2m(rv, Rh, dummy, false);
#pragma unroll
        for (int i = 0; i < 3; ++i) tt[i] += Rh[i][0] * Tt[0] + Rh[i][1] * Tt[1] + Rh[i][2] * Tt[2];
        double Rf[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) Rf[i][j] = Rh[i][0] * Rt[0][j] + Rh[i][1] * Rt[1][j] + Rh[i][2] * Rt[2][j];
        rodrigues_m2v(Rf, p);
        p[3] = tt[0]; p[4] = tt[1]; p[5] = tt[2];
      }
    }
    if (!init_ok) { p[0] = p[1] = p[2] = p[3] = p[4] = p[5] = 0.0; init_ok = true; }
  return init_ok;
}

// non-planar initial guess of cvFindExtrinsicCameraParams2 (12 x 12 DLT)
__device__ __noinline__ bool pnp_init_dlt(Shared& sh, const float* obj, const float* img, int ninl, int lane, const Cam& cam, const double mc[3], const double Vc[3][3], double p[6]) {
    // DLT: L^T L from 4 weighted sums of P P^T, P = [X Y Z 1]
    double s40[40];
#pragma unroll
    for (int k = 0; k < 40; ++k) s40[k] = 0.0;
    for (int i = lane; i < ninl; i += 64)
      {
        const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        const double x = -(((double)img[2 * i] - cam.cx) / cam.fx), y = -(((double)img[2 * i + 1] - cam.cy) / cam.fy);
        const double pp[10] = {X * X, X * Y, X * Z, X, Y * Y, Y * Z, Y, Z * Z, Z, 1.0};
        const double w = x * x + y * y;
#pragma unroll
        for (int k = 0; k < 10; ++k) { s40[k] += pp[k]; s40[10 + k] += x * pp[k]; s40[20 + k] += y * pp[k]; s40[30 + k] += w * pp[k]; }
      }
#pragma unroll
    for (int k = 0; k < 40; ++k) s40[k] = wsum(s40[k]);
    wave_sync();
    for (int idx = lane; idx < 144; idx += 64) {
      const int r = idx / 12, c = idx % 12, br = r / 4, bc = c / 4, ir = r % 4, ic = c % 4;
      const int lo = ir < ic ? ir : ic, hi = ir < ic ? ic : ir;
      const int sidx = lo == 0 ? hi : (lo == 1 ? 3 + hi : (lo == 2 ? 5 + hi : 9));  // upper-tri index of 4x4
      double v = 0.0;
      if (br == bc && br < 2) v = s40[sidx];
      else if (br == 2 && bc == 2) v = s40[30 + sidx];
      else if ((br == 0 && bc == 2) || (br == 2 && bc == 0)) v = s40[10 + sidx];
      else if ((br == 1 && bc == 2) || (br == 2 && bc == 1)) v = s40[20 + sidx];
      sh.A[idx] = v;
    }
    wave_sync();
    smallest_eigvec_reg<12>(sh.A, sh.V, sh.ord, lane);
    const int c0 = sh.ord[0];
    double RR[3][3], tt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) RR[i][j] = sh.V[(4 * i + j) * 12 + c0];
      tt[i] = sh.V[(4 * i + 3) * 12 + c0];
    }
    if (det3(RR) < 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { tt[i] = -tt[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) RR[i][j] = -RR[i][j]; }
    }
    double sc = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) sc += RR[i][j] * RR[i][j];
    sc = sqrt(sc);
    double Rq[3][3];
    polar3(RR, Rq);
    double rn = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) rn += Rq[i][j] * Rq[i][j];
    rn = sqrt(rn);
    rodrigues_m2v(Rq, p);
    p[3] = tt[0] * (rn / sc); p[4] = tt[1] * (rn / sc); p[5] = tt[2] * (rn / sc);
  return true;
}

// Two launches per batch.  k_pnp_hyp: one single-wave workgroup per (pair, RANSAC hypothesis) -- the
// RNG stream and therefore the subsets do not depend on the models, so workgroup h replays cv::RNG to its
// own subset, runs EPnP and scores it with the full 512-register budget of a lone wave.  k_pnp_refine:
// one wave per pair replays the sequential `good > max(maxGood, 4)` / RANSACUpdateNumIters logic over
// the results in hypothesis order (hypotheses past the adapted iteration count are ignored, exactly as
// the sequential loop would never have computed them) and refines the winner.
__global__ __launch_bounds__(64) void k_pnp_hyp(PnpArgs a) {
  __shared__ Shared sh;
  const int b = blockIdx.y, wave = blockIdx.x, lane = threadIdx.x;
  const int n = a.n_pts[b];
  const float* obj = a.obj + (size_t)b * a.kstride * 3;
  const float* img = a.img + (size_t)b * a.kstride * 2;
  uint8_t* mask_w = a.mask_ws + ((size_t)b * kMaxHyp + wave) * a.kstride;
  HypResult* hyp = a.hyp + (size_t)b * kMaxHyp;
  const Cam cam = {a.fx, a.fy, a.cx, a.cy};
  if (n < a.min_pts || n < 5) return;
  {
    unsigned long long rng = 0xFFFFFFFFFFFFFFFFull;
    int idx[5] = {0, 1, 2, 3, 4};
    if (n > 5) {
      for (int h = 0; h <= wave; ++h) {   // replay the stream up to this wave's subset
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          for (;;) {
            const int v = (int)(rng_next(rng) % (unsigned)n);
            bool dup = false;
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) if (jj < i && idx[jj] == v) dup = true;
            if (!dup) { idx[i] = v; break; }
          }
        }
      }
    }
    if (lane < 5) {
      int id = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i) if (i == lane) id = idx[i];
      sh.pws[lane * 3 + 0] = obj[3 * id]; sh.pws[lane * 3 + 1] = obj[3 * id + 1]; sh.pws[lane * 3 + 2] = obj[3 * id + 2];
      // cv::undistortPoints output takes the input's depth (float32): computed in double, stored as float (oracle: solve_pnp_ransac)
      // ... and epnp::init_points re-applies the intrinsics to them: us = x fu + uc, in double
      sh.us[2 * lane] = (double)(float)(((double)img[2 * id] - cam.cx) / cam.fx) * cam.fx + cam.cx;
      sh.us[2 * lane + 1] = (double)(float)(((double)img[2 * id + 1] - cam.cy) / cam.fy) * cam.fy + cam.cy;
    }
    if (lane == 0) { sh.cam[0] = cam.fx; sh.cam[1] = cam.fy; sh.cam[2] = cam.cx; sh.cam[3] = cam.cy; }
    wave_sync();
    double R[3][3], t[3];
    long long ts[9];
    if (a.dbg_ts) ts[0] = (long long)__builtin_amdgcn_s_memtime();
    const bool okm = epnp5(sh, lane, R, t, nullptr, a.dbg_ts ? ts : nullptr);
    int good = 0;
    if (okm) {
      const float thr = a.reproj * a.reproj;
      for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        bool inl = false;
        if (i < n) {
          const double M0 = obj[3 * i], M1 = obj[3 * i + 1], M2 = obj[3 * i + 2];
          const double X = R[0][0] * M0 + R[0][1] * M1 + R[0][2] * M2 + t[0];
          const double Y = R[1][0] * M0 + R[1][1] * M1 + R[1][2] * M2 + t[1];
          const double Z = R[2][0] * M0 + R[2][1] * M1 + R[2][2] * M2 + t[2];
          const double z = Z != 0 ? 1.0 / Z : 1.0;
          const float pu = (float)(X * z * cam.fx + cam.cx), pv = (float)(Y * z * cam.fy + cam.cy);
          const float dx = img[2 * i] - pu, dy = img[2 * i + 1] - pv;
          const float e = dx * dx + dy * dy;   // float32 squared reprojection error, tested <= thr
          inl = e <= thr;
          mask_w[i] = inl ? 1 : 0;
        }
        good += __popcll(__ballot(inl));
      }
    }
    if (a.dbg_ts && lane == 0) {   // developer: s_memtime phase stamps of this hypothesis
      ts[7] = (long long)__builtin_amdgcn_s_memtime();
      for (int k = 0; k < 9; ++k) a.dbg_ts[((size_t)b * kMaxHyp + wave) * 16 + k] = ts[k];
    }
    if (lane == 0) {
      HypResult& hr = hyp[wave];
      hr.good = good; hr.valid = okm ? 1 : 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) { hr.t[i] = t[i];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) hr.R[3 * i + jj] = R[i][jj]; }
    }
  }
}

// ---- P3P: cv::solvePnPRansac with npoints == 4 makes one solvePnP(SOLVEPNP_P3P) call (core/_shared.py:109-116 -> calib3d solvepnp.cpp): Gao's
// P3P on the first three points (OpenCV p3p.cpp solve_for_lengths), the fourth point picks the pose.  Scalar f64 code run by ONE lane -- the
// branch is unreachable from PoseNode (MIN_MATCHES 15) and TwistNode (30) and exists for the compute_pose seam.  Restated in oracle/pnp_ransac.py
// (solve_p3p): same polynomial, same acceptance tests, same candidate order (ascending root), triad alignment on both sides.
__device__ int quartic_real_roots(double A, double B, double C, double D, double E, double out[4]) {
  // all four roots by Aberth-Ehrlich iteration in complex arithmetic (monic form), then the real ones, ascending
  const double b = B / A, c = C / A, d = D / A, e = E / A;
  const double rad = 1.0 + fmax(fmax(fabs(b), fabs(c)), fmax(fabs(d), fabs(e)));
  double zr[4], zi[4];
  for (int k = 0; k < 4; ++k) { const double ang = 0.7 + 1.5707963267948966 * k; zr[k] = 0.5 * rad * cos(ang); zi[k] = 0.5 * rad * sin(ang); }
  for (int it = 0; it < 200; ++it) {
    double moved = 0.0;
    for (int k = 0; k < 4; ++k) {
      // p(z), p'(z) by Horner
      double pr = 1.0, pi = 0.0, dr = 0.0, di = 0.0;
      const double co[4] = {b, c, d, e};
      for (int j = 0; j < 4; ++j) {
        const double ndr = dr * zr[k] - di * zi[k] + pr, ndi = dr * zi[k] + di * zr[k] + pi;
        dr = ndr; di = ndi;
        const double npr = pr * zr[k] - pi * zi[k] + co[j], npi = pr * zi[k] + pi * zr[k];
        pr = npr; pi = npi;
      }
      double dn = dr * dr + di * di;
      if (dn == 0.0) { dr = 1e-150; dn = 1e-300; }     // (a vanishing derivative: step away instead of dividing by zero; 1e-600 would itself round to 0)
      double wr = (pr * dr + pi * di) / dn, wi = (pi * dr - pr * di) / dn;       // w = p / p'
      double sr = 0.0, si = 0.0;                                               // sum 1 / (z_k - z_j)
      for (int j = 0; j < 4; ++j) {
        if (j == k) continue;
        const double xr = zr[k] - zr[j], xi = zi[k] - zi[j], xn = xr * xr + xi * xi;
        if (xn > 0.0) { sr += xr / xn; si -= xi / xn; }
      }
      const double qr = 1.0 - (wr * sr - wi * si), qi = -(wr * si + wi * sr), qn = qr * qr + qi * qi;
      if (qn > 0.0) { const double ur = (wr * qr + wi * qi) / qn, ui = (wi * qr - wr * qi) / qn; wr = ur; wi = ui; }
      zr[k] -= wr; zi[k] -= wi;
      moved = fmax(moved, fabs(wr) + fabs(wi));
    }
    if (moved < 1e-15 * rad) break;
  }
  int n = 0;
  for (int k = 0; k < 4; ++k)
    if (fabs(zi[k]) <= 1e-9 * fmax(1.0, fabs(zr[k]))) {
      double x = zr[k];
      for (int it = 0; it < 4; ++it) {          // Newton polish on the real axis (the oracle does the same after numpy's eigenvalue solver)
        const double pv = (((x + b) * x + c) * x + d) * x + e, dv = ((4.0 * x + 3.0 * b) * x + 2.0 * c) * x + d;
        if (dv == 0.0) break;
        x -= pv / dv;
      }
      out[n++] = x;
    }
  for (int i = 1; i < n; ++i) for (int j = i; j > 0 && out[j - 1] > out[j]; --j) { const double tmp = out[j]; out[j] = out[j - 1]; out[j - 1] = tmp; }
  return n;
}

__device__ inline void triad3(const double p0[3], const double p1[3], const double p2[3], double F[3][3]) {
  double e1[3], u[3], e3[3], e2[3];
  for (int i = 0; i < 3; ++i) { e1[i] = p1[i] - p0[i]; u[i] = p2[i] - p0[i]; }
  double n1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
  for (int i = 0; i < 3; ++i) e1[i] /= n1;
  cross3(e1, u, e3);
  n1 = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
  for (int i = 0; i < 3; ++i) e3[i] /= n1;
  cross3(e3, e1, e2);
  for (int i = 0; i < 3; ++i) { F[i][0] = e1[i]; F[i][1] = e2[i]; F[i][2] = e3[i]; }
}

__device__ __noinline__ bool p3p_gao(const float* obj, const float* img, const Cam& cam, double Rb[3][3], double tb[3]) {
  double X[4][3], u[4][2], f[3][3];
  for (int i = 0; i < 4; ++i) {
    for (int k = 0; k < 3; ++k) X[i][k] = (double)obj[3 * i + k];
    // cv::undistortPoints keeps the input's depth: normalised coordinates computed in double, stored as float32
    u[i][0] = (double)(float)(((double)img[2 * i] - cam.cx) / cam.fx);
    u[i][1] = (double)(float)(((double)img[2 * i + 1] - cam.cy) / cam.fy);
  }
  for (int i = 0; i < 3; ++i) {
    const double nn = sqrt(u[i][0] * u[i][0] + u[i][1] * u[i][1] + 1.0);
    f[i][0] = u[i][0] / nn; f[i][1] = u[i][1] / nn; f[i][2] = 1.0 / nn;
  }
  auto dist = [&](int i, int j) { const double a0 = X[i][0] - X[j][0], a1 = X[i][1] - X[j][1], a2 = X[i][2] - X[j][2]; return sqrt(a0 * a0 + a1 * a1 + a2 * a2); };
  auto dotf = [&](int i, int j) { return f[i][0] * f[j][0] + f[i][1] * f[j][1] + f[i][2] * f[j][2]; };
  const double d0 = dist(1, 2), d1 = dist(0, 2), d2 = dist(0, 1);
  if (d0 == 0.0 || d1 == 0.0 || d2 == 0.0) return false;
  const double p = 2 * dotf(1, 2), q = 2 * dotf(0, 2), r = 2 * dotf(0, 1);
  const double inv_d22 = 1.0 / (d2 * d2);
  const double a = inv_d22 * d0 * d0, b = inv_d22 * d1 * d1;
  const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r, pr = p * r, pqr = q * p * r;
  if (p2 + q2 + r2 - pqr - 1 == 0) return false;
  const double ab = a * b, a_2 = 2 * a, a_4 = 4 * a;
  const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
  if (A == 0) return false;
  const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
  const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
  const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
  const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
  const double temp = p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr;
  const double b0 = b * temp * temp;
  if (b0 == 0) return false;
  double roots[4];
  const int nr = quartic_real_roots(A, B, C, D, E, roots);
  const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
  double Fo[3][3];
  triad3(X[0], X[1], X[2], Fo);
  double best = INFINITY;
  bool found = false;
  for (int k = 0; k < nr; ++k) {
    const double x = roots[k];
    if (x <= 0) continue;
    const double x2 = x * x;
    const double b1 = ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
        (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
          (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
         (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
          pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
         2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
         p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
    if (b1 <= 0) continue;
    const double y = b1 / b0, v = x2 + y * y - x * y * r;
    if (v <= 0) continue;
    const double Z = d2 / sqrt(v);
    const double len[3] = {x * Z, y * Z, Z};
    double M[3][3], Fc[3][3], R[3][3], t[3];
    for (int i = 0; i < 3; ++i) for (int c2 = 0; c2 < 3; ++c2) M[i][c2] = f[i][c2] * len[i];
    triad3(M[0], M[1], M[2], Fc);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = Fc[i][0] * Fo[j][0] + Fc[i][1] * Fo[j][1] + Fc[i][2] * Fo[j][2];
    for (int i = 0; i < 3; ++i) t[i] = M[0][i] - (R[i][0] * X[0][0] + R[i][1] * X[0][1] + R[i][2] * X[0][2]);
    double pc[3];
    for (int i = 0; i < 3; ++i) pc[i] = R[i][0] * X[3][0] + R[i][1] * X[3][1] + R[i][2] * X[3][2] + t[i];
    double err = INFINITY;
    if (pc[2] != 0) { const double ex = pc[0] / pc[2] - u[3][0], ey = pc[1] / pc[2] - u[3][1]; err = ex * ex + ey * ey; }
    if (!found || err < best) {
      best = err; found = true;
      for (int i = 0; i < 3; ++i) { tb[i] = t[i]; for (int j = 0; j < 3; ++j) Rb[i][j] = R[i][j]; }
    }
  }
  return found;
}

constexpr int kLdsPts = 2048;   // inlier correspondences k_pnp_refine keeps in LDS (20 bytes each)
__global__ __launch_bounds__(64) void k_pnp_refine(PnpArgs a) {
  __shared__ Shared sh;
  __shared__ float cpts[5 * kLdsPts];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nhyp = a.iterations;
  const int n = a.n_pts[b];
  const HypResult* hyp = a.hyp + (size_t)b * kMaxHyp;
  double* Rout = a.R + (size_t)b * 9;
  double* tout = a.t + (size_t)b * 3;
  const Cam cam = {a.fx, a.fy, a.cx, a.cy};
  long long ts[8];
  auto stamp = [&](int k) __attribute__((always_inline)) { if (a.dbg_ts) ts[k] = (long long)__builtin_amdgcn_s_memtime(); };
  stamp(0);
  if (n == 4 && a.min_pts <= 4) {   // solvePnPRansac's npoints == 4 branch: one P3P solve, every point an inlier, no refinement
    if (lane == 0) {
      double Rb[3][3], tb[3], rv[3], Rf[3][3], dummy[3][9];
      bool ok4 = p3p_gao(a.obj + (size_t)b * a.kstride * 3, a.img + (size_t)b * a.kstride * 2, cam, Rb, tb);
      if (ok4) {
        rodrigues_m2v(Rb, rv);                       // the reference applies cv2.Rodrigues to the returned rvec
        rodrigues_v2m(rv, Rf, dummy, false);
        for (int i = 0; i < 3; ++i) { ok4 = ok4 && isfinite(tb[i]); for (int j = 0; j < 3; ++j) ok4 = ok4 && isfinite(Rf[i][j]); }
      }
      for (int i = 0; i < 3; ++i) { tout[i] = ok4 ? tb[i] : 0.0; for (int j = 0; j < 3; ++j) Rout[3 * i + j] = ok4 ? Rf[i][j] : (i == j ? 1.0 : 0.0); }
      a.ok[b] = ok4 ? 1 : 0; a.n_inliers[b] = ok4 ? 4 : 0;
    }
    return;
  }
  if (n == 5 && a.min_pts <= 5) {   // the same `model_points == npoints` block: one solvePnP(SOLVEPNP_EPNP) on all five points (hypothesis 0 IS that solve), every point an inlier, no refinement
    if (lane == 0) {
      bool ok5 = hyp[0].valid != 0;
      double Rb[3][3], rv[3], Rf[3][3], dummy[3][9];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rb[i][j] = hyp[0].R[3 * i + j];
      if (ok5) {
        rodrigues_m2v(Rb, rv);
        rodrigues_v2m(rv, Rf, dummy, false);
        for (int i = 0; i < 3; ++i) { ok5 = ok5 && isfinite(hyp[0].t[i]); for (int j = 0; j < 3; ++j) ok5 = ok5 && isfinite(Rf[i][j]); }
      }
      for (int i = 0; i < 3; ++i) { tout[i] = ok5 ? hyp[0].t[i] : 0.0; for (int j = 0; j < 3; ++j) Rout[3 * i + j] = ok5 ? Rf[i][j] : (i == j ? 1.0 : 0.0); }
      a.ok[b] = ok5 ? 1 : 0; a.n_inliers[b] = ok5 ? 5 : 0;
    }
    return;
  }
  if (n < a.min_pts || n < 5) {
    if (lane < 9) Rout[lane] = (lane % 4 == 0) ? 1.0 : 0.0;
    if (lane < 3) tout[lane] = 0.0;
    if (lane == 0) { a.ok[b] = 0; a.n_inliers[b] = 0; }
    return;
  }

  // ---- sequential selection over the hypotheses ----------------------------------------------------
  int niters = nhyp, max_good = 0, best = -1;
  for (int it = 0; it < nhyp; ++it) {
    if (it >= niters) break;
    if (!hyp[it].valid) continue;
    const int good = hyp[it].good;
    if (good > (max_good > 4 ? max_good : 4)) {
      best = it; max_good = good;
      niters = ransac_update_iters(a.confidence, (double)(n - good) / n, 5, niters);
    }
  }
  if (best < 0) {
    if (lane < 9) Rout[lane] = (lane % 4 == 0) ? 1.0 : 0.0;
    if (lane < 3) tout[lane] = 0.0;
    if (lane == 0) { a.ok[b] = 0; a.n_inliers[b] = 0; }
    return;
  }
  stamp(1);
  double bestR[3][3], bestT[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { bestT[i] = hyp[best].t[i];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) bestR[i][jj] = hyp[best].R[3 * i + jj]; }
  const uint8_t* mask_best = a.mask_ws + ((size_t)b * kMaxHyp + best) * a.kstride;

  // ---- solvePnP(SOLVEPNP_ITERATIVE) on the inliers ---------------------------------------------
  // The inliers are compacted once (order kept) into LDS -- object points [k][3], image points [k][2] -- and every pass below (two for
  // the PCA, 3 + 11 for a planar start or 1 for the DLT, up to ~40 of the pose LM) strides over ninl entries: no mask test, no
  // global-memory latency per point, ninl / 64 instead of n / 64 trips.  More than kLdsPts inliers: the same layout in a.pts_ws.
  const uint8_t* mask = mask_best;
  const int ninl = max_good;
  const bool in_lds = ninl <= kLdsPts;
  float* const cobj = in_lds ? cpts : a.pts_ws + (size_t)b * a.kstride * 5;
  float* const cimg = cobj + 3 * (size_t)(in_lds ? kLdsPts : a.kstride);
  {
    // four chunks of 64 candidates per trip: their 24 loads are in flight together (the loop is otherwise one memory latency per chunk)
    const float* const gobj = a.obj + (size_t)b * a.kstride * 3;
    const float* const gimg = a.img + (size_t)b * a.kstride * 2;
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
      bool m[4]; float o[4][3], u[4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = i0 + 64 * c + lane, ic = i < n ? i : n - 1;
        m[c] = i < n && mask[ic] != 0;
        o[c][0] = gobj[3 * ic]; o[c][1] = gobj[3 * ic + 1]; o[c][2] = gobj[3 * ic + 2];
        u[c][0] = gimg[2 * ic]; u[c][1] = gimg[2 * ic + 1];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned long long bal = __ballot(m[c]);
        const int k = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (m[c] && k < ninl) {     // k < ninl always (ninl is the popcount k_pnp_hyp took of this mask); the bound keeps a corrupted count inside the buffer
          cobj[3 * k] = o[c][0]; cobj[3 * k + 1] = o[c][1]; cobj[3 * k + 2] = o[c][2];
          cimg[2 * k] = u[c][0]; cimg[2 * k + 1] = u[c][1];
        }
        base += __popcll(bal);
      }
    }
    wave_sync();
  }
  const float* const obj = cobj;
  const float* const img = cimg;
  double p[6];
  // PCA of the inlier object points
  double mc[3] = {0, 0, 0};
  for (int i = lane; i < ninl; i += 64) { mc[0] += obj[3 * i]; mc[1] += obj[3 * i + 1]; mc[2] += obj[3 * i + 2]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) mc[k] = wsum(mc[k]) / ninl;
  double mm[6] = {0, 0, 0, 0, 0, 0};
  for (int i = lane; i < ninl; i += 64) {
      const double d0 = obj[3 * i] - mc[0], d1 = obj[3 * i + 1] - mc[1], d2 = obj[3 * i + 2] - mc[2];
      mm[0] += d0 * d0; mm[1] += d0 * d1; mm[2] += d0 * d2; mm[3] += d1 * d1; mm[4] += d1 * d2; mm[5] += d2 * d2;
    }
#pragma unroll
  for (int k = 0; k < 6; ++k) mm[k] = wsum(mm[k]);
  double MM[3][3] = {{mm[0], mm[1], mm[2]}, {mm[1], mm[3], mm[4]}, {mm[2], mm[4], mm[5]}};
  double W[3], Vc[3][3];
  eig3(MM, W, Vc);
  stamp(2);
  bool init_ok = true;
  if (W[2] / W[1] < 1e-3) {
    init_ok = pnp_init_planar(sh, obj, img, ninl, lane, cam, mc, Vc, p);
  } else if (ninl >= 6) {
    init_ok = pnp_init_dlt(sh, obj, img, ninl, lane, cam, mc, Vc, p);
  } else {
    init_ok = false;  // < 6 non-planar inliers: OpenCV >= 4.5 falls back to the RANSAC model
  }

  stamp(3);
  double Rf[3][3], dummy[3][9];
  if (init_ok) {
    levmarq_pose(obj, img, ninl, lane, cam, p);
    stamp(4);
    rodrigues_v2m(p, Rf, dummy, false);
  } else {
    // rvec = Rodrigues(bestR) then back, as the reference applies cv2.Rodrigues to the returned rvec
    double rv[3];
    rodrigues_m2v(bestR, rv);
    rodrigues_v2m(rv, Rf, dummy, false);
    p[3] = bestT[0]; p[4] = bestT[1]; p[5] = bestT[2];
  }
  if (a.dbg_ts && lane == 0) {
    stamp(5);
    for (int k = 0; k < 6; ++k) a.dbg_ts[((size_t)a.B * kMaxHyp + b) * 16 + k] = ts[k];
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { tout[i] = p[3 + i];
#pragma unroll
      for (int j = 0; j < 3; ++j) Rout[3 * i + j] = Rf[i][j]; }
    bool fin = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) fin = fin && isfinite(p[k]);
    a.ok[b] = fin ? 1 : 0;
    a.n_inliers[b] = ninl;
  }
}
}  // namespace

// test hook: EPnP on n independent 5-point sets (world points f64 [n][5][3], normalised image points
// f64 [n][5][2]); out [n][64]: R (9), t (3), candidate errors (3), candidate betas (12), eigenvalues (12), rho(6), L row 0 (10)
__global__ __launch_bounds__(64) void k_epnp_debug(const double* pws, const double* us, double* out, const double* cam4) {
  __shared__ Shared sh;
  const int b = blockIdx.x, lane = threadIdx.x;
  if (lane < 15) sh.pws[lane] = pws[b * 15 + lane];
  if (lane < 10) sh.us[lane] = us[b * 10 + lane];
  if (lane < 4) sh.cam[lane] = cam4 ? cam4[lane] : (lane < 2 ? 1.0 : 0.0);      // default: identity camera (us are normalised coordinates)
  __syncthreads();
  double R[3][3], t[3];
  double* o = out + (size_t)b * 64;
  const bool ok = epnp5(sh, lane, R, t, o);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[9 + i] = t[i];
#pragma unroll
      for (int j = 0; j < 3; ++j) o[3 * i + j] = R[i][j]; }
    for (int k = 0; k < 12; ++k) o[27 + k] = sh.A[sh.ord[k] * 12 + sh.ord[k]];
    for (int k = 0; k < 6; ++k) o[39 + k] = sh.rho[k];
    for (int k = 0; k < 10; ++k) o[45 + k] = sh.L[k];
    o[55] = ok ? 1.0 : 0.0;
  }
}

void launch_epnp_debug(const double* pws, const double* us, double* out, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_epnp_debug, dim3(n), dim3(64), 0, s, pws, us, out, (const double*)nullptr);
}

void launch_pnp(const PnpArgs& a, hipStream_t s) {
  const int nh = a.iterations < 1 ? 1 : (a.iterations > kMaxHyp ? kMaxHyp : a.iterations);
  hipLaunchKernelGGL(k_pnp_hyp, dim3(nh, a.B), dim3(64), 0, s, a);
  hipLaunchKernelGGL(k_pnp_refine, dim3(a.B), dim3(64), 0, s, a);
}

}  // namespace gn
