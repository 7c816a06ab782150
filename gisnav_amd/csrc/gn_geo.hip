// Post-pose georeferencing of GISNav's PoseNode, host-side scalar code:
// camera position in the reference raster -> WGS 84 -> ECEF, orientation -> ENU -> ECEF quaternion.
//   ros/gisnav/gisnav/core/pose_node.py:333-381, ros/gisnav/gisnav/_transformations.py:298-393
// (pyproj `latlong -> geocent` on the WGS 84 datum and transforms3d `mat2quat` are restated in closed form).
#include "gn_common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {
constexpr double kA = 6378137.0;                    // WGS 84 semi-major axis
constexpr double kF = 1.0 / 298.257223563;          // WGS 84 flattening
constexpr double kPi = 3.14159265358979323846;

// symmetric 4x4 eigen-decomposition by cyclic Jacobi; returns the eigenvector of the largest eigenvalue
void top_eigenvector4(double K[4][4], double q[4]) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0;
    for (int p = 0; p < 4; ++p) for (int r = p + 1; r < 4; ++r) off += K[p][r] * K[p][r];
    if (off < 1e-300) break;
    for (int p = 0; p < 4; ++p)
      for (int r = p + 1; r < 4; ++r) {
        if (K[p][r] == 0.0) continue;
        const double theta = (K[r][r] - K[p][p]) / (2.0 * K[p][r]);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < 4; ++k) { const double kp = K[k][p], kr = K[k][r]; K[k][p] = c * kp - s * kr; K[k][r] = s * kp + c * kr; }
        for (int k = 0; k < 4; ++k) { const double pk = K[p][k], rk = K[r][k]; K[p][k] = c * pk - s * rk; K[r][k] = s * pk + c * rk; }
        for (int k = 0; k < 4; ++k) { const double vp = V[k][p], vr = V[k][r]; V[k][p] = c * vp - s * vr; V[k][r] = s * vp + c * vr; }
      }
  }
  int best = 0;
  for (int i = 1; i < 4; ++i) if (K[i][i] > K[best][best]) best = i;
  for (int k = 0; k < 4; ++k) q[k] = V[k][best];
}
}  // namespace

extern "C" {

// _transformations.py:298-323 proj_to_affine: "+proj=affine +xoff=.. +yoff=.. +zoff=.. +s11=.. ... +s33=.." -> 3x4 row-major
int gn_proj_to_affine(const char* proj_str, double* affine12) {
  if (!proj_str || !affine12) return GN_ERR_ARG;
  static const char* keys[12] = {"+s11", "+s12", "+s13", "+xoff", "+s21", "+s22", "+s23", "+yoff", "+s31", "+s32", "+s33", "+zoff"};
  const std::string s(proj_str);
  for (int i = 0; i < 12; ++i) {
    const std::string key = std::string(keys[i]) + "=";
    size_t pos = 0; bool found = false;
    while ((pos = s.find(key, pos)) != std::string::npos) {
      if (pos == 0 || s[pos - 1] == ' ') { found = true; break; }      // whole token, as str.split() sees it
      pos += key.size();
    }
    if (!found) return GN_ERR_NAME;
    char* end = nullptr;
    const char* start = s.c_str() + pos + key.size();
    affine12[i] = std::strtod(start, &end);
    if (end == start) return GN_ERR_NAME;
  }
  return GN_OK;
}

// _transformations.py:326-345 wgs84_to_ecef (pyproj latlong -> geocent, WGS 84): closed form
int gn_wgs84_to_ecef(double lon_deg, double lat_deg, double alt, double* xyz3) {
  if (!xyz3) return GN_ERR_ARG;
  const double lon = lon_deg * (kPi / 180.0), lat = lat_deg * (kPi / 180.0);
  const double e2 = kF * (2.0 - kF);
  const double sl = std::sin(lat), cl = std::cos(lat);
  const double N = kA / std::sqrt(1.0 - e2 * sl * sl);
  xyz3[0] = (N + alt) * cl * std::cos(lon);
  xyz3[1] = (N + alt) * cl * std::sin(lon);
  xyz3[2] = (N * (1.0 - e2) + alt) * sl;
  return GN_OK;
}

// pose_node.py:333-381: (r, t) of compute_pose + the raster's affine CRS -> earth-frame position and orientation.
// Returns GN_OK, or 1 when the camera centre falls outside the expected range of the reference raster (the node logs a
// warning and returns None, pose_node.py:339-341).
int gn_pose_to_earth(const double* R9, const double* t3, const double* affine12, int ref_h, int ref_w,
                     double* position_ecef3, double* quat_xyzw4, double* lonlatalt3) {
  if (!R9 || !t3 || !affine12 || !position_ecef3 || !quat_xyzw4) return GN_ERR_ARG;
  double ri[3][3];                                    // r_inv = r.T
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ri[i][j] = R9[j * 3 + i];
  double pos[3];                                      // camera_optical_position_in_world = -r_inv @ t
  for (int i = 0; i < 3; ++i) pos[i] = -(ri[i][0] * t3[0] + ri[i][1] * t3[1] + ri[i][2] * t3[2]);
  const long long xi = (long long)pos[0], yi = (long long)pos[1];     // int(x): truncation toward zero
  if (!(0 <= xi && xi <= ref_h && 0 <= yi && yi <= ref_w)) return 1;  // the reference compares x with shape[0], y with shape[1]
  double w84[3];                                      // t_wgs84 = affine @ [pos; 1]
  for (int i = 0; i < 3; ++i) w84[i] = affine12[4 * i] * pos[0] + affine12[4 * i + 1] * pos[1] + affine12[4 * i + 2] * pos[2] + affine12[4 * i + 3];
  if (lonlatalt3) { lonlatalt3[0] = w84[0]; lonlatalt3[1] = w84[1]; lonlatalt3[2] = w84[2]; }
  gn_wgs84_to_ecef(w84[0], w84[1], w84[2], position_ecef3);
  double Rn[3][3];                                    // R = affine[:3, :3] / column norms
  for (int j = 0; j < 3; ++j) {
    const double n = std::sqrt(affine12[j] * affine12[j] + affine12[4 + j] * affine12[4 + j] + affine12[8 + j] * affine12[8 + j]);
    for (int i = 0; i < 3; ++i) Rn[i][j] = affine12[4 * i + j] / n;
  }
  double enu[3][3];                                   // camera_optical_rotation_in_enu = R @ r_inv
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) enu[i][j] = Rn[i][0] * ri[0][j] + Rn[i][1] * ri[1][j] + Rn[i][2] * ri[2][j];
  const double lon = w84[0] * (kPi / 180.0), lat = w84[1] * (kPi / 180.0);     // enu_to_ecef_matrix(lon, lat)
  const double slat = std::sin(lat), clat = std::cos(lat), slon = std::sin(lon), clon = std::cos(lon);
  const double E[3][3] = {{-slon, -slat * clon, clat * clon}, {clon, -slat * slon, clat * slon}, {0, clat, slat}};
  double M[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i][j] = E[i][0] * enu[0][j] + E[i][1] * enu[1][j] + E[i][2] * enu[2][j];
  // transforms3d.quaternions.mat2quat: eigenvector of the largest eigenvalue of K, w >= 0; tf_transformations order x, y, z, w
  const double Qxx = M[0][0], Qyx = M[0][1], Qzx = M[0][2], Qxy = M[1][0], Qyy = M[1][1], Qzy = M[1][2], Qxz = M[2][0], Qyz = M[2][1], Qzz = M[2][2];
  double K[4][4] = {{Qxx - Qyy - Qzz, Qyx + Qxy, Qzx + Qxz, Qyz - Qzy},
                    {Qyx + Qxy, Qyy - Qxx - Qzz, Qzy + Qyz, Qzx - Qxz},
                    {Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, Qxy - Qyx},
                    {Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz}};
  for (auto& row : K) for (double& v : row) v /= 3.0;
  double q[4];
  top_eigenvector4(K, q);                              // (x, y, z, w)
  const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double sgn = q[3] < 0 ? -1.0 : 1.0;
  for (int k = 0; k < 4; ++k) quat_xyzw4[k] = sgn * q[k] / nq;
  return GN_OK;
}

}  // extern "C"
