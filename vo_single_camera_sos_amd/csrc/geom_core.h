// Device-side restatement of the reference's numpy geometry on the hot path.
//   omnistereo/panorama.py:635-641, :616-622   pano pixel -> (azimuth, elevation), NaN outside
//   omnistereo/camera_models.py:1049-1065      angles -> unit bearing
//   omnistereo/camera_models.py:3333-3340, :2437-2490  rays on the unit cylinder, midpoint triangulation
//   omnistereo/camera_models.py:3309-3319      range filter on the homogeneous row (norm includes the 1)
//   omnistereo/common_cv.py:177-186            pixel gates
// FP64 throughout; sin / cos / atan are trig_core.h's (the text of oracle/trig_core.h: + - * / only), not the device math
// library's, so that the angles, bearings and triangulated points equal the CPU oracle's bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "sosvo.h"
#include "trig_core.h"

#define SV_TWO_PI (2 * 3.14159265358979323846 * 1.0) /* cyl_circumference, panorama.py:152 */

__device__ __forceinline__ static double sv_nan() { return __longlong_as_double(0x7FF8000000000000LL); }

// pano = {cols, rows, pixel_size, cyl_height_max}
__device__ __forceinline__ static void sv_pano_angles(double u, double v, const double* pano, double* az, double* el) {
  *az = (0.0 <= u && u < pano[0]) ? SV_TWO_PI - pano[2] * u : sv_nan();
  *el = (0.0 <= v && v < pano[1]) ? sv_atan(pano[3] - pano[2] * v) : sv_nan();  // atan2(h, 1)
}

__device__ __forceinline__ static void sv_bearing(double az, double el, double* b3) {
  double b, z, ca, sa;
  sv_sincos(el, &z, &b);
  sv_sincos(az, &sa, &ca);
  b3[0] = b * ca;
  b3[1] = b * sa;
  b3[2] = z;
}

__device__ __forceinline__ static void sv_triangulate(double az1, double el1, double az2, double el2, const double* F1,
                                                       const double* F2, double* X) {
  double sa1, ca1, se1, ce1, sa2, ca2, se2, ce2;
  sv_sincos(az1, &sa1, &ca1);
  sv_sincos(el1, &se1, &ce1);
  sv_sincos(az2, &sa2, &ca2);
  sv_sincos(el2, &se2, &ce2);
  const double v1[3] = {ca1, sa1, se1 / ce1};  // (cos psi, sin psi, tan theta)
  const double v2[3] = {ca2, sa2, se2 / ce2};
  const double pv[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
  const double mag = sqrt(pv[0] * pv[0] + pv[1] * pv[1] + pv[2] * pv[2]);
  const double nh[3] = {pv[0] / mag, pv[1] / mag, pv[2] / mag};
  const double a[3] = {v1[0], v1[1], v1[2]}, b[3] = {-v2[0], -v2[1], -v2[2]}, c[3] = {nh[0], nh[1], nh[2]};
  const double d[3] = {F2[0] - F1[0], F2[1] - F1[1], F2[2] - F1[2]};
  const double bxc[3] = {b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]};
  const double det = a[0] * bxc[0] + a[1] * bxc[1] + a[2] * bxc[2];
  const double l1 = (d[0] * bxc[0] + d[1] * bxc[1] + d[2] * bxc[2]) / det;
  const double bxd[3] = {b[1] * d[2] - b[2] * d[1], b[2] * d[0] - b[0] * d[2], b[0] * d[1] - b[1] * d[0]};
  const double lp = (a[0] * bxd[0] + a[1] * bxd[1] + a[2] * bxd[2]) / det;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double G1 = F1[k] + l1 * v1[k];
    X[k] = G1 + lp / 2.0 * nh[k];
  }
}

// OpenGV triangulation::triangulate2 (closed-form midpoint in frame 1); 2x2 system by Cramer's rule.
__device__ __forceinline__ static void sv_triangulate2(const double* f1, const double* f2, const double* t, const double* R,
                                                        double* X) {
  const double g[3] = {R[0] * f2[0] + R[1] * f2[1] + R[2] * f2[2], R[3] * f2[0] + R[4] * f2[1] + R[5] * f2[2],
                       R[6] * f2[0] + R[7] * f2[1] + R[8] * f2[2]};
  const double b0 = t[0] * f1[0] + t[1] * f1[1] + t[2] * f1[2], b1 = t[0] * g[0] + t[1] * g[1] + t[2] * g[2];
  const double a00 = f1[0] * f1[0] + f1[1] * f1[1] + f1[2] * f1[2], a10 = f1[0] * g[0] + f1[1] * g[1] + f1[2] * g[2];
  const double a01 = -a10, a11 = -(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
  const double det = a00 * a11 - a01 * a10;
  const double l0 = (b0 * a11 - a01 * b1) / det, l1 = (a00 * b1 - a10 * b0) / det;
#pragma unroll
  for (int k = 0; k < 3; ++k) X[k] = ((l0 * f1[k]) + (t[k] + l1 * g[k])) / 2.0;
}

__device__ __forceinline__ static bool sv_range_ok(const double* X, double min_range, double max_range) {
  const double nrm = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2] + 1.0);
  bool good = true;
  if (min_range > 0) good = good && (nrm >= min_range);
  if (max_range > 0) good = good && (nrm <= max_range);
  return good;
}

__device__ __forceinline__ static bool sv_pixel_gate(double ut, double vt, double ub, double vb, double min_disp,
                                                      double max_hdiff) {
  bool good = true;
  if (max_hdiff > 0) good = fabs(ut - ub) <= max_hdiff;
  if (min_disp >= 0) good = good && (vt - vb >= min_disp);
  return good;
}
