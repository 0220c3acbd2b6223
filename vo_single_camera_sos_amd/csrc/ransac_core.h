// Device-side numeric core of the absolute-pose RANSAC (K8/K10) and LM refinement (K9): counter-based sampler, Kneip
// P3P, real quartic roots, score, Cayley parametrisation, LM pieces.  Reference call sites replaced:
// pyopengv.absolute_pose_noncentral_ransac (omnistereo/pose_est_tools.py:785), absolute_pose_ransac (:915),
// *_optimize_nonlinear (:830, :937); score definition from pose_est_tools.py:150-203 (+ :181-185 non-central).
// Only + - * / sqrt and comparisons, fully parenthesised, built with -ffp-contract=off.  GENERATED from oracle/ransac_core.h by
// tests/gen_device_headers.py (same text, device prefixes): the CPU oracle evaluates the same operations in the same
// order, tests/test_abi.py checks that the two files stay identical.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdbool.h>
#include <stdint.h>

#define SV_LM_LAMBDA0 1e-3
#define SV_LM_MAX_TRIES 6
/* OpenGV's optimize_nonlinear sets ftol = xtol = 1e10 * eps on Eigen's LevenbergMarquardt */
#define SV_LM_FTOL 2.220446049250313e-6 /* relative cost decrease that ends the refinement */
#define SV_LM_XTOL 2.220446049250313e-6 /* relative step size that ends the refinement */
#define SV_SQRT_EPS 1.4901161193847656e-08
#define SV_RANSAC_PROB_FAIL 0.01 /* 1 - 0.99, OpenGV sac::Ransac default probability */

__device__ __forceinline__ static void sv_T_to_Rt(const double* T, double* R, double* t) {
  for (int i = 0; i < 3; ++i) {
    R[3 * i + 0] = T[4 * i + 0];
    R[3 * i + 1] = T[4 * i + 1];
    R[3 * i + 2] = T[4 * i + 2];
    t[i] = T[4 * i + 3];
  }
}

__device__ __forceinline__ static void sv_Rt_to_T(const double* R, const double* t, double* T) {
  for (int i = 0; i < 3; ++i) {
    T[4 * i + 0] = R[3 * i + 0];
    T[4 * i + 1] = R[3 * i + 1];
    T[4 * i + 2] = R[3 * i + 2];
    T[4 * i + 3] = t[i];
  }
}

/* ---- counter-based sampler ------------------------------------------------------------------ */
__device__ __forceinline__ static uint64_t sv_mix64(uint64_t seed, uint64_t iter, uint64_t draw) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (iter * 8ULL + draw + 1ULL);
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return z;
}

__device__ __forceinline__ static uint32_t sv_below(uint64_t z, uint32_t range) {
  return (uint32_t)(((z >> 32) * (uint64_t)range) >> 32);
}

/* Four distinct point indices: three solve points from ONE camera (the camera of the first
 * draw), the fourth (disambiguation) from all points.  Works in "perm position" space, perm
 * being the stable partition of the point indices by camera.  Returns 0 if impossible. */
__device__ __forceinline__ static int sv_sample4(const int32_t* cam, int32_t n, const int32_t* perm, const int32_t* cstart,
                              const int32_t* ccount, uint64_t seed, uint64_t it, int32_t* s) {
  if (n < 4) return 0;
  int32_t j0 = (int32_t)sv_below(sv_mix64(seed, it, 0), (uint32_t)n);
  int32_t i0 = perm[j0];
  int32_t c = cam ? cam[i0] : 0;
  int32_t cs = cstart[c], cc = ccount[c];
  if (cc < 3) return 0;
  int32_t pos0 = j0 - cs;
  int32_t a = (int32_t)sv_below(sv_mix64(seed, it, 1), (uint32_t)(cc - 1));
  if (a >= pos0) a++;
  int32_t b = (int32_t)sv_below(sv_mix64(seed, it, 2), (uint32_t)(cc - 2));
  int32_t lo = pos0 < a ? pos0 : a, hi = pos0 < a ? a : pos0;
  if (b >= lo) b++;
  if (b >= hi) b++;
  int32_t j1 = cs + a, j2 = cs + b;
  /* sort the three positions */
  int32_t s0 = j0, s1 = j1, s2 = j2, tmp;
  if (s0 > s1) { tmp = s0; s0 = s1; s1 = tmp; }
  if (s1 > s2) { tmp = s1; s1 = s2; s2 = tmp; }
  if (s0 > s1) { tmp = s0; s0 = s1; s1 = tmp; }
  int32_t d = (int32_t)sv_below(sv_mix64(seed, it, 3), (uint32_t)(n - 3));
  if (d >= s0) d++;
  if (d >= s1) d++;
  if (d >= s2) d++;
  s[0] = i0;
  s[1] = perm[j1];
  s[2] = perm[j2];
  s[3] = perm[d];
  return 1;
}

/* ---- score: 1 - f . normalize(R_c^T (R^T p - R^T t - o_c)) ----------------------------------
 * pose_est_tools.py:158-160,:177 build inverse = [R^T | -R^T t] and apply it to the homogeneous
 * point; :181-185 give the non-central extension; :199 the score. */
__device__ __forceinline__ static double sv_score(const double* R, const double* t, const double* f, const double* p,
                               const double* o, const double* Rc) {
  const double itx = -(((R[0] * t[0]) + (R[3] * t[1])) + (R[6] * t[2]));
  const double ity = -(((R[1] * t[0]) + (R[4] * t[1])) + (R[7] * t[2]));
  const double itz = -(((R[2] * t[0]) + (R[5] * t[1])) + (R[8] * t[2]));
  const double vx = (((R[0] * p[0]) + (R[3] * p[1])) + (R[6] * p[2])) + itx;
  const double vy = (((R[1] * p[0]) + (R[4] * p[1])) + (R[7] * p[2])) + ity;
  const double vz = (((R[2] * p[0]) + (R[5] * p[1])) + (R[8] * p[2])) + itz;
  const double wx = vx - o[0], wy = vy - o[1], wz = vz - o[2];
  const double ux = ((Rc[0] * wx) + (Rc[3] * wy)) + (Rc[6] * wz);
  const double uy = ((Rc[1] * wx) + (Rc[4] * wy)) + (Rc[7] * wz);
  const double uz = ((Rc[2] * wx) + (Rc[5] * wy)) + (Rc[8] * wz);
  const double nrm = sqrt(((ux * ux) + (uy * uy)) + (uz * uz));
  const double gx = ux / nrm, gy = uy / nrm, gz = uz / nrm;
  return 1.0 - (((f[0] * gx) + (f[1] * gy)) + (f[2] * gz));
}

/* Adaptive stop of OpenGV's sac::Ransac: continue while iterations < k,
 * k = log(1 - 0.99) / log(1 - w^4), w = best/N clamped to [eps, 1 - eps].  Evaluated without
 * log: iterations < k  <=>  (1 - w^4)^iterations > 0.01.  sv_adaptive_k returns the base
 * (1 - w^4); sv_ransac_continue evaluates the power by repeated squaring. */
__device__ __forceinline__ static double sv_adaptive_base(int32_t best, int32_t n) {
  const double w = (double)best / (double)n;
  double pno = 1.0 - (((w * w) * w) * w);
  const double eps = 2.220446049250313e-16;
  if (pno < eps) pno = eps;
  if (pno > 1.0 - eps) pno = 1.0 - eps;
  return pno;
}

// the same for 6-point samples (EPnP): 1 - w^6
__device__ __forceinline__ static double sv_adaptive_base6(int32_t best, int32_t n) {
  const double w = (double)best / (double)n;
  const double w2 = w * w;
  double pno = 1.0 - ((w2 * w2) * w2);
  const double eps = 2.220446049250313e-16;
  if (pno < eps) pno = eps;
  if (pno > 1.0 - eps) pno = 1.0 - eps;
  return pno;
}

// ... and for 2-point samples (TWOPT): 1 - w^2
__device__ __forceinline__ static double sv_adaptive_base2(int32_t best, int32_t n) {
  const double w = (double)best / (double)n;
  double pno = 1.0 - (w * w);
  const double eps = 2.220446049250313e-16;
  if (pno < eps) pno = eps;
  if (pno > 1.0 - eps) pno = 1.0 - eps;
  return pno;
}

__device__ __forceinline__ static int sv_ransac_continue(double base, int32_t iterations) {
  double result = 1.0, b = base;
  int32_t m = iterations;
  while (m > 0) {
    if (m & 1) result = result * b;
    b = b * b;
    m >>= 1;
  }
  return result > SV_RANSAC_PROB_FAIL;
}

/* ---- real roots of a quartic a[0] x^4 + a[1] x^3 + a[2] x^2 + a[3] x + a[4] ------------------ */
__device__ __forceinline__ static double sv_cubic_eval(double c2, double c1, double c0, double z) {
  return (((((z + c2) * z) + c1) * z) + c0);
}

/* One real root of z^3 + c2 z^2 + c1 z + c0 inside [lo, hi] with g(lo) <= 0 < g(hi):
 * Newton safeguarded by bisection, fixed iteration budget. */
__device__ __forceinline__ static double sv_cubic_root_bracketed(double c2, double c1, double c0, double lo, double hi) {
  double z = 0.5 * (lo + hi);
  for (int it = 0; it < 200; ++it) {
    const double g = sv_cubic_eval(c2, c1, c0, z);
    if (g == 0.0) return z;
    if (g > 0.0) hi = z; else lo = z;
    const double dg = ((((3.0 * z) + (2.0 * c2)) * z) + c1);
    double zn = z - (g / dg);
    if (!(zn > lo && zn < hi)) zn = 0.5 * (lo + hi); /* also catches NaN/inf */
    if (zn == z) return z;
    if (!(hi > lo)) return z;
    z = zn;
  }
  return z;
}

__device__ __forceinline__ static int sv_quartic(const double* a, double* roots) {
  const double a0 = a[0];
  if (!(a0 != 0.0) || !isfinite(a0)) return 0;
  const double b = a[1] / a0, c = a[2] / a0, d = a[3] / a0, e = a[4] / a0;
  if (!(isfinite(b) && isfinite(c) && isfinite(d) && isfinite(e))) return 0;
  const double b2 = b * b;
  const double p = c - (0.375 * b2);
  const double q = (d - ((0.5 * b) * c)) + ((0.125 * b2) * b);
  const double r = ((e - ((0.25 * b) * d)) + ((0.0625 * b2) * c)) - ((0.01171875 * b2) * b2);
  const double shift = 0.25 * b;
  double y[4];
  int n = 0;
  /* resolvent cubic z^3 + 2p z^2 + (p^2 - 4r) z - q^2, g(0) = -q^2 <= 0 */
  const double c2 = 2.0 * p, c1 = (p * p) - (4.0 * r), c0 = -(q * q);
  double z = 0.0;
  if (c0 != 0.0) {
    double bound = fabs(c2);
    if (fabs(c1) > bound) bound = fabs(c1);
    if (fabs(c0) > bound) bound = fabs(c0);
    bound = bound + 1.0;
    const double z1 = sv_cubic_root_bracketed(c2, c1, c0, 0.0, bound);
    /* deflate and take the largest real root (best conditioned factorisation) */
    const double al = c2 + z1, be = c1 + (z1 * al);
    const double disc = (al * al) - (4.0 * be);
    z = z1;
    if (disc >= 0.0) {
      const double sq = sqrt(disc);
      const double zb = 0.5 * (-al + sq);
      if (zb > z) z = zb;
    }
    /* two Newton polishing steps on the cubic */
    for (int k = 0; k < 2; ++k) {
      const double g = sv_cubic_eval(c2, c1, c0, z);
      const double dg = ((((3.0 * z) + (2.0 * c2)) * z) + c1);
      const double zn = z - (g / dg);
      if (isfinite(zn) && zn > 0.0) z = zn;
    }
  }
  if (!(z > 0.0)) {
    /* biquadratic y^4 + p y^2 + r = 0 */
    double disc = (p * p) - (4.0 * r);
    if (disc < 0.0) return 0;
    const double sq = sqrt(disc);
    const double w1 = 0.5 * (-p + sq), w2 = 0.5 * (-p - sq);
    if (w1 >= 0.0) {
      const double s = sqrt(w1);
      y[n++] = s;
      y[n++] = -s;
    }
    if (w2 >= 0.0 && sq != 0.0) {
      const double s = sqrt(w2);
      y[n++] = s;
      y[n++] = -s;
    }
  } else {
    const double s = sqrt(z);
    const double qs = q / s;
    const double u = 0.5 * ((p + z) - qs);
    const double v = 0.5 * ((p + z) + qs);
    double d1 = z - (4.0 * u), d2 = z - (4.0 * v);
    const double tol = 1e-12 * (z + fabs(4.0 * u) + fabs(4.0 * v));
    if (d1 < 0.0 && d1 > -tol) d1 = 0.0;
    if (d2 < 0.0 && d2 > -tol) d2 = 0.0;
    if (d1 >= 0.0) {
      const double sq = sqrt(d1);
      y[n++] = 0.5 * (-s + sq);
      y[n++] = 0.5 * (-s - sq);
    }
    if (d2 >= 0.0) {
      const double sq = sqrt(d2);
      y[n++] = 0.5 * (s + sq);
      y[n++] = 0.5 * (s - sq);
    }
  }
  for (int k = 0; k < n; ++k) {
    double x = y[k] - shift;
    for (int it = 0; it < 2; ++it) {
      const double h = ((((((x + b) * x) + c) * x) + d) * x) + e;
      const double dh = (((((4.0 * x) + (3.0 * b)) * x) + (2.0 * c)) * x) + d;
      const double xn = x - (h / dh);
      if (isfinite(xn)) x = xn;
    }
    roots[k] = x;
  }
  return n;
}

/* ---- Kneip P3P (CVPR 2011).  f1..f3 unit bearings in the camera frame, P1..P3 world points.
 * Returns up to 4 solutions: R_out[k] (row-major, camera->world) and C_out[k] (camera centre),
 * i.e. P = R x_cam + C.  -------------------------------------------------------------------- */
__device__ __forceinline__ static void sv_cross(const double* a, const double* b, double* o) {
  o[0] = (a[1] * b[2]) - (a[2] * b[1]);
  o[1] = (a[2] * b[0]) - (a[0] * b[2]);
  o[2] = (a[0] * b[1]) - (a[1] * b[0]);
}

__device__ __forceinline__ static double sv_dot(const double* a, const double* b) {
  return ((a[0] * b[0]) + (a[1] * b[1])) + (a[2] * b[2]);
}

__device__ __forceinline__ static int sv_p3p(const double* f1_in, const double* f2_in, const double* f3_in, const double* P1_in,
                          const double* P2_in, const double* P3_in, double* R_out, double* C_out) {
  double f1[3], f2[3], f3[3], P1[3], P2[3], P3[3];
  for (int i = 0; i < 3; ++i) {
    f1[i] = f1_in[i];
    f2[i] = f2_in[i];
    f3[i] = f3_in[i];
    P1[i] = P1_in[i];
    P2[i] = P2_in[i];
    P3[i] = P3_in[i];
  }
  double t1[3], t2[3], cr[3];
  for (int i = 0; i < 3; ++i) {
    t1[i] = P2[i] - P1[i];
    t2[i] = P3[i] - P1[i];
  }
  sv_cross(t1, t2, cr);
  if (!(sv_dot(cr, cr) > 0.0)) return 0; /* collinear world points */

  double e1[3], e2[3], e3[3], T[9], f3t[3];
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 3; ++i) e1[i] = f1[i];
    sv_cross(f1, f2, e3);
    const double n3 = sqrt(sv_dot(e3, e3));
    if (!(n3 > 0.0)) return 0;
    for (int i = 0; i < 3; ++i) e3[i] = e3[i] / n3;
    sv_cross(e3, e1, e2);
    for (int i = 0; i < 3; ++i) {
      T[i] = e1[i];
      T[3 + i] = e2[i];
      T[6 + i] = e3[i];
    }
    f3t[0] = sv_dot(T, f3);
    f3t[1] = sv_dot(T + 3, f3);
    f3t[2] = sv_dot(T + 6, f3);
    if (pass == 0 && f3t[2] > 0.0) {
      /* enforce f3[2] <= 0 by swapping the first two correspondences */
      for (int i = 0; i < 3; ++i) {
        double tmp = f1[i];
        f1[i] = f2[i];
        f2[i] = tmp;
        tmp = P1[i];
        P1[i] = P2[i];
        P2[i] = tmp;
      }
      continue;
    }
    break;
  }

  double n1[3], n2[3], n3v[3], N[9], d3[3];
  for (int i = 0; i < 3; ++i) {
    n1[i] = P2[i] - P1[i];
    d3[i] = P3[i] - P1[i];
  }
  const double d_12 = sqrt(sv_dot(n1, n1));
  for (int i = 0; i < 3; ++i) n1[i] = n1[i] / d_12;
  sv_cross(n1, d3, n3v);
  const double nn3 = sqrt(sv_dot(n3v, n3v));
  if (!(nn3 > 0.0)) return 0;
  for (int i = 0; i < 3; ++i) n3v[i] = n3v[i] / nn3;
  sv_cross(n3v, n1, n2);
  for (int i = 0; i < 3; ++i) {
    N[i] = n1[i];
    N[3 + i] = n2[i];
    N[6 + i] = n3v[i];
  }
  const double p_1 = sv_dot(N, d3);
  const double p_2 = sv_dot(N + 3, d3);

  if (!(f3t[2] != 0.0)) return 0;
  const double f_1 = f3t[0] / f3t[2];
  const double f_2 = f3t[1] / f3t[2];
  if (!(f_2 != 0.0)) return 0;

  const double cos_beta = sv_dot(f1, f2);
  double b = (1.0 / (1.0 - (cos_beta * cos_beta))) - 1.0;
  if (!(b >= 0.0)) return 0;
  b = (cos_beta < 0.0) ? -sqrt(b) : sqrt(b);

  const double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2;
  const double p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
  const double p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2;
  const double d_12_pw2 = d_12 * d_12, b_pw2 = b * b;

  double fac[5];
  fac[0] = ((-(f_2_pw2 * p_2_pw4)) - (p_2_pw4 * f_1_pw2)) - p_2_pw4;
  fac[1] = ((((2.0 * p_2_pw3) * d_12) * b) + ((((2.0 * f_2_pw2) * p_2_pw3) * d_12) * b)) -
           ((((2.0 * f_2) * p_2_pw3) * f_1) * d_12);
  fac[2] = ((((((((((-((f_2_pw2 * p_2_pw2) * p_1_pw2)) - (((f_2_pw2 * p_2_pw2) * d_12_pw2) * b_pw2)) -
                  ((f_2_pw2 * p_2_pw2) * d_12_pw2)) +
                 (f_2_pw2 * p_2_pw4)) +
                (p_2_pw4 * f_1_pw2)) +
               (((2.0 * p_1) * p_2_pw2) * d_12)) +
              ((((((2.0 * f_1) * f_2) * p_1) * p_2_pw2) * d_12) * b)) -
             ((p_2_pw2 * p_1_pw2) * f_1_pw2)) +
            ((((2.0 * p_1) * p_2_pw2) * f_2_pw2) * d_12)) -
           ((p_2_pw2 * d_12_pw2) * b_pw2)) -
          ((2.0 * p_1_pw2) * p_2_pw2);
  fac[3] = ((((((2.0 * p_1_pw2) * p_2) * d_12) * b) + ((((2.0 * f_2) * p_2_pw3) * f_1) * d_12)) -
            ((((2.0 * f_2_pw2) * p_2_pw3) * d_12) * b)) -
           ((((2.0 * p_1) * p_2) * d_12_pw2) * b);
  fac[4] = ((((((((-((((((2.0 * f_2) * p_2_pw2) * f_1) * p_1) * d_12) * b)) + ((f_2_pw2 * p_2_pw2) * d_12_pw2)) +
                ((2.0 * p_1_pw3) * d_12)) -
               (p_1_pw2 * d_12_pw2)) +
              ((f_2_pw2 * p_2_pw2) * p_1_pw2)) -
             p_1_pw4) -
            ((((2.0 * f_2_pw2) * p_2_pw2) * p_1) * d_12)) +
           ((p_2_pw2 * f_1_pw2) * p_1_pw2)) +
          (((f_2_pw2 * p_2_pw2) * d_12_pw2) * b_pw2);

  double roots[4];
  const int nr = sv_quartic(fac, roots);
  int ns = 0;
  for (int k = 0; k < nr; ++k) {
    const double cos_theta = roots[k];
    const double st2 = 1.0 - (cos_theta * cos_theta);
    if (!(st2 >= 0.0)) continue;
    const double cot_alpha = ((((-f_1) * p_1) / f_2) - (cos_theta * p_2) + (d_12 * b)) /
                             (((((-f_1) * cos_theta) * p_2) / f_2) + p_1 - d_12);
    if (!isfinite(cot_alpha)) continue;
    const double sin_theta = sqrt(st2);
    const double sin_alpha = sqrt(1.0 / ((cot_alpha * cot_alpha) + 1.0));
    double cos_alpha = sqrt(1.0 - (sin_alpha * sin_alpha));
    if (cot_alpha < 0.0) cos_alpha = -cos_alpha;
    const double sb = (sin_alpha * b) + cos_alpha;
    double Cn[3];
    Cn[0] = (d_12 * cos_alpha) * sb;
    Cn[1] = ((cos_theta * d_12) * sin_alpha) * sb;
    Cn[2] = ((sin_theta * d_12) * sin_alpha) * sb;
    double* C = C_out + 3 * ns;
    /* C = P1 + N^T Cn */
    for (int i = 0; i < 3; ++i) C[i] = P1[i] + (((N[i] * Cn[0]) + (N[3 + i] * Cn[1])) + (N[6 + i] * Cn[2]));
    /* Q = R_kneip^T, R = N^T Q T */
    double Q[9];
    Q[0] = -cos_alpha;
    Q[1] = sin_alpha;
    Q[2] = 0.0;
    Q[3] = -(sin_alpha * cos_theta);
    Q[4] = -(cos_alpha * cos_theta);
    Q[5] = -sin_theta;
    Q[6] = -(sin_alpha * sin_theta);
    Q[7] = -(cos_alpha * sin_theta);
    Q[8] = cos_theta;
    double QT[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        QT[3 * i + j] = ((Q[3 * i] * T[j]) + (Q[3 * i + 1] * T[3 + j])) + (Q[3 * i + 2] * T[6 + j]);
    double* R = R_out + 9 * ns;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        R[3 * i + j] = ((N[i] * QT[j]) + (N[3 + i] * QT[3 + j])) + (N[6 + i] * QT[6 + j]);
    int fin = 1;
    for (int i = 0; i < 9; ++i) fin &= isfinite(R[i]) ? 1 : 0;
    for (int i = 0; i < 3; ++i) fin &= isfinite(C[i]) ? 1 : 0;
    if (fin) ns++;
  }
  return ns;
}

/* One RANSAC hypothesis: sample, solve in the sampled camera, move to the body frame, pick the
 * solution with the smallest score on the 4th point (first wins ties). */
__device__ __forceinline__ static int sv_hypothesis(const double* f, const double* p, const int32_t* cam, const double* cam_off,
                                 const double* cam_rot, int32_t n, const int32_t* perm, const int32_t* cstart,
                                 const int32_t* ccount, uint64_t seed, uint64_t it, double* R_best, double* t_best) {
  int32_t s[4];
  if (!sv_sample4(cam, n, perm, cstart, ccount, seed, it, s)) return 0;
  const int32_t c = cam ? cam[s[0]] : 0;
  const double* o = cam_off + 3 * c;
  const double* Rc = cam_rot + 9 * c;
  double Rs[36], Cs[12];
  const int ns = sv_p3p(f + 3 * s[0], f + 3 * s[1], f + 3 * s[2], p + 3 * s[0], p + 3 * s[1], p + 3 * s[2], Rs, Cs);
  const int32_t c3 = cam ? cam[s[3]] : 0;
  double best = 0.0;
  int found = 0;
  for (int k = 0; k < ns; ++k) {
    /* camera pose (Rw, Cw) -> body pose: R = Rw Rc^T, t = Cw - R o */
    const double* Rw = Rs + 9 * k;
    const double* Cw = Cs + 3 * k;
    double R[9], t[3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        R[3 * i + j] = ((Rw[3 * i] * Rc[3 * j]) + (Rw[3 * i + 1] * Rc[3 * j + 1])) + (Rw[3 * i + 2] * Rc[3 * j + 2]);
    for (int i = 0; i < 3; ++i) t[i] = Cw[i] - (((R[3 * i] * o[0]) + (R[3 * i + 1] * o[1])) + (R[3 * i + 2] * o[2]));
    const double sc = sv_score(R, t, f + 3 * s[3], p + 3 * s[3], cam_off + 3 * c3, cam_rot + 9 * c3);
    if (!(sc == sc)) continue; /* NaN */
    if (!found || sc < best) {
      found = 1;
      best = sc;
      for (int i = 0; i < 9; ++i) R_best[i] = R[i];
      for (int i = 0; i < 3; ++i) t_best[i] = t[i];
    }
  }
  return found;
}

/* ---- Cayley parametrisation -------------------------------------------------------------- */
__device__ __forceinline__ static void sv_rot2cayley(const double* R, double* c) {
  const double s = 1.0 + ((R[0] + R[4]) + R[8]);
  c[0] = (R[7] - R[5]) / s;
  c[1] = (R[2] - R[6]) / s;
  c[2] = (R[3] - R[1]) / s;
}

__device__ __forceinline__ static void sv_cayley2rot(const double* c, double* R) {
  const double x = c[0], y = c[1], z = c[2];
  const double xx = x * x, yy = y * y, zz = z * z;
  const double s = 1.0 + ((xx + yy) + zz);
  R[0] = (((1.0 + xx) - yy) - zz) / s;
  R[1] = (2.0 * ((x * y) - z)) / s;
  R[2] = (2.0 * ((x * z) + y)) / s;
  R[3] = (2.0 * ((x * y) + z)) / s;
  R[4] = (((1.0 - xx) + yy) - zz) / s;
  R[5] = (2.0 * ((y * z) - x)) / s;
  R[6] = (2.0 * ((x * z) - y)) / s;
  R[7] = (2.0 * ((y * z) + x)) / s;
  R[8] = (((1.0 - xx) - yy) + zz) / s;
}

__device__ __forceinline__ static double sv_residual(const double* x, const double* f, const double* p, const double* o,
                                  const double* Rc) {
  double R[9];
  sv_cayley2rot(x + 3, R);
  return sv_score(R, x, f, p, o, Rc);
}

__device__ __forceinline__ static void sv_residual_jac(const double* x, const double* f, const double* p, const double* o,
                                    const double* Rc, double* r, double* J, double* Hq) {
  /* Analytic Jacobian of r = 1 - f . u/|u|, u = Rc^T (R(c)^T (p - t) - o), wrt (t, Cayley c).
   * (OpenGV differentiates numerically; the forward-difference noise stalls LM ~1e-5 short of the
   * minimum and makes the result depend on last-bit input noise, so the exact gradient is used.) */
  *r = sv_residual(x, f, p, o, Rc);
  double R[9];
  sv_cayley2rot(x + 3, R);
  const double c0 = x[3], c1 = x[4], c2 = x[5];
  const double s = 1.0 + (((c0 * c0) + (c1 * c1)) + (c2 * c2));
  const double d0 = p[0] - x[0], d1 = p[1] - x[1], d2 = p[2] - x[2];
  const double w0 = ((R[0] * d0) + (R[3] * d1)) + (R[6] * d2);
  const double w1 = ((R[1] * d0) + (R[4] * d1)) + (R[7] * d2);
  const double w2 = ((R[2] * d0) + (R[5] * d1)) + (R[8] * d2);
  const double e0 = w0 - o[0], e1 = w1 - o[1], e2 = w2 - o[2];
  const double u0 = ((Rc[0] * e0) + (Rc[3] * e1)) + (Rc[6] * e2);
  const double u1 = ((Rc[1] * e0) + (Rc[4] * e1)) + (Rc[7] * e2);
  const double u2 = ((Rc[2] * e0) + (Rc[5] * e1)) + (Rc[8] * e2);
  const double nrm = sqrt(((u0 * u0) + (u1 * u1)) + (u2 * u2));
  const double g0 = u0 / nrm, g1 = u1 / nrm, g2 = u2 / nrm;
  const double fg = ((f[0] * g0) + (f[1] * g1)) + (f[2] * g2);
  const double a0 = (f[0] - (fg * g0)) / nrm, a1 = (f[1] - (fg * g1)) / nrm, a2 = (f[2] - (fg * g2)) / nrm;
  const double b0 = ((Rc[0] * a0) + (Rc[1] * a1)) + (Rc[2] * a2);
  const double b1 = ((Rc[3] * a0) + (Rc[4] * a1)) + (Rc[5] * a2);
  const double b2 = ((Rc[6] * a0) + (Rc[7] * a1)) + (Rc[8] * a2);
  /* dr/dt = R b */
  J[0] = ((R[0] * b0) + (R[1] * b1)) + (R[2] * b2);
  J[1] = ((R[3] * b0) + (R[4] * b1)) + (R[5] * b2);
  J[2] = ((R[6] * b0) + (R[7] * b1)) + (R[8] * b2);
  /* dr/dc_k = -b . dw/dc_k,  dw/dc_k = ((dN/dc_k)^T d - 2 c_k w) / s,
   * (dN/dc_k)^T d = -2 c_k d - 2 e_k x d + 2 (c d_k + e_k (c . d)) */
  const double cd = ((c0 * d0) + (c1 * d1)) + (c2 * d2);
  double Q[3][3];
  {
    const double m0 = ((-2.0 * c0) * d0) + (2.0 * ((c0 * d0) + cd));
    const double m1 = (((-2.0 * c0) * d1) - (2.0 * (-d2))) + (2.0 * (c1 * d0));
    const double m2 = (((-2.0 * c0) * d2) - (2.0 * d1)) + (2.0 * (c2 * d0));
    Q[0][0] = (m0 - ((2.0 * c0) * w0)) / s;
    Q[0][1] = (m1 - ((2.0 * c0) * w1)) / s;
    Q[0][2] = (m2 - ((2.0 * c0) * w2)) / s;
  }
  {
    const double m0 = (((-2.0 * c1) * d0) - (2.0 * d2)) + (2.0 * (c0 * d1));
    const double m1 = ((-2.0 * c1) * d1) + (2.0 * ((c1 * d1) + cd));
    const double m2 = (((-2.0 * c1) * d2) - (2.0 * (-d0))) + (2.0 * (c2 * d1));
    Q[1][0] = (m0 - ((2.0 * c1) * w0)) / s;
    Q[1][1] = (m1 - ((2.0 * c1) * w1)) / s;
    Q[1][2] = (m2 - ((2.0 * c1) * w2)) / s;
  }
  {
    const double m0 = (((-2.0 * c2) * d0) - (2.0 * (-d1))) + (2.0 * (c0 * d2));
    const double m1 = (((-2.0 * c2) * d1) - (2.0 * d0)) + (2.0 * (c1 * d2));
    const double m2 = ((-2.0 * c2) * d2) + (2.0 * ((c2 * d2) + cd));
    Q[2][0] = (m0 - ((2.0 * c2) * w0)) / s;
    Q[2][1] = (m1 - ((2.0 * c2) * w1)) / s;
    Q[2][2] = (m2 - ((2.0 * c2) * w2)) / s;
  }
  J[3] = -(((b0 * Q[0][0]) + (b1 * Q[0][1])) + (b2 * Q[0][2]));
  J[4] = -(((b0 * Q[1][0]) + (b1 * Q[1][1])) + (b2 * Q[1][2]));
  J[5] = -(((b0 * Q[2][0]) + (b1 * Q[2][1])) + (b2 * Q[2][2]));
  /* Second-order term of the quartic cost: the cost is sum r^2 with r = 1 - cos(angle) ~ |e|^2 / 2 (e = tangent
   * error), so its Hessian is sum (grad r grad r^T + r G^T G) up to O(|e|) relative, G = d(u/|u|)/dx.  Plain
   * Gauss-Newton keeps only the first term and crawls (linear convergence); adding r G^T G makes the step a
   * Newton step.  Hq = r G^T G, packed upper triangle like the normal matrix. */
  double V[6][3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      V[j][i] = -(((Rc[i] * R[3 * j]) + (Rc[3 + i] * R[3 * j + 1])) + (Rc[6 + i] * R[3 * j + 2]));
      V[3 + j][i] = ((Rc[i] * Q[j][0]) + (Rc[3 + i] * Q[j][1])) + (Rc[6 + i] * Q[j][2]);
    }
  double G[6][3];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double gv = ((g0 * V[j][0]) + (g1 * V[j][1])) + (g2 * V[j][2]);
    G[j][0] = (V[j][0] - (g0 * gv)) / nrm;
    G[j][1] = (V[j][1] - (g1 * gv)) / nrm;
    G[j][2] = (V[j][2] - (g2 * gv)) / nrm;
  }
  {
    const double rr = *r;
    int a = 0;
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int v = u; v < 6; ++v) Hq[a++] = rr * (((G[u][0] * G[v][0]) + (G[u][1] * G[v][1])) + (G[u][2] * G[v][2]));
  }
}

/* Solve (A + lambda diag(A)) dx = -g for the packed upper triangle A (21) by LDL^T without
 * pivoting.  Returns 0 if a pivot is not positive. */
__device__ __forceinline__ static int sv_solve_damped(const double* Apacked, const double* g, double lambda, double* dx) {
  // Every loop is fully unrolled so that M, L, yv live in registers (runtime-indexed local arrays would go to
  // scratch memory); a failed pivot only clears `ok`, the arithmetic of the successful case is unchanged.
  double M[36];
  {
    int a = 0;
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int v = u; v < 6; ++v) {
        M[6 * u + v] = Apacked[a];
        M[6 * v + u] = Apacked[a];
        a++;
      }
  }
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    double dg = M[7 * u];
    if (dg < 1e-30) dg = 1e-30;
    M[7 * u] = M[7 * u] + (lambda * dg);
  }
  /* Cholesky M = L L^T */
  double L[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) L[i] = 0.0;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = M[7 * j];
#pragma unroll
    for (int k = 0; k < j; ++k) s = s - (L[6 * j + k] * L[6 * j + k]);
    ok = ok && (s > 0.0);
    const double ljj = sqrt(s);
    L[7 * j] = ljj;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = M[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t = t - (L[6 * i + k] * L[6 * j + k]);
      L[6 * i + j] = t / ljj;
    }
  }
  double yv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double t = -g[i];
#pragma unroll
    for (int k = 0; k < i; ++k) t = t - (L[6 * i + k] * yv[k]);
    yv[i] = t / L[7 * i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double t = yv[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) t = t - (L[6 * k + i] * dx[k]);
    dx[i] = t / L[7 * i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) ok = ok && isfinite(dx[i]);
  return ok ? 1 : 0;
}

/* ---- TWOPT: the translation of a camera whose rotation is KNOWN, from two correspondences ---------------
 * (pyopengv.absolute_pose_ransac(..., "TWOPT", ...), omnistereo/pose_est_tools.py:95-107: OpenGV's twopt takes the
 * rotation prior of its adapter, which the binding leaves at the identity.)  Restated as the least-squares meeting
 * point of the two rays through the world points: minimise sum |(I - f_i f_i^T)(p_i - t)|^2, f_i the bearings rotated
 * into the world frame -- a 3 x 3 linear system, solved by Cramer's rule.  Returns 0 for (nearly) parallel rays. */
__device__ __forceinline__ static int sv_twopt(const double* f1, const double* f2, const double* p1, const double* p2, const double* R,
                            double* t) {
  double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  for (int k = 0; k < 2; ++k) {
    const double* f = k ? f2 : f1;
    const double* p = k ? p2 : p1;
    double g[3]; /* bearing in the world frame */
    for (int i = 0; i < 3; ++i) g[i] = ((R[3 * i] * f[0]) + (R[3 * i + 1] * f[1])) + (R[3 * i + 2] * f[2]);
    const double gp = ((g[0] * p[0]) + (g[1] * p[1])) + (g[2] * p[2]);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) A[3 * i + j] = A[3 * i + j] + ((i == j ? 1.0 : 0.0) - (g[i] * g[j]));
      b[i] = b[i] + (p[i] - (g[i] * gp));
    }
  }
  const double c00 = (A[4] * A[8]) - (A[5] * A[7]), c01 = (A[5] * A[6]) - (A[3] * A[8]), c02 = (A[3] * A[7]) - (A[4] * A[6]);
  const double det = ((A[0] * c00) + (A[1] * c01)) + (A[2] * c02);
  if (!(det > 1e-12)) return 0; /* A = 2 I - g1 g1^T - g2 g2^T has determinant sin^2 of the angle between the rays */
  const double c10 = (A[2] * A[7]) - (A[1] * A[8]), c11 = (A[0] * A[8]) - (A[2] * A[6]), c12 = (A[1] * A[6]) - (A[0] * A[7]);
  const double c20 = (A[1] * A[5]) - (A[2] * A[4]), c21 = (A[2] * A[3]) - (A[0] * A[5]), c22 = (A[0] * A[4]) - (A[1] * A[3]);
  t[0] = (((c00 * b[0]) + (c10 * b[1])) + (c20 * b[2])) / det;
  t[1] = (((c01 * b[0]) + (c11 * b[1])) + (c21 * b[2])) / det;
  t[2] = (((c02 * b[0]) + (c12 * b[1])) + (c22 * b[2])) / det;
  return 1;
}

/* Sum of 256 partials: four groups of 64, each by the binary tree v[l] += v[l + o], o = 32..1,
 * then ((w0 + w1) + w2) + w3. */
__device__ __forceinline__ static double sv_tree_sum256(const double* part) {
  double w[4];
  for (int k = 0; k < 4; ++k) {
    double v[64];
    for (int l = 0; l < 64; ++l) v[l] = part[64 * k + l];
    for (int o = 32; o > 0; o >>= 1)
      for (int l = 0; l < o; ++l) v[l] = v[l] + v[l + o];
    w[k] = v[0];
  }
  return ((w[0] + w[1]) + w[2]) + w[3];
}
