// Generalised P3P, the minimal solver of the non-central absolute-pose RANSAC (the reference's
// absolute_pose_noncentral_ransac "will ALWAYS use GP3P", omnistereo/pose_est_tools.py:696, :785): three quadrics in
// the three depths -> an octic in the first one (resultants) -> Laguerre roots -> Newton polish -> triangle alignment;
// up to 8 poses, the fourth correspondence picks one.  GENERATED from oracle/gp3p_core.h by tests/gen_device_headers.py (same text,
// device prefixes): the CPU oracle evaluates the same operations in the same order, tests/test_abi.py checks that the
// two files stay identical.  The derivation and the independent checks are documented in the oracle header.
#pragma once
#include <hip/hip_runtime.h>
#include "epnp_core.h"

#define SV_GP3P_MAXSOL 8

typedef struct {
  double re, im;
} sv_cplx;

__device__ static sv_cplx sv_c(double re, double im) {
  sv_cplx z;
  z.re = re;
  z.im = im;
  return z;
}
__device__ static sv_cplx sv_cadd(sv_cplx a, sv_cplx b) { return sv_c(a.re + b.re, a.im + b.im); }
__device__ static sv_cplx sv_csub(sv_cplx a, sv_cplx b) { return sv_c(a.re - b.re, a.im - b.im); }
__device__ static sv_cplx sv_cmul(sv_cplx a, sv_cplx b) {
  return sv_c((a.re * b.re) - (a.im * b.im), (a.re * b.im) + (a.im * b.re));
}
__device__ static sv_cplx sv_cscale(sv_cplx a, double s) { return sv_c(a.re * s, a.im * s); }
__device__ static double sv_cabs(sv_cplx a) { return sqrt((a.re * a.re) + (a.im * a.im)); }
__device__ static sv_cplx sv_cdiv(sv_cplx a, sv_cplx b) { /* Smith's formula */
  if (fabs(b.re) >= fabs(b.im)) {
    const double r = b.im / b.re, den = b.re + (r * b.im);
    return sv_c((a.re + (r * a.im)) / den, (a.im - (r * a.re)) / den);
  }
  const double r = b.re / b.im, den = b.im + (r * b.re);
  return sv_c(((a.re * r) + a.im) / den, ((a.im * r) - a.re) / den);
}
__device__ static sv_cplx sv_csqrt(sv_cplx z) { /* principal square root from real square roots */
  if (z.re == 0.0 && z.im == 0.0) return sv_c(0.0, 0.0);
  const double x = fabs(z.re), y = fabs(z.im);
  double w;
  if (x >= y) {
    const double r = y / x;
    w = sqrt(x) * sqrt(0.5 * (1.0 + sqrt(1.0 + (r * r))));
  } else {
    const double r = x / y;
    w = sqrt(y) * sqrt(0.5 * (r + sqrt(1.0 + (r * r))));
  }
  if (z.re >= 0.0) return sv_c(w, z.im / (2.0 * w));
  const double im = z.im >= 0.0 ? w : -w;
  return sv_c(z.im / (2.0 * im), im);
}

/* Laguerre's iteration on a polynomial of degree m with REAL coefficients a[0..m] (a[m] leading), from *x; the classic
 * formulation with a fractional step every tenth iteration to break limit cycles.  Magnitudes that only feed the
 * rounding-error bound of the evaluation (Adams' running bound) or a comparison are taken in the 1-norm / squared:
 * a square root per coefficient and three per iteration were more than half of the instructions of an iteration. */
__device__ static double sv_cabs1(sv_cplx a) { return fabs(a.re) + fabs(a.im); }
__device__ static void sv_laguerre(const double* a, int m, sv_cplx* x) {
  const double frac[9] = {0.0, 0.5, 0.25, 0.75, 0.13, 0.38, 0.62, 0.88, 1.0};
  for (int iter = 1; iter <= 80; ++iter) {
    sv_cplx b = sv_c(a[m], 0.0), d = sv_c(0.0, 0.0), f = sv_c(0.0, 0.0);
    double err = sv_cabs1(b);
    const double abx = sv_cabs1(*x);
    for (int j = m - 1; j >= 0; --j) {
      f = sv_cadd(sv_cmul(*x, f), d);
      d = sv_cadd(sv_cmul(*x, d), b);
      b = sv_cmul(*x, b);
      b.re = b.re + a[j];
      err = sv_cabs1(b) + (abx * err);
    }
    err = err * 1e-15;
    if (sv_cabs1(b) <= err) return; /* on a root */
    const sv_cplx g = sv_cdiv(d, b), g2 = sv_cmul(g, g);
    const sv_cplx h = sv_csub(g2, sv_cscale(sv_cdiv(f, b), 2.0));
    const sv_cplx sq = sv_csqrt(sv_cscale(sv_csub(sv_cscale(h, (double)m), g2), (double)(m - 1)));
    sv_cplx gp = sv_cadd(g, sq);
    const sv_cplx gm = sv_csub(g, sq);
    double abp = (gp.re * gp.re) + (gp.im * gp.im);
    const double abm = (gm.re * gm.re) + (gm.im * gm.im);
    if (abp < abm) {
      gp = gm;
      abp = abm;
    }
    const sv_cplx dx = abp > 0.0 ? sv_cdiv(sv_c((double)m, 0.0), gp) : sv_c((1.0 + abx) * 0.6, (1.0 + abx) * 0.8);
    const sv_cplx x1 = sv_csub(*x, dx);
    if (x->re == x1.re && x->im == x1.im) return; /* converged */
    if (iter % 10 != 0) *x = x1;
    else *x = sv_csub(*x, sv_cscale(dx, frac[iter / 10]));
  }
}

/* All roots of the real polynomial c[0] + c[1] x + ... + c[m] x^m (m <= 10, c[m] != 0): Laguerre from 0 on the
 * deflated polynomial, which stays REAL -- a real root is divided out as (x - r), a complex one together with its
 * conjugate as the real quadratic x^2 - 2 Re(z) x + |z|^2 (one iteration run finds both: conjugate roots of a real
 * polynomial need not be searched twice); once four roots are left, the real ones among them come from the closed-form
 * quartic solver.  Then the roots that can still turn out real (|Im| <= 1e-3 (1 + |Re|): every
 * caller discards the others) are polished on the undeflated polynomial. */
__device__ static void sv_poly_roots(const double* c, int m, sv_cplx* roots) {
  double ad[11];
  for (int j = 0; j <= m; ++j) ad[j] = c[j];
  int deg = m, nr = 0;
  while (deg >= 1) {
    if (deg == 4) { /* the last four in closed form -- every caller wants the REAL roots only (sv_quartic: Ferrari) */
      const double q4[5] = {ad[4], ad[3], ad[2], ad[1], ad[0]};
      double rr[4];
      const int k4 = sv_quartic(q4, rr);
      for (int k = 0; k < 4; ++k)
        if (k < k4) roots[nr++] = sv_c(rr[k], 0.0);
      while (nr < m) roots[nr++] = sv_c(0.0, 1.0); /* stand-ins for the complex ones: skipped by every caller */
      break;
    }
    sv_cplx x = sv_c(0.0, 0.0);
    sv_laguerre(ad, deg, &x);
    if (deg == 1 || fabs(x.im) <= (1e-14 * (1.0 + fabs(x.re)))) {
      x.im = 0.0;
      roots[nr++] = x;
      double b = ad[deg];
      for (int jj = deg - 1; jj >= 0; --jj) { /* deflate by (x - r) */
        const double t = ad[jj];
        ad[jj] = b;
        b = (x.re * b) + t;
      }
      deg = deg - 1;
    } else {
      roots[nr++] = x;
      roots[nr++] = sv_c(x.re, -x.im);
      const double p2 = 2.0 * x.re, q = (x.re * x.re) + (x.im * x.im); /* divide by x^2 - p2 x + q */
      double b1 = 0.0, b0 = 0.0; /* quotient coefficients of the two degrees above the current one */
      for (int jj = deg; jj >= 2; --jj) {
        const double t = (ad[jj] + (p2 * b0)) - (q * b1);
        b1 = b0;
        b0 = t;
        ad[jj] = t; /* quotient coefficient of x^(jj - 2), stored two places up for now */
      }
      for (int jj = 0; jj <= deg - 2; ++jj) ad[jj] = ad[jj + 2];
      deg = deg - 2;
    }
  }
  for (int j = 0; j < m; ++j)
    if (fabs(roots[j].im) <= (1e-3 * (1.0 + fabs(roots[j].re)))) sv_laguerre(c, m, &roots[j]);
}

/* small real-polynomial helpers: p (degree dp) times q (degree dq) accumulated into out with weight w */
__device__ static void sv_pmul_acc(const double* p, int dp, const double* q, int dq, double w, double* out) {
  for (int i = 0; i <= dp; ++i)
    for (int j = 0; j <= dq; ++j) out[i + j] = out[i + j] + (w * (p[i] * q[j]));
}

/* The coefficients of one quadric E_ij (see the header): {c, a, b, k}. */
__device__ static void sv_gp3p_pair(const double* fi, const double* fj, const double* oi, const double* oj, const double* Pi,
                                 const double* Pj, double* e) {
  const double d0 = oi[0] - oj[0], d1 = oi[1] - oj[1], d2 = oi[2] - oj[2];
  const double q0 = Pi[0] - Pj[0], q1 = Pi[1] - Pj[1], q2 = Pi[2] - Pj[2];
  e[0] = ((fi[0] * fj[0]) + (fi[1] * fj[1])) + (fi[2] * fj[2]);
  e[1] = ((fi[0] * d0) + (fi[1] * d1)) + (fi[2] * d2);
  e[2] = ((fj[0] * d0) + (fj[1] * d1)) + (fj[2] * d2);
  e[3] = (((d0 * d0) + (d1 * d1)) + (d2 * d2)) - (((q0 * q0) + (q1 * q1)) + (q2 * q2));
}
__device__ static double sv_gp3p_eval(const double* e, double li, double lj) {
  return (((((li * li) + (lj * lj)) - (((2.0 * e[0]) * li) * lj)) + ((2.0 * e[1]) * li)) - ((2.0 * e[2]) * lj)) + e[3];
}

/* The octic in l_1 (oct[0..8], oct[8] leading) and the polynomials A (degree <= 3), B (degree <= 4) of the linear
 * relation A l_3 + B = 0.  e12, e13, e23: quadric coefficients {c, a, b, k}. */
__device__ static void sv_gp3p_octic(const double* e12, const double* e13, const double* e23, double* oct, double* Apoly,
                                  double* Bpoly) {
  /* E_12 as a quadratic in l_2: l_2^2 + p l_2 + q,  p = -2 c12 l_1 - 2 b12,  q = l_1^2 + 2 a12 l_1 + k12 */
  const double p[2] = {-(2.0 * e12[2]), -(2.0 * e12[0])};
  const double q[3] = {e12[3], 2.0 * e12[1], 1.0};
  /* E_23 as a quadratic in l_2: l_2^2 + r l_2 + s,  r = -2 c23 l_3 + 2 a23,  s = l_3^2 - 2 b23 l_3 + k23 (in l_3) */
  const double r[2] = {2.0 * e23[1], -(2.0 * e23[0])};
  const double s[3] = {e23[3], -(2.0 * e23[2]), 1.0};
  /* Resultant (q - s)^2 + (p - r)(p s - r q) as rho[k][*]: coefficient of l_3^k, a polynomial in l_1 (degree <= 4) */
  double rho[5][5];
  for (int k = 0; k < 5; ++k)
    for (int j = 0; j < 5; ++j) rho[k][j] = 0.0;
  /* (q - s)^2 = q^2 - 2 q s + s^2 */
  {
    double qq[5] = {0, 0, 0, 0, 0};
    sv_pmul_acc(q, 2, q, 2, 1.0, qq);
    for (int j = 0; j < 5; ++j) rho[0][j] = rho[0][j] + qq[j];
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 3; ++j) rho[k][j] = rho[k][j] - ((2.0 * s[k]) * q[j]);
    for (int k = 0; k < 3; ++k)
      for (int k2 = 0; k2 < 3; ++k2) rho[k + k2][0] = rho[k + k2][0] + (s[k] * s[k2]);
  }
  /* (p - r)(p s - r q) = p^2 s - p r q - r p s + r^2 q */
  {
    double pq[4] = {0, 0, 0, 0}, pp[3] = {0, 0, 0};
    sv_pmul_acc(p, 1, q, 2, 1.0, pq);
    sv_pmul_acc(p, 1, p, 1, 1.0, pp);
    for (int k = 0; k < 3; ++k) /* + s p^2 */
      for (int j = 0; j < 3; ++j) rho[k][j] = rho[k][j] + (s[k] * pp[j]);
    for (int k = 0; k < 2; ++k) /* - r (p q) */
      for (int j = 0; j < 4; ++j) rho[k][j] = rho[k][j] - (r[k] * pq[j]);
    for (int k = 0; k < 2; ++k) /* - r s p */
      for (int k2 = 0; k2 < 3; ++k2)
        for (int j = 0; j < 2; ++j) rho[k + k2][j] = rho[k + k2][j] - ((r[k] * s[k2]) * p[j]);
    for (int k = 0; k < 2; ++k) /* + r^2 q */
      for (int k2 = 0; k2 < 2; ++k2)
        for (int j = 0; j < 3; ++j) rho[k + k2][j] = rho[k + k2][j] + ((r[k] * r[k2]) * q[j]);
  }
  /* E_13 as a quadratic in l_3: l_3^2 + u l_3 + v,  u = -2 c13 l_1 - 2 b13,  v = l_1^2 + 2 a13 l_1 + k13 */
  const double u[2] = {-(2.0 * e13[2]), -(2.0 * e13[0])};
  const double v[3] = {e13[3], 2.0 * e13[1], 1.0};
  /* powers of l_3 modulo E_13:  l_3^2 = -u l_3 - v,  l_3^3 = (u^2 - v) l_3 + u v,  l_3^4 = (2 u v - u^3) l_3 + (v^2 - u^2 v) */
  double uu[3] = {0, 0, 0}, uv[4] = {0, 0, 0, 0}, uuu[4] = {0, 0, 0, 0}, vv[5] = {0, 0, 0, 0, 0}, uuv[5] = {0, 0, 0, 0, 0};
  sv_pmul_acc(u, 1, u, 1, 1.0, uu);
  sv_pmul_acc(u, 1, v, 2, 1.0, uv);
  sv_pmul_acc(uu, 2, u, 1, 1.0, uuu);
  sv_pmul_acc(v, 2, v, 2, 1.0, vv);
  sv_pmul_acc(uu, 2, v, 2, 1.0, uuv);
  double m3a[4] = {uu[0] - v[0], uu[1] - v[1], uu[2] - v[2], 0.0}; /* u^2 - v (degree 2) */
  double m4a[4], m4b[5];
  for (int j = 0; j < 4; ++j) m4a[j] = (2.0 * uv[j]) - uuu[j];
  for (int j = 0; j < 5; ++j) m4b[j] = vv[j] - uuv[j];
  double A[9], B[9]; /* generous sizes; the true degrees are 3 and 4 */
  for (int j = 0; j < 9; ++j) {
    A[j] = 0.0;
    B[j] = 0.0;
  }
  for (int j = 0; j < 5; ++j) { /* l_3^1 and l_3^0 terms as they are */
    A[j] = A[j] + rho[1][j];
    B[j] = B[j] + rho[0][j];
  }
  sv_pmul_acc(rho[2], 4, u, 1, -1.0, A);  /* rho_2 l_3^2 */
  sv_pmul_acc(rho[2], 4, v, 2, -1.0, B);
  sv_pmul_acc(rho[3], 4, m3a, 2, 1.0, A); /* rho_3 l_3^3 */
  sv_pmul_acc(rho[3], 4, uv, 3, 1.0, B);
  sv_pmul_acc(rho[4], 4, m4a, 3, 1.0, A); /* rho_4 l_3^4 */
  sv_pmul_acc(rho[4], 4, m4b, 4, 1.0, B);
  /* (the cancellations of the elimination leave A of degree 3 and B of degree 4; higher entries hold rounding noise only
   * and are dropped so that the octic stays an octic) */
  for (int j = 0; j < 4; ++j) Apoly[j] = A[j];
  for (int j = 0; j < 5; ++j) Bpoly[j] = B[j];
  for (int j = 0; j < 9; ++j) oct[j] = 0.0;
  double AB[8] = {0, 0, 0, 0, 0, 0, 0, 0}, AA[7] = {0, 0, 0, 0, 0, 0, 0};
  sv_pmul_acc(Bpoly, 4, Bpoly, 4, 1.0, oct);
  sv_pmul_acc(Apoly, 3, Bpoly, 4, 1.0, AB);
  sv_pmul_acc(AB, 7, u, 1, -1.0, oct);
  sv_pmul_acc(Apoly, 3, Apoly, 3, 1.0, AA);
  sv_pmul_acc(AA, 6, v, 2, 1.0, oct);
}

__device__ static double sv_peval(const double* c, int deg, double x) {
  double y = c[deg];
  for (int j = deg - 1; j >= 0; --j) y = (y * x) + c[j];
  return y;
}

/* Generalised P3P.  fb[9]: three unit bearings in the BODY frame, o[9]: their camera offsets (body frame), P[9]: the
 * three world points.  -> up to 8 body poses (R [9] row-major, t [3]) with x_body = R^T (P - t), i.e. P = R x + t.
 * Returns the number of solutions. */
__device__ static int sv_gp3p(const double* fb, const double* o, const double* P, double* R_out, double* t_out) {
  /* lengths in units of the largest side of the world triangle: depths and coefficients of order one */
  double L = 0.0;
  for (int i = 0; i < 3; ++i) {
    const int j = (i + 1) % 3;
    const double d0 = P[3 * i] - P[3 * j], d1 = P[3 * i + 1] - P[3 * j + 1], d2 = P[3 * i + 2] - P[3 * j + 2];
    const double d = sqrt(((d0 * d0) + (d1 * d1)) + (d2 * d2));
    if (d > L) L = d;
  }
  if (!(L > 0.0)) return 0;
  double os[9], Ps[9];
  for (int k = 0; k < 9; ++k) {
    os[k] = o[k] / L;
    Ps[k] = P[k] / L;
  }
  double e12[4], e13[4], e23[4];
  sv_gp3p_pair(fb, fb + 3, os, os + 3, Ps, Ps + 3, e12);
  sv_gp3p_pair(fb, fb + 6, os, os + 6, Ps, Ps + 6, e13);
  sv_gp3p_pair(fb + 3, fb + 6, os + 3, os + 6, Ps + 3, Ps + 6, e23);
  double oct[9], Ap[4], Bp[5];
  sv_gp3p_octic(e12, e13, e23, oct, Ap, Bp);
  double cmax = 0.0;
  for (int j = 0; j < 9; ++j)
    if (fabs(oct[j]) > cmax) cmax = fabs(oct[j]);
  if (!(cmax > 0.0) || !(fabs(oct[8]) > (1e-13 * cmax))) return 0; /* degenerate configuration */
  sv_cplx roots[8];
  sv_poly_roots(oct, 8, roots);
  int ns = 0;
  for (int k = 0; k < 8 && ns < SV_GP3P_MAXSOL; ++k) {
    if (!(fabs(roots[k].im) <= (1e-6 * (1.0 + fabs(roots[k].re))))) continue; /* complex root */
    double l1 = roots[k].re;
    const double Av = sv_peval(Ap, 3, l1), Bv = sv_peval(Bp, 4, l1);
    if (!(fabs(Av) > 0.0)) continue;
    double l3 = -(Bv / Av);
    /* l_2: the common root of l_2^2 + p l_2 + q (E_12) and l_2^2 + r l_2 + s (E_23): (p - r) l_2 + (q - s) = 0 */
    const double pv = -(((2.0 * e12[0]) * l1) + (2.0 * e12[2])), qv = (((l1 * l1) + ((2.0 * e12[1]) * l1)) + e12[3]);
    const double rv = (2.0 * e23[1]) - ((2.0 * e23[0]) * l3), sv = (((l3 * l3) - ((2.0 * e23[2]) * l3)) + e23[3]);
    if (!(fabs(pv - rv) > 0.0)) continue;
    double l2 = (sv - qv) / (pv - rv);
    /* three Newton steps on (E_12, E_13, E_23)(l_1, l_2, l_3) */
    for (int itn = 0; itn < 3; ++itn) {
      const double F0 = sv_gp3p_eval(e12, l1, l2), F1 = sv_gp3p_eval(e13, l1, l3), F2 = sv_gp3p_eval(e23, l2, l3);
      const double j00 = ((2.0 * l1) - ((2.0 * e12[0]) * l2)) + (2.0 * e12[1]);
      const double j01 = ((2.0 * l2) - ((2.0 * e12[0]) * l1)) - (2.0 * e12[2]);
      const double j10 = ((2.0 * l1) - ((2.0 * e13[0]) * l3)) + (2.0 * e13[1]);
      const double j12 = ((2.0 * l3) - ((2.0 * e13[0]) * l1)) - (2.0 * e13[2]);
      const double j21 = ((2.0 * l2) - ((2.0 * e23[0]) * l3)) + (2.0 * e23[1]);
      const double j22 = ((2.0 * l3) - ((2.0 * e23[0]) * l2)) - (2.0 * e23[2]);
      /* J = [j00 j01 0; j10 0 j12; 0 j21 j22]; det = -j00 j12 j21 - j01 j10 j22 */
      const double det = -((j00 * j12) * j21) - ((j01 * j10) * j22);
      if (!(fabs(det) > 0.0)) break;
      /* Cramer */
      const double d1 = (-((F0 * j12) * j21)) - (j01 * ((F1 * j22) - (j12 * F2)));
      const double d2 = (j00 * ((F1 * j22) - (j12 * F2))) - ((F0 * j10) * j22);
      const double d3 = (j00 * (-(F1 * j21))) - (j01 * ((j10 * F2)) ) + ((F0 * j10) * j21);
      l1 = l1 - (d1 / det);
      l2 = l2 - (d2 / det);
      l3 = l3 - (d3 / det);
    }
    if (!(l1 > 0.0) || !(l2 > 0.0) || !(l3 > 0.0)) continue; /* points behind their cameras */
    const double res = (fabs(sv_gp3p_eval(e12, l1, l2)) + fabs(sv_gp3p_eval(e13, l1, l3))) + fabs(sv_gp3p_eval(e23, l2, l3));
    if (!(res <= 1e-9)) continue; /* not a solution of the three quadrics (spurious real part of a complex pair) */
    /* body-frame points, then the rotation that takes the body triangle onto the world triangle */
    double X[9];
    const double ls[3] = {l1, l2, l3};
    for (int i = 0; i < 3; ++i)
      for (int c = 0; c < 3; ++c) X[3 * i + c] = os[3 * i + c] + (ls[i] * fb[3 * i + c]);
    double Eb[9], Ew[9];
    int okf = 1;
    for (int w = 0; w < 2; ++w) {
      const double* Q = w ? Ps : X;
      double* E = w ? Ew : Eb;
      double a0 = Q[3] - Q[0], a1 = Q[4] - Q[1], a2 = Q[5] - Q[2];
      const double b0 = Q[6] - Q[0], b1 = Q[7] - Q[1], b2 = Q[8] - Q[2];
      const double na = sqrt(((a0 * a0) + (a1 * a1)) + (a2 * a2));
      a0 = a0 / na;
      a1 = a1 / na;
      a2 = a2 / na;
      double n0 = (a1 * b2) - (a2 * b1), n1 = (a2 * b0) - (a0 * b2), n2 = (a0 * b1) - (a1 * b0);
      const double nn = sqrt(((n0 * n0) + (n1 * n1)) + (n2 * n2));
      if (!(na > 0.0) || !(nn > 0.0)) okf = 0;
      n0 = n0 / nn;
      n1 = n1 / nn;
      n2 = n2 / nn;
      E[0] = a0;
      E[1] = a1;
      E[2] = a2;
      E[3] = (n1 * a2) - (n2 * a1);
      E[4] = (n2 * a0) - (n0 * a2);
      E[5] = (n0 * a1) - (n1 * a0);
      E[6] = n0;
      E[7] = n1;
      E[8] = n2;
    }
    if (!okf) continue;
    double* R = R_out + 9 * ns;
    double* t = t_out + 3 * ns;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R[3 * i + j] = ((Ew[i] * Eb[j]) + (Ew[3 + i] * Eb[3 + j])) + (Ew[6 + i] * Eb[6 + j]);
    for (int i = 0; i < 3; ++i)
      t[i] = (Ps[i] - (((R[3 * i] * X[0]) + (R[3 * i + 1] * X[1])) + (R[3 * i + 2] * X[2]))) * L;
    int fin = 1;
    for (int i = 0; i < 9; ++i) fin &= (R[i] == R[i]) ? 1 : 0;
    for (int i = 0; i < 3; ++i) fin &= (t[i] == t[i]) ? 1 : 0;
    if (fin) ns++;
  }
  return ns;
}

/* One RANSAC hypothesis of the non-central problem with GP3P: four distinct correspondences out of all n (any camera),
 * GP3P on the first three, the pose with the smallest score on the fourth (first wins ties). */
__device__ static int sv_hypothesis_gp3p(const double* f, const double* p, const int32_t* cam, const double* cam_off,
                                      const double* cam_rot, int32_t n, uint64_t seed, uint64_t it, double* R_best,
                                      double* t_best) {
  int32_t s[4];
  if (!sv_sample_distinct(n, 4, seed, it, s)) return 0;
  double fb[9], o[9], P[9];
  for (int i = 0; i < 3; ++i) {
    const int32_t c = cam ? cam[s[i]] : 0;
    const double* Rc = cam_rot + 9 * c;
    const double* fi = f + 3 * s[i];
    for (int r = 0; r < 3; ++r) {
      fb[3 * i + r] = ((Rc[3 * r] * fi[0]) + (Rc[3 * r + 1] * fi[1])) + (Rc[3 * r + 2] * fi[2]);
      o[3 * i + r] = cam_off[3 * c + r];
      P[3 * i + r] = p[3 * s[i] + r];
    }
  }
  double Rs[9 * SV_GP3P_MAXSOL], ts[3 * SV_GP3P_MAXSOL];
  const int ns = sv_gp3p(fb, o, P, Rs, ts);
  const int32_t c3 = cam ? cam[s[3]] : 0;
  double best = 0.0;
  int found = 0;
  for (int k = 0; k < ns; ++k) {
    const double sc = sv_score(Rs + 9 * k, ts + 3 * k, f + 3 * s[3], p + 3 * s[3], cam_off + 3 * c3, cam_rot + 9 * c3);
    if (!(sc == sc)) continue; /* NaN */
    if (!found || sc < best) {
      found = 1;
      best = sc;
      for (int i = 0; i < 9; ++i) R_best[i] = Rs[9 * k + i];
      for (int i = 0; i < 3; ++i) t_best[i] = ts[3 * k + i];
    }
  }
  return found;
}

/* One RANSAC hypothesis of the central problem with TWOPT (known rotation = identity, the binding's prior): two distinct
 * correspondences, the translation from sv_twopt. */
__device__ static int sv_hypothesis_twopt(const double* f, const double* p, int32_t n, uint64_t seed, uint64_t it, double* R_out,
                                       double* t_out) {
  int32_t s[2];
  if (!sv_sample_distinct(n, 2, seed, it, s)) return 0;
  for (int i = 0; i < 9; ++i) R_out[i] = (i % 4 == 0) ? 1.0 : 0.0;
  return sv_twopt(f + 3 * s[0], f + 3 * s[1], p + 3 * s[0], p + 3 * s[1], R_out, t_out);
}
