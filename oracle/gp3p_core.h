/* TEST INFRASTRUCTURE ONLY -- CPU oracle (see ransac.c for the provenance header).
 *
 * Generalised P3P: the minimal solver of the NON-CENTRAL absolute-pose RANSAC.  The reference calls
 * pyopengv.absolute_pose_noncentral_ransac, which "will ALWAYS use GP3P" (omnistereo/pose_est_tools.py:696, :785):
 * OpenGV's AbsolutePoseSacProblem draws 4 correspondences from ALL cameras of the rig, solves the generalised
 * three-point problem on the first three (up to 8 poses) and keeps the pose that reprojects the fourth one best.
 * OpenGV's gp3p is machine-generated Groebner-basis code (Kneip, Furgale, Siegwart, ICRA 2013) that is not in the
 * reference tree; this is a restatement of the PROBLEM it solves, with an own elimination:
 *   rays x_i = o_i + l_i f_i in the body frame (o_i camera offset, f_i unit bearing rotated into the body frame), world
 *   points P_i, i = 1..3.  Rigid motion keeps distances: |x_i - x_j|^2 = |P_i - P_j|^2 gives three quadrics
 *     E_ij: l_i^2 + l_j^2 - 2 c_ij l_i l_j + 2 a_ij l_i - 2 b_ij l_j + k_ij = 0,
 *     c_ij = f_i.f_j, a_ij = f_i.(o_i - o_j), b_ij = f_j.(o_i - o_j), k_ij = |o_i - o_j|^2 - |P_i - P_j|^2,
 *   each in two unknowns only.  E_12 and E_23 are monic quadratics in l_2: their resultant R(l_1, l_3) is quartic in
 *   l_3; modulo E_13 (monic quadratic in l_3) it becomes A(l_1) l_3 + B(l_1), and substituting l_3 = -B / A into E_13
 *   leaves the OCTIC B^2 - A B u + A^2 v in l_1 (Nister & Stewenius 2007 reach an octic for the same problem): up to 8
 *   solutions, as Bezout's bound for three quadrics says.  Roots by Laguerre's method with deflation (all eight,
 *   complex arithmetic), real ones kept; l_3 = -B / A, l_2 from the common root of E_12 and E_23; three Newton steps
 *   on (E_12, E_13, E_23) remove what the elimination lost in accuracy; solutions with a non-positive depth or a
 *   residual are dropped; the pose follows from aligning the two triangles.
 * Only + - * / sqrt, fabs and comparisons, fully parenthesised, no FMA contraction: the HIP side is the TEXT of this
 * file (tests/gen_device_headers.py writes vo_single_camera_sos_amd/csrc/gp3p_core.h from it), so both sides produce
 * the same bits.  Independent checks: tests/test_oracle_gp3p.py (numpy.roots on the octic, residuals, planted poses).
 */
#pragma once
#include "epnp_core.h"

#define ORC_GP3P_MAXSOL 8

typedef struct {
  double re, im;
} orc_cplx;

static inline orc_cplx orc_c(double re, double im) {
  orc_cplx z;
  z.re = re;
  z.im = im;
  return z;
}
static inline orc_cplx orc_cadd(orc_cplx a, orc_cplx b) { return orc_c(a.re + b.re, a.im + b.im); }
static inline orc_cplx orc_csub(orc_cplx a, orc_cplx b) { return orc_c(a.re - b.re, a.im - b.im); }
static inline orc_cplx orc_cmul(orc_cplx a, orc_cplx b) {
  return orc_c((a.re * b.re) - (a.im * b.im), (a.re * b.im) + (a.im * b.re));
}
static inline orc_cplx orc_cscale(orc_cplx a, double s) { return orc_c(a.re * s, a.im * s); }
static inline double orc_cabs(orc_cplx a) { return sqrt((a.re * a.re) + (a.im * a.im)); }
static inline orc_cplx orc_cdiv(orc_cplx a, orc_cplx b) { /* Smith's formula */
  if (fabs(b.re) >= fabs(b.im)) {
    const double r = b.im / b.re, den = b.re + (r * b.im);
    return orc_c((a.re + (r * a.im)) / den, (a.im - (r * a.re)) / den);
  }
  const double r = b.re / b.im, den = b.im + (r * b.re);
  return orc_c(((a.re * r) + a.im) / den, ((a.im * r) - a.re) / den);
}
static inline orc_cplx orc_csqrt(orc_cplx z) { /* principal square root from real square roots */
  if (z.re == 0.0 && z.im == 0.0) return orc_c(0.0, 0.0);
  const double x = fabs(z.re), y = fabs(z.im);
  double w;
  if (x >= y) {
    const double r = y / x;
    w = sqrt(x) * sqrt(0.5 * (1.0 + sqrt(1.0 + (r * r))));
  } else {
    const double r = x / y;
    w = sqrt(y) * sqrt(0.5 * (r + sqrt(1.0 + (r * r))));
  }
  if (z.re >= 0.0) return orc_c(w, z.im / (2.0 * w));
  const double im = z.im >= 0.0 ? w : -w;
  return orc_c(z.im / (2.0 * im), im);
}

/* Laguerre's iteration on a polynomial of degree m with REAL coefficients a[0..m] (a[m] leading), from *x; the classic
 * formulation with a fractional step every tenth iteration to break limit cycles.  Magnitudes that only feed the
 * rounding-error bound of the evaluation (Adams' running bound) or a comparison are taken in the 1-norm / squared:
 * a square root per coefficient and three per iteration were more than half of the instructions of an iteration. */
static inline double orc_cabs1(orc_cplx a) { return fabs(a.re) + fabs(a.im); }
static inline void orc_laguerre(const double* a, int m, orc_cplx* x) {
  const double frac[9] = {0.0, 0.5, 0.25, 0.75, 0.13, 0.38, 0.62, 0.88, 1.0};
  for (int iter = 1; iter <= 80; ++iter) {
    orc_cplx b = orc_c(a[m], 0.0), d = orc_c(0.0, 0.0), f = orc_c(0.0, 0.0);
    double err = orc_cabs1(b);
    const double abx = orc_cabs1(*x);
    for (int j = m - 1; j >= 0; --j) {
      f = orc_cadd(orc_cmul(*x, f), d);
      d = orc_cadd(orc_cmul(*x, d), b);
      b = orc_cmul(*x, b);
      b.re = b.re + a[j];
      err = orc_cabs1(b) + (abx * err);
    }
    err = err * 1e-15;
    if (orc_cabs1(b) <= err) return; /* on a root */
    const orc_cplx g = orc_cdiv(d, b), g2 = orc_cmul(g, g);
    const orc_cplx h = orc_csub(g2, orc_cscale(orc_cdiv(f, b), 2.0));
    const orc_cplx sq = orc_csqrt(orc_cscale(orc_csub(orc_cscale(h, (double)m), g2), (double)(m - 1)));
    orc_cplx gp = orc_cadd(g, sq);
    const orc_cplx gm = orc_csub(g, sq);
    double abp = (gp.re * gp.re) + (gp.im * gp.im);
    const double abm = (gm.re * gm.re) + (gm.im * gm.im);
    if (abp < abm) {
      gp = gm;
      abp = abm;
    }
    const orc_cplx dx = abp > 0.0 ? orc_cdiv(orc_c((double)m, 0.0), gp) : orc_c((1.0 + abx) * 0.6, (1.0 + abx) * 0.8);
    const orc_cplx x1 = orc_csub(*x, dx);
    if (x->re == x1.re && x->im == x1.im) return; /* converged */
    if (iter % 10 != 0) *x = x1;
    else *x = orc_csub(*x, orc_cscale(dx, frac[iter / 10]));
  }
}

/* All roots of the real polynomial c[0] + c[1] x + ... + c[m] x^m (m <= 10, c[m] != 0): Laguerre from 0 on the
 * deflated polynomial, which stays REAL -- a real root is divided out as (x - r), a complex one together with its
 * conjugate as the real quadratic x^2 - 2 Re(z) x + |z|^2 (one iteration run finds both: conjugate roots of a real
 * polynomial need not be searched twice); once four roots are left, the real ones among them come from the closed-form
 * quartic solver.  Then the roots that can still turn out real (|Im| <= 1e-3 (1 + |Re|): every
 * caller discards the others) are polished on the undeflated polynomial. */
static inline void orc_poly_roots(const double* c, int m, orc_cplx* roots) {
  double ad[11];
  for (int j = 0; j <= m; ++j) ad[j] = c[j];
  int deg = m, nr = 0;
  while (deg >= 1) {
    if (deg == 4) { /* the last four in closed form -- every caller wants the REAL roots only (orc_quartic: Ferrari) */
      const double q4[5] = {ad[4], ad[3], ad[2], ad[1], ad[0]};
      double rr[4];
      const int k4 = orc_quartic(q4, rr);
      for (int k = 0; k < 4; ++k)
        if (k < k4) roots[nr++] = orc_c(rr[k], 0.0);
      while (nr < m) roots[nr++] = orc_c(0.0, 1.0); /* stand-ins for the complex ones: skipped by every caller */
      break;
    }
    orc_cplx x = orc_c(0.0, 0.0);
    orc_laguerre(ad, deg, &x);
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
      roots[nr++] = orc_c(x.re, -x.im);
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
    if (fabs(roots[j].im) <= (1e-3 * (1.0 + fabs(roots[j].re)))) orc_laguerre(c, m, &roots[j]);
}

/* small real-polynomial helpers: p (degree dp) times q (degree dq) accumulated into out with weight w */
static inline void orc_pmul_acc(const double* p, int dp, const double* q, int dq, double w, double* out) {
  for (int i = 0; i <= dp; ++i)
    for (int j = 0; j <= dq; ++j) out[i + j] = out[i + j] + (w * (p[i] * q[j]));
}

/* The coefficients of one quadric E_ij (see the header): {c, a, b, k}. */
static inline void orc_gp3p_pair(const double* fi, const double* fj, const double* oi, const double* oj, const double* Pi,
                                 const double* Pj, double* e) {
  const double d0 = oi[0] - oj[0], d1 = oi[1] - oj[1], d2 = oi[2] - oj[2];
  const double q0 = Pi[0] - Pj[0], q1 = Pi[1] - Pj[1], q2 = Pi[2] - Pj[2];
  e[0] = ((fi[0] * fj[0]) + (fi[1] * fj[1])) + (fi[2] * fj[2]);
  e[1] = ((fi[0] * d0) + (fi[1] * d1)) + (fi[2] * d2);
  e[2] = ((fj[0] * d0) + (fj[1] * d1)) + (fj[2] * d2);
  e[3] = (((d0 * d0) + (d1 * d1)) + (d2 * d2)) - (((q0 * q0) + (q1 * q1)) + (q2 * q2));
}
static inline double orc_gp3p_eval(const double* e, double li, double lj) {
  return (((((li * li) + (lj * lj)) - (((2.0 * e[0]) * li) * lj)) + ((2.0 * e[1]) * li)) - ((2.0 * e[2]) * lj)) + e[3];
}

/* The octic in l_1 (oct[0..8], oct[8] leading) and the polynomials A (degree <= 3), B (degree <= 4) of the linear
 * relation A l_3 + B = 0.  e12, e13, e23: quadric coefficients {c, a, b, k}. */
static inline void orc_gp3p_octic(const double* e12, const double* e13, const double* e23, double* oct, double* Apoly,
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
    orc_pmul_acc(q, 2, q, 2, 1.0, qq);
    for (int j = 0; j < 5; ++j) rho[0][j] = rho[0][j] + qq[j];
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 3; ++j) rho[k][j] = rho[k][j] - ((2.0 * s[k]) * q[j]);
    for (int k = 0; k < 3; ++k)
      for (int k2 = 0; k2 < 3; ++k2) rho[k + k2][0] = rho[k + k2][0] + (s[k] * s[k2]);
  }
  /* (p - r)(p s - r q) = p^2 s - p r q - r p s + r^2 q */
  {
    double pq[4] = {0, 0, 0, 0}, pp[3] = {0, 0, 0};
    orc_pmul_acc(p, 1, q, 2, 1.0, pq);
    orc_pmul_acc(p, 1, p, 1, 1.0, pp);
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
  orc_pmul_acc(u, 1, u, 1, 1.0, uu);
  orc_pmul_acc(u, 1, v, 2, 1.0, uv);
  orc_pmul_acc(uu, 2, u, 1, 1.0, uuu);
  orc_pmul_acc(v, 2, v, 2, 1.0, vv);
  orc_pmul_acc(uu, 2, v, 2, 1.0, uuv);
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
  orc_pmul_acc(rho[2], 4, u, 1, -1.0, A);  /* rho_2 l_3^2 */
  orc_pmul_acc(rho[2], 4, v, 2, -1.0, B);
  orc_pmul_acc(rho[3], 4, m3a, 2, 1.0, A); /* rho_3 l_3^3 */
  orc_pmul_acc(rho[3], 4, uv, 3, 1.0, B);
  orc_pmul_acc(rho[4], 4, m4a, 3, 1.0, A); /* rho_4 l_3^4 */
  orc_pmul_acc(rho[4], 4, m4b, 4, 1.0, B);
  /* (the cancellations of the elimination leave A of degree 3 and B of degree 4; higher entries hold rounding noise only
   * and are dropped so that the octic stays an octic) */
  for (int j = 0; j < 4; ++j) Apoly[j] = A[j];
  for (int j = 0; j < 5; ++j) Bpoly[j] = B[j];
  for (int j = 0; j < 9; ++j) oct[j] = 0.0;
  double AB[8] = {0, 0, 0, 0, 0, 0, 0, 0}, AA[7] = {0, 0, 0, 0, 0, 0, 0};
  orc_pmul_acc(Bpoly, 4, Bpoly, 4, 1.0, oct);
  orc_pmul_acc(Apoly, 3, Bpoly, 4, 1.0, AB);
  orc_pmul_acc(AB, 7, u, 1, -1.0, oct);
  orc_pmul_acc(Apoly, 3, Apoly, 3, 1.0, AA);
  orc_pmul_acc(AA, 6, v, 2, 1.0, oct);
}

static inline double orc_peval(const double* c, int deg, double x) {
  double y = c[deg];
  for (int j = deg - 1; j >= 0; --j) y = (y * x) + c[j];
  return y;
}

/* Generalised P3P.  fb[9]: three unit bearings in the BODY frame, o[9]: their camera offsets (body frame), P[9]: the
 * three world points.  -> up to 8 body poses (R [9] row-major, t [3]) with x_body = R^T (P - t), i.e. P = R x + t.
 * Returns the number of solutions. */
static inline int orc_gp3p(const double* fb, const double* o, const double* P, double* R_out, double* t_out) {
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
  orc_gp3p_pair(fb, fb + 3, os, os + 3, Ps, Ps + 3, e12);
  orc_gp3p_pair(fb, fb + 6, os, os + 6, Ps, Ps + 6, e13);
  orc_gp3p_pair(fb + 3, fb + 6, os + 3, os + 6, Ps + 3, Ps + 6, e23);
  double oct[9], Ap[4], Bp[5];
  orc_gp3p_octic(e12, e13, e23, oct, Ap, Bp);
  double cmax = 0.0;
  for (int j = 0; j < 9; ++j)
    if (fabs(oct[j]) > cmax) cmax = fabs(oct[j]);
  if (!(cmax > 0.0) || !(fabs(oct[8]) > (1e-13 * cmax))) return 0; /* degenerate configuration */
  orc_cplx roots[8];
  orc_poly_roots(oct, 8, roots);
  int ns = 0;
  for (int k = 0; k < 8 && ns < ORC_GP3P_MAXSOL; ++k) {
    if (!(fabs(roots[k].im) <= (1e-6 * (1.0 + fabs(roots[k].re))))) continue; /* complex root */
    double l1 = roots[k].re;
    const double Av = orc_peval(Ap, 3, l1), Bv = orc_peval(Bp, 4, l1);
    if (!(fabs(Av) > 0.0)) continue;
    double l3 = -(Bv / Av);
    /* l_2: the common root of l_2^2 + p l_2 + q (E_12) and l_2^2 + r l_2 + s (E_23): (p - r) l_2 + (q - s) = 0 */
    const double pv = -(((2.0 * e12[0]) * l1) + (2.0 * e12[2])), qv = (((l1 * l1) + ((2.0 * e12[1]) * l1)) + e12[3]);
    const double rv = (2.0 * e23[1]) - ((2.0 * e23[0]) * l3), sv = (((l3 * l3) - ((2.0 * e23[2]) * l3)) + e23[3]);
    if (!(fabs(pv - rv) > 0.0)) continue;
    double l2 = (sv - qv) / (pv - rv);
    /* three Newton steps on (E_12, E_13, E_23)(l_1, l_2, l_3) */
    for (int itn = 0; itn < 3; ++itn) {
      const double F0 = orc_gp3p_eval(e12, l1, l2), F1 = orc_gp3p_eval(e13, l1, l3), F2 = orc_gp3p_eval(e23, l2, l3);
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
    const double res = (fabs(orc_gp3p_eval(e12, l1, l2)) + fabs(orc_gp3p_eval(e13, l1, l3))) + fabs(orc_gp3p_eval(e23, l2, l3));
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
static inline int orc_hypothesis_gp3p(const double* f, const double* p, const int32_t* cam, const double* cam_off,
                                      const double* cam_rot, int32_t n, uint64_t seed, uint64_t it, double* R_best,
                                      double* t_best) {
  int32_t s[4];
  if (!orc_sample_distinct(n, 4, seed, it, s)) return 0;
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
  double Rs[9 * ORC_GP3P_MAXSOL], ts[3 * ORC_GP3P_MAXSOL];
  const int ns = orc_gp3p(fb, o, P, Rs, ts);
  const int32_t c3 = cam ? cam[s[3]] : 0;
  double best = 0.0;
  int found = 0;
  for (int k = 0; k < ns; ++k) {
    const double sc = orc_score(Rs + 9 * k, ts + 3 * k, f + 3 * s[3], p + 3 * s[3], cam_off + 3 * c3, cam_rot + 9 * c3);
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
 * correspondences, the translation from orc_twopt. */
static inline int orc_hypothesis_twopt(const double* f, const double* p, int32_t n, uint64_t seed, uint64_t it, double* R_out,
                                       double* t_out) {
  int32_t s[2];
  if (!orc_sample_distinct(n, 2, seed, it, s)) return 0;
  for (int i = 0; i < 9; ++i) R_out[i] = (i % 4 == 0) ? 1.0 : 0.0;
  return orc_twopt(f + 3 * s[0], f + 3 * s[1], p + 3 * s[0], p + 3 * s[1], R_out, t_out);
}
