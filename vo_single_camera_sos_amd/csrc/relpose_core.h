// Device-side numeric core of the 2D-2D relative-pose RANSAC (pyopengv.relative_pose_ransac, reference call site
// omnistereo/pose_est_tools.py:78): the reference's own score of a correspondence under a relative pose
// (pose_est_tools.py:150-203), the decomposition of an essential matrix, the minimal solvers, one hypothesis per lane.
// GENERATED from oracle/relpose_core.h by tests/gen_device_headers.py (same text, device prefixes): the CPU oracle evaluates the same
// operations in the same order, tests/test_abi.py checks that the two files stay identical.
#pragma once
#include <hip/hip_runtime.h>
#include "epnp_core.h"
#include "gp3p_core.h"

/* OpenGV's closed-form midpoint of the two rays (triangulation::triangulate2): the point in frame 1 and the two ray
 * parameters (depth along f1, depth along R f2). */
__device__ static void sv_rel_triangulate(const double* R, const double* t, const double* f1, const double* f2, double* X,
                                       double* l0_out, double* l1_out) {
  double g[3];
  for (int i = 0; i < 3; ++i) g[i] = ((R[3 * i] * f2[0]) + (R[3 * i + 1] * f2[1])) + (R[3 * i + 2] * f2[2]);
  const double b0 = ((t[0] * f1[0]) + (t[1] * f1[1])) + (t[2] * f1[2]), b1 = ((t[0] * g[0]) + (t[1] * g[1])) + (t[2] * g[2]);
  const double a00 = ((f1[0] * f1[0]) + (f1[1] * f1[1])) + (f1[2] * f1[2]);
  const double a10 = ((f1[0] * g[0]) + (f1[1] * g[1])) + (f1[2] * g[2]);
  const double a01 = -a10, a11 = -(((g[0] * g[0]) + (g[1] * g[1])) + (g[2] * g[2]));
  const double det = (a00 * a11) - (a01 * a10);
  const double l0 = ((b0 * a11) - (a01 * b1)) / det, l1 = ((a00 * b1) - (a10 * b0)) / det;
  for (int k = 0; k < 3; ++k) X[k] = ((l0 * f1[k]) + (t[k] + (l1 * g[k]))) / 2.0;
  *l0_out = l0;
  *l1_out = l1;
}

/* pose_est_tools.py:150-203 (relative case): sum of the two reprojection errors 1 - cos(angle). */
__device__ static double sv_rel_score(const double* R, const double* t, const double* f1, const double* f2) {
  double X[3], l0, l1;
  sv_rel_triangulate(R, t, f1, f2, X, &l0, &l1);
  const double n1 = sqrt(((X[0] * X[0]) + (X[1] * X[1])) + (X[2] * X[2]));
  const double e1 = 1.0 - ((((f1[0] * X[0]) + (f1[1] * X[1])) + (f1[2] * X[2])) / n1);
  const double d0 = X[0] - t[0], d1 = X[1] - t[1], d2 = X[2] - t[2];
  const double y0 = ((R[0] * d0) + (R[3] * d1)) + (R[6] * d2);
  const double y1 = ((R[1] * d0) + (R[4] * d1)) + (R[7] * d2);
  const double y2 = ((R[2] * d0) + (R[5] * d1)) + (R[8] * d2);
  const double n2 = sqrt(((y0 * y0) + (y1 * y1)) + (y2 * y2));
  const double e2 = 1.0 - ((((f2[0] * y0) + (f2[1] * y1)) + (f2[2] * y2)) / n2);
  return e1 + e2;
}

/* the adaptive stop of sac::Ransac for k-point samples: 1 - w^k (k = the problem's sample size INCLUDING the points it
 * draws for disambiguation: 8 for the eight-point and the five-point solvers (5 + 3), 9 for the seven-point one (7 + 2)) */
__device__ static double sv_adaptive_base_k(int32_t best, int32_t n, int k) {
  const double w = (double)best / (double)n;
  double wk = 1.0;
  for (int j = 0; j < k; ++j) wk = wk * w;
  double pno = 1.0 - wk;
  const double eps = 2.220446049250313e-16;
  if (pno < eps) pno = eps;
  if (pno > 1.0 - eps) pno = 1.0 - eps;
  return pno;
}

#define SV_REL_FIVEPT 5
#define SV_REL_SEVENPT 7
#define SV_REL_EIGHTPT 8

/* The four (R, t) of an essential matrix E = [t]x R (up to scale and sign), |t| = 1: SVD of E through the Jacobi
 * eigen-decomposition of E^T E = V diag(s^2) V^T, U = E V / s (one Gram-Schmidt step), R = U W V^T or U W^T V^T with
 * W = [0 -1 0; 1 0 0; 0 0 1], t = +-u2 (Hartley & Zisserman 9.6.2).  Each candidate is rated on the m given
 * correspondences by the SUM of the reference's score (the triangulated point reprojected into both views: a point
 * behind a view costs ~2) -- the candidate with the smallest sum wins, as OpenGV's CentralRelativePoseSacProblem picks
 * among the decompositions of its essential matrices.  *best_q carries the smallest sum so far across calls (several
 * essential matrices of one sample); R_out / t_out are overwritten only by a strictly better candidate. */
__device__ static void sv_rel_decompose_pick(const double* E, const double* f1, const double* f2, int m, double* best_q,
                                          double* R_out, double* t_out) {
  double S[9], Vs[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) S[3 * r + c] = ((E[r] * E[c]) + (E[3 + r] * E[3 + c])) + (E[6 + r] * E[6 + c]);
  sv_jacobi_sym(S, 3, Vs);
  /* eigenvalues descending with their eigenvectors: a three-element sorting network on values, not indices */
  double e0 = S[0], e1 = S[4], e2 = S[8];
  double v0[3] = {Vs[0], Vs[3], Vs[6]}, v1[3] = {Vs[1], Vs[4], Vs[7]}, v2[3] = {Vs[2], Vs[5], Vs[8]};
#define SV_REL_CSWAP(ea, va, eb, vb)              \
  if (ea < eb) {                                   \
    double t_ = ea; ea = eb; eb = t_;              \
    t_ = va[0]; va[0] = vb[0]; vb[0] = t_;         \
    t_ = va[1]; va[1] = vb[1]; vb[1] = t_;         \
    t_ = va[2]; va[2] = vb[2]; vb[2] = t_;         \
  }
  SV_REL_CSWAP(e0, v0, e1, v1)
  SV_REL_CSWAP(e1, v1, e2, v2)
  SV_REL_CSWAP(e0, v0, e1, v1)
#undef SV_REL_CSWAP
  if (!(e1 > 0.0)) return; /* rank < 2 */
  double u0[3], u1[3], u2[3];
  v2[0] = (v0[1] * v1[2]) - (v0[2] * v1[1]); /* right-handed V */
  v2[1] = (v0[2] * v1[0]) - (v0[0] * v1[2]);
  v2[2] = (v0[0] * v1[1]) - (v0[1] * v1[0]);
  const double s0 = sqrt(e0), s1 = sqrt(e1);
  for (int r = 0; r < 3; ++r) {
    u0[r] = (((E[3 * r] * v0[0]) + (E[3 * r + 1] * v0[1])) + (E[3 * r + 2] * v0[2])) / s0;
    u1[r] = (((E[3 * r] * v1[0]) + (E[3 * r + 1] * v1[1])) + (E[3 * r + 2] * v1[2])) / s1;
  }
  { /* u1 orthogonal to u0 up to rounding: one Gram-Schmidt step */
    const double d = ((u0[0] * u1[0]) + (u0[1] * u1[1])) + (u0[2] * u1[2]);
    for (int r = 0; r < 3; ++r) u1[r] = u1[r] - (d * u0[r]);
    const double nn = sqrt(((u1[0] * u1[0]) + (u1[1] * u1[1])) + (u1[2] * u1[2]));
    const double n0 = sqrt(((u0[0] * u0[0]) + (u0[1] * u0[1])) + (u0[2] * u0[2]));
    if (!(nn > 0.0) || !(n0 > 0.0)) return;
    for (int r = 0; r < 3; ++r) {
      u1[r] = u1[r] / nn;
      u0[r] = u0[r] / n0;
    }
  }
  u2[0] = (u0[1] * u1[2]) - (u0[2] * u1[1]); /* right-handed U: E ~ U diag(1, 1, 0) V^T up to sign */
  u2[1] = (u0[2] * u1[0]) - (u0[0] * u1[2]);
  u2[2] = (u0[0] * u1[1]) - (u0[1] * u1[0]);
  for (int cand = 0; cand < 4; ++cand) {
    double Rc[9], tc[3];
    const double sg = (cand & 1) ? -1.0 : 1.0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        /* U W V^T: columns of U W = (u1, -u0, u2); U W^T: (-u1, u0, u2) */
        const double a = (cand & 2) ? -u1[r] : u1[r], b = (cand & 2) ? u0[r] : -u0[r];
        Rc[3 * r + c] = ((a * v0[c]) + (b * v1[c])) + (u2[r] * v2[c]);
      }
    for (int r = 0; r < 3; ++r) tc[r] = sg * u2[r];
    double q = 0.0;
    for (int i = 0; i < m; ++i) q = q + sv_rel_score(Rc, tc, f1 + 3 * i, f2 + 3 * i);
    if (q < *best_q) { /* NaN: never */
      *best_q = q;
      for (int k = 0; k < 9; ++k) R_out[k] = Rc[k];
      for (int k = 0; k < 3; ++k) t_out[k] = tc[k];
    }
  }
}

/* A^T A of the epipolar constraints f1^T E f2 = 0 of m correspondences (rows of A: f1_r f2_c at 3 r + c), its Jacobi
 * eigen-decomposition, and the k eigenvectors of the SMALLEST eigenvalues (ascending) as 3 x 3 matrices Eb[k][9]: an
 * orthonormal basis of the null space of A for m = 9 - k generic correspondences. */
__device__ static void sv_rel_null_basis(const double* f1, const double* f2, int m, int k, double* Eb) {
  double M[81], V[81];
  for (int q = 0; q < 81; ++q) M[q] = 0.0;
  for (int i = 0; i < m; ++i) {
    double a[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) a[3 * r + c] = f1[3 * i + r] * f2[3 * i + c];
    for (int r = 0; r < 9; ++r)
      for (int c = 0; c < 9; ++c) M[9 * r + c] = M[9 * r + c] + (a[r] * a[c]);
  }
  sv_jacobi_sym(M, 9, V);
  int used[9];
  for (int q = 0; q < 9; ++q) used[q] = 0;
  for (int s = 0; s < k; ++s) {
    int km = -1;
    for (int q = 0; q < 9; ++q)
      if (!used[q] && (km < 0 || M[10 * q] < M[10 * km])) km = q;
    used[km] = 1;
    for (int q = 0; q < 9; ++q) Eb[9 * s + q] = V[9 * q + km];
  }
}

/* Eight-point algorithm on 8 correspondences (f1[24], f2[24]) -> R, t (|t| = 1).  Returns 0 on a degenerate sample. */
__device__ static int sv_eightpt(const double* f1, const double* f2, double* R_out, double* t_out) {
  double E[9];
  sv_rel_null_basis(f1, f2, 8, 1, E);
  double best_q = 1e300;
  sv_rel_decompose_pick(E, f1, f2, 8, &best_q, R_out, t_out);
  return best_q < 1e300;
}

/* ---- seven-point algorithm (Hartley & Zisserman, alg. 11.1, on bearing vectors) -------------------------------------
 * The two-dimensional null space E2 + a (E1 - E2) of seven epipolar constraints; det = 0 is a cubic in a with one or
 * three real roots.  E_out[3][9]; returns the number of matrices. */
__device__ static int sv_sevenpt(const double* f1, const double* f2, double* E_out) {
  double Eb[18];
  sv_rel_null_basis(f1, f2, 7, 2, Eb);
  /* entries as polynomials of degree 1 in a: e[k] = {E2_k, E1_k - E2_k} */
  double e[9][2];
  for (int k = 0; k < 9; ++k) {
    e[k][0] = Eb[9 + k];
    e[k][1] = Eb[k] - Eb[9 + k];
  }
  double c[4] = {0.0, 0.0, 0.0, 0.0};
  const int cof[3][5] = {{0, 4, 8, 5, 7}, {1, 3, 8, 5, 6}, {2, 3, 7, 4, 6}}; /* e_a (e_b e_c - e_d e_e), signs + - + */
  for (int q = 0; q < 3; ++q) {
    double m2[3] = {0.0, 0.0, 0.0};
    sv_pmul_acc(e[cof[q][1]], 1, e[cof[q][2]], 1, 1.0, m2);
    sv_pmul_acc(e[cof[q][3]], 1, e[cof[q][4]], 1, -1.0, m2);
    sv_pmul_acc(e[cof[q][0]], 1, m2, 2, (q == 1) ? -1.0 : 1.0, c);
  }
  double cmax = 0.0;
  for (int j = 0; j < 4; ++j)
    if (fabs(c[j]) > cmax) cmax = fabs(c[j]);
  if (!(cmax > 0.0)) return 0;
  int deg = 3;
  while (deg > 0 && !(fabs(c[deg]) > (1e-13 * cmax))) deg--;
  if (deg == 0) return 0;
  sv_cplx roots[3];
  sv_poly_roots(c, deg, roots);
  int ns = 0;
  for (int k = 0; k < deg; ++k) {
    if (!(fabs(roots[k].im) <= (1e-6 * (1.0 + fabs(roots[k].re))))) continue;
    const double a = roots[k].re;
    for (int q = 0; q < 9; ++q) E_out[9 * ns + q] = e[q][0] + (a * e[q][1]);
    ns++;
  }
  return ns;
}

/* ---- five-point algorithm (Nister, "An efficient solution to the five-point relative pose problem", PAMI 2004) -------
 * serves the names "NISTER" and "STEWENIUS" (Stewenius, Engels, Nister 2006 solve the same ten cubics through a Groebner
 * basis / action matrix: the same solution set).  E = x X + y Y + z Z + W on the four-dimensional null space of the five
 * epipolar constraints; det E = 0 and 2 E E^T E - tr(E E^T) E = 0 are ten cubics in (x, y, z); Gauss-Jordan elimination
 * on the ten monomials of highest (x, y) order leaves rows from which three relations x p(z) + y q(z) + r(z) = 0 follow
 * (degrees 3, 3, 4); their 3 x 3 determinant is a polynomial of degree ten in z; each real root gives (x, y) from the
 * null vector of the 3 x 3 system, and an essential matrix.
 * Polynomials in (x, y, z) of total degree <= 3 are arrays over the 20 monomials in the paper's elimination order:
 *   0 x^3  1 y^3  2 x^2y  3 xy^2  4 x^2z  5 x^2  6 y^2z  7 y^2  8 xyz  9 xy | 10 xz^2  11 xz  12 x  13 yz^2  14 yz  15 y
 *   16 z^3  17 z^2  18 z  19 1.   A linear form is {x, y, z, 1}. */
__device__ static void sv_fp_lin_lin_acc(const double* p, const double* q, double w, double* out) {
  const int pr[16] = {5, 9, 11, 12, 9, 7, 14, 15, 11, 14, 17, 18, 12, 15, 18, 19};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out[pr[(4 * i) + j]] = out[pr[(4 * i) + j]] + (w * (p[i] * q[j]));
}

__device__ static void sv_fp_quad_lin_acc(const double* p, const double* q, double w, double* out) {
  const int qi[10] = {5, 7, 17, 9, 11, 14, 12, 15, 18, 19}; /* x^2 y^2 z^2 xy xz yz x y z 1 */
  const int pr[40] = {0, 2, 4, 5,    3, 1, 6, 7,     10, 13, 16, 17, 2, 3, 8, 9,     4, 8, 10, 11,
                      8, 6, 13, 14,  5, 9, 11, 12,   9, 7, 14, 15,   11, 14, 17, 18, 12, 15, 18, 19};
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 4; ++j) out[pr[(4 * i) + j]] = out[pr[(4 * i) + j]] + (w * (p[qi[i]] * q[j]));
}

/* f1, f2: five correspondences.  E_out[10][9]; returns the number of real solutions. */
__device__ static int sv_fivept(const double* f1, const double* f2, double* E_out) {
  double Eb[36];
  sv_rel_null_basis(f1, f2, 5, 4, Eb);
  double L[9][4]; /* entry k of E as a linear form {x, y, z, 1} */
  for (int k = 0; k < 9; ++k)
    for (int s = 0; s < 4; ++s) L[k][s] = Eb[(9 * s) + k];
  double A[10][20];
  for (int r = 0; r < 10; ++r)
    for (int c = 0; c < 20; ++c) A[r][c] = 0.0;
  { /* E E^T (symmetric, quadratic), Lambda = E E^T - tr/2 I, rows 0..8: Lambda E */
    double G[6][20]; /* (0,0) (0,1) (0,2) (1,1) (1,2) (2,2) */
    const int ga[6] = {0, 0, 0, 1, 1, 2}, gb[6] = {0, 1, 2, 1, 2, 2};
    for (int g = 0; g < 6; ++g) {
      for (int c = 0; c < 20; ++c) G[g][c] = 0.0;
      for (int k = 0; k < 3; ++k) sv_fp_lin_lin_acc(L[(3 * ga[g]) + k], L[(3 * gb[g]) + k], 1.0, G[g]);
    }
    double tr2[20];
    for (int c = 0; c < 20; ++c) tr2[c] = 0.5 * ((G[0][c] + G[3][c]) + G[5][c]);
    for (int c = 0; c < 20; ++c) {
      G[0][c] = G[0][c] - tr2[c];
      G[3][c] = G[3][c] - tr2[c];
      G[5][c] = G[5][c] - tr2[c];
    }
    const int sym[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c)
        for (int b = 0; b < 3; ++b) sv_fp_quad_lin_acc(G[sym[(3 * a) + b]], L[(3 * b) + c], 1.0, A[(3 * a) + c]);
  }
  { /* row 9: det E */
    const int cof[3][5] = {{0, 4, 8, 5, 7}, {1, 3, 8, 5, 6}, {2, 3, 7, 4, 6}};
    for (int q = 0; q < 3; ++q) {
      double m2[20];
      for (int c = 0; c < 20; ++c) m2[c] = 0.0;
      sv_fp_lin_lin_acc(L[cof[q][1]], L[cof[q][2]], 1.0, m2);
      sv_fp_lin_lin_acc(L[cof[q][3]], L[cof[q][4]], -1.0, m2);
      sv_fp_quad_lin_acc(m2, L[cof[q][0]], (q == 1) ? -1.0 : 1.0, A[9]);
    }
  }
  double A0[10][20]; /* the ten cubics as built: the Gauss-Newton polish below works on these */
  for (int r = 0; r < 10; ++r)
    for (int c = 0; c < 20; ++c) A0[r][c] = A[r][c];
  /* Gauss-Jordan on columns 0..9 with partial pivoting */
  for (int col = 0; col < 10; ++col) {
    int pr = col;
    for (int r = col + 1; r < 10; ++r)
      if (fabs(A[r][col]) > fabs(A[pr][col])) pr = r;
    if (!(fabs(A[pr][col]) > 1e-14)) return 0; /* degenerate sample */
    if (pr != col)
      for (int c = 0; c < 20; ++c) {
        const double tmp = A[pr][c];
        A[pr][c] = A[col][c];
        A[col][c] = tmp;
      }
    const double piv = A[col][col];
    for (int c = 0; c < 20; ++c) A[col][c] = A[col][c] / piv;
    for (int r = 0; r < 10; ++r) {
      if (r == col) continue;
      const double m = A[r][col];
      if (m == 0.0) continue;
      for (int c = 0; c < 20; ++c) A[r][c] = A[r][c] - (m * A[col][c]);
    }
  }
  /* rows (4, 5), (6, 7), (8, 9) = (x^2z, x^2), (y^2z, y^2), (xyz, xy): upper - z * lower; the tail columns 10..19 are
   * {xz^2, xz, x, yz^2, yz, y, z^3, z^2, z, 1}.  B[r] = {X (degree 3), Y (degree 3), C (degree 4)}, ascending powers of z */
  double BX[3][4], BY[3][4], BC[3][5];
  for (int r = 0; r < 3; ++r) {
    const double* u = A[4 + (2 * r)] + 10;
    const double* l = A[5 + (2 * r)] + 10;
    BX[r][0] = u[2];
    BX[r][1] = u[1] - l[2];
    BX[r][2] = u[0] - l[1];
    BX[r][3] = -l[0];
    BY[r][0] = u[5];
    BY[r][1] = u[4] - l[5];
    BY[r][2] = u[3] - l[4];
    BY[r][3] = -l[3];
    BC[r][0] = u[9];
    BC[r][1] = u[8] - l[9];
    BC[r][2] = u[7] - l[8];
    BC[r][3] = u[6] - l[7];
    BC[r][4] = -l[6];
  }
  double n10[11];
  for (int j = 0; j < 11; ++j) n10[j] = 0.0;
  { /* det B = X0 (Y1 C2 - C1 Y2) - Y0 (X1 C2 - C1 X2) + C0 (X1 Y2 - Y1 X2) */
    double m7[8], m6[7];
    for (int j = 0; j < 8; ++j) m7[j] = 0.0;
    sv_pmul_acc(BY[1], 3, BC[2], 4, 1.0, m7);
    sv_pmul_acc(BC[1], 4, BY[2], 3, -1.0, m7);
    sv_pmul_acc(BX[0], 3, m7, 7, 1.0, n10);
    for (int j = 0; j < 8; ++j) m7[j] = 0.0;
    sv_pmul_acc(BX[1], 3, BC[2], 4, 1.0, m7);
    sv_pmul_acc(BC[1], 4, BX[2], 3, -1.0, m7);
    sv_pmul_acc(BY[0], 3, m7, 7, -1.0, n10);
    for (int j = 0; j < 7; ++j) m6[j] = 0.0;
    sv_pmul_acc(BX[1], 3, BY[2], 3, 1.0, m6);
    sv_pmul_acc(BY[1], 3, BX[2], 3, -1.0, m6);
    sv_pmul_acc(BC[0], 4, m6, 6, 1.0, n10);
  }
  double cmax = 0.0;
  for (int j = 0; j < 11; ++j)
    if (fabs(n10[j]) > cmax) cmax = fabs(n10[j]);
  if (!(cmax > 0.0) || !(fabs(n10[10]) > (1e-13 * cmax))) return 0;
  sv_cplx roots[10];
  sv_poly_roots(n10, 10, roots);
  int ns = 0;
  for (int k = 0; k < 10; ++k) {
    /* a root that is real up to the accuracy of a degree-ten polynomial's roots (close pairs split into complex ones):
     * the polish decides whether there is a solution of the ten cubics next to it */
    if (!(fabs(roots[k].im) <= (1e-3 * (1.0 + fabs(roots[k].re))))) continue;
    double z = roots[k].re;
    double b[3][3];
    for (int r = 0; r < 3; ++r) {
      b[r][0] = sv_peval(BX[r], 3, z);
      b[r][1] = sv_peval(BY[r], 3, z);
      b[r][2] = sv_peval(BC[r], 4, z);
    }
    /* (x, y, 1) spans the null space of b: the cross product of two rows, the pair with the largest third component */
    double best[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < 3; ++q) {
      const double* r0 = b[(q == 2) ? 1 : 0];
      const double* r1 = b[(q == 0) ? 1 : 2];
      double cr[3];
      sv_cross(r0, r1, cr);
      if (fabs(cr[2]) > fabs(best[2])) {
        best[0] = cr[0];
        best[1] = cr[1];
        best[2] = cr[2];
      }
    }
    if (!(fabs(best[2]) > 0.0)) continue;
    double x = best[0] / best[2], y = best[1] / best[2];
    /* Gauss-Newton on the ten cubics in (x, y, z): normal equations by Cramer's rule */
    for (int itn = 0; itn < 4; ++itn) {
      const double x2 = x * x, y2 = y * y, z2 = z * z;
      const double mo[20] = {x2 * x, y2 * y, x2 * y, x * y2, x2 * z, x2, y2 * z, y2, (x * y) * z, x * y,
                             x * z2, x * z,  x,      y * z2, y * z,  y,  z2 * z, z2, z,           1.0};
      const double dx[20] = {3.0 * x2, 0.0, 2.0 * (x * y), y2, 2.0 * (x * z), 2.0 * x, 0.0, 0.0, y * z, y,
                             z2,       z,   1.0,           0.0, 0.0,          0.0,     0.0, 0.0, 0.0,   0.0};
      const double dy[20] = {0.0, 3.0 * y2, x2,  2.0 * (x * y), 0.0, 0.0, 2.0 * (y * z), 2.0 * y, x * z, x,
                             0.0, 0.0,      0.0, z2,            z,   1.0, 0.0,           0.0,     0.0,   0.0};
      const double dz[20] = {0.0,           0.0, 0.0, 0.0,           x2, 0.0, y2,       0.0,     x * y, 0.0,
                             2.0 * (x * z), x,   0.0, 2.0 * (y * z), y,  0.0, 3.0 * z2, 2.0 * z, 1.0,   0.0};
      double h[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, g[3] = {0.0, 0.0, 0.0}; /* J^T J (xx xy xz yy yz zz), J^T r */
      for (int r = 0; r < 10; ++r) {
        double rv = 0.0, jx = 0.0, jy = 0.0, jz = 0.0;
        for (int c = 0; c < 20; ++c) {
          rv = rv + (A0[r][c] * mo[c]);
          jx = jx + (A0[r][c] * dx[c]);
          jy = jy + (A0[r][c] * dy[c]);
          jz = jz + (A0[r][c] * dz[c]);
        }
        h[0] = h[0] + (jx * jx);
        h[1] = h[1] + (jx * jy);
        h[2] = h[2] + (jx * jz);
        h[3] = h[3] + (jy * jy);
        h[4] = h[4] + (jy * jz);
        h[5] = h[5] + (jz * jz);
        g[0] = g[0] + (jx * rv);
        g[1] = g[1] + (jy * rv);
        g[2] = g[2] + (jz * rv);
      }
      const double c00 = (h[3] * h[5]) - (h[4] * h[4]), c01 = (h[2] * h[4]) - (h[1] * h[5]), c02 = (h[1] * h[4]) - (h[2] * h[3]);
      const double det = ((h[0] * c00) + (h[1] * c01)) + (h[2] * c02);
      if (!(fabs(det) > 0.0)) break;
      const double c11 = (h[0] * h[5]) - (h[2] * h[2]), c12 = (h[1] * h[2]) - (h[0] * h[4]), c22 = (h[0] * h[3]) - (h[1] * h[1]);
      x = x - ((((c00 * g[0]) + (c01 * g[1])) + (c02 * g[2])) / det);
      y = y - ((((c01 * g[0]) + (c11 * g[1])) + (c12 * g[2])) / det);
      z = z - ((((c02 * g[0]) + (c12 * g[1])) + (c22 * g[2])) / det);
    }
    double* E = E_out + (9 * ns);
    double nn = 0.0;
    for (int q = 0; q < 9; ++q) {
      E[q] = (((x * L[q][0]) + (y * L[q][1])) + (z * L[q][2])) + L[q][3];
      nn = nn + (E[q] * E[q]);
    }
    nn = sqrt(nn);
    if (!(nn > 0.0)) continue;
    for (int q = 0; q < 9; ++q) E[q] = E[q] / nn;
    /* keep it only if it IS an essential matrix: |2 E E^T E - tr(E E^T) E| small on the unit-norm E (a complex pair's
     * real part does not polish into a solution) */
    double G[9], tr = 0.0, worst = 0.0;
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) G[(3 * a) + c] = ((E[3 * a] * E[3 * c]) + (E[(3 * a) + 1] * E[(3 * c) + 1])) + (E[(3 * a) + 2] * E[(3 * c) + 2]);
    tr = (G[0] + G[4]) + G[8];
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) {
        const double v = (2.0 * (((G[3 * a] * E[c]) + (G[(3 * a) + 1] * E[3 + c])) + (G[(3 * a) + 2] * E[6 + c]))) - (tr * E[(3 * a) + c]);
        if (fabs(v) > worst) worst = fabs(v);
      }
    if (!(worst <= 1e-9)) continue;
    ns++;
  }
  return ns;
}

/* One RANSAC hypothesis: distinct correspondences from the counter-based sampler (the solver's points first, then the
 * ones OpenGV's problem class draws for disambiguation: 5 + 3, 7 + 2, 8), the minimal solver, and among all the
 * decompositions of all its essential matrices the (R, t) with the smallest summed score over the whole sample. */
__device__ static int sv_rel_hypothesis(const double* f1, const double* f2, int32_t n, int algorithm, uint64_t seed, uint64_t it,
                                     double* R, double* t) {
  int32_t s[9];
  const int k = (algorithm == SV_REL_SEVENPT) ? 9 : 8;
  if (algorithm != SV_REL_EIGHTPT && algorithm != SV_REL_SEVENPT && algorithm != SV_REL_FIVEPT) return 0;
  if (!sv_sample_distinct(n, k, seed, it, s)) return 0;
  double a[27], b[27];
  for (int i = 0; i < k; ++i)
    for (int c = 0; c < 3; ++c) {
      a[3 * i + c] = f1[3 * s[i] + c];
      b[3 * i + c] = f2[3 * s[i] + c];
    }
  if (algorithm == SV_REL_EIGHTPT) return sv_eightpt(a, b, R, t);
  double E[90];
  const int ns = (algorithm == SV_REL_SEVENPT) ? sv_sevenpt(a, b, E) : sv_fivept(a, b, E);
  double best_q = 1e300;
  for (int q = 0; q < ns; ++q) sv_rel_decompose_pick(E + (9 * q), a, b, k, &best_q, R, t);
  return best_q < 1e300;
}
