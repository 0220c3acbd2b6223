/* TEST INFRASTRUCTURE ONLY -- CPU oracle (see ransac.c for the provenance header).
 *
 * EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009) on n >= 5 bearing / point pairs, the central absolute-pose solver
 * OpenGV runs inside absolute_pose_ransac(..., "EPNP", ...) on 6-point samples (omnistereo/pose_est_tools.py:697,
 * :915).  OpenGV is not in the reference tree: this is a restatement of the published algorithm (four control points
 * from the principal axes of the world points, barycentric coordinates, the null space of M^T M by a symmetric
 * eigen-decomposition, the three beta initialisations with five Gauss-Newton steps each, absolute orientation, the
 * candidate with the smallest reprojection error), with own numerics: Jacobi rotations for the symmetric
 * eigen-problems (row-cyclic order for the 3 x 3 ones, round-robin order for the 12 x 12 one), normal equations for the small least-squares fits, and the rotation from the Jacobi SVD of the
 * 3x3 correlation matrix completed to a proper rotation by cross products.
 * Only + - * / sqrt and comparisons, fully parenthesised, no FMA contraction.  Everything outside the section marked
 * "oracle only" below is ALSO the text of the HIP side: tests/gen_device_headers.py writes
 * vo_single_camera_sos_amd/csrc/epnp_core.h from it (tests/test_abi.py checks that the two stay identical).  The one piece the
 * device words differently is the 12 x 12 eigen-solver (registers, rounds unrolled: csrc/epnp_eig12_reg.h, hand-written);
 * it is checked against orc_jacobi12_rr by the bit-exact GPU parity tests, and orc_jacobi12_rr against numpy.linalg.eigh and
 * against the independent QL solver of this file (tests/test_oracle_ransac.py).
 */
#pragma once
#include "ransac_core.h"

#define ORC_EPNP_MAXN 8
#define ORC_JACOBI_MAXN 12

/* Cyclic Jacobi on a symmetric n x n matrix A (row-major, destroyed: its diagonal ends as the eigenvalues), n <=
 * ORC_JACOBI_MAXN; V (n x n) receives the eigenvectors as columns.  The classic symmetric update: a rotation in the
 * (p, q) plane changes rows / columns p and q only -- a'kp = c akp - s akq, a'kq = s akp + c akq for k != p, q
 * (mirrored), a'pp = app - t apq, a'qq = aqq + t apq, a'pq = 0 exactly.  The iterations of each inner loop touch
 * disjoint elements: all their loads come before the stores. */
static inline void orc_jacobi_sym(double* A, int n, double* V) {
  #pragma unroll
  for (int i = 0; i < n; ++i)
    #pragma unroll
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    #pragma unroll
    for (int p = 0; p < n; ++p) {
      diag = diag + (A[p * n + p] * A[p * n + p]);
      #pragma unroll
      for (int q = p + 1; q < n; ++q) off = off + (A[p * n + q] * A[p * n + q]);
    }
    if (!(off > (1e-40 * diag))) break;
    #pragma unroll
    for (int p = 0; p < n - 1; ++p)
      #pragma unroll
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double app = A[p * n + p], aqq = A[q * n + q];
        double xp[ORC_JACOBI_MAXN], xq[ORC_JACOBI_MAXN], yp[ORC_JACOBI_MAXN], yq[ORC_JACOBI_MAXN];
        #pragma unroll
        for (int k = 0; k < n; ++k) {  // (rows k of columns p, q: the values of k = p, q are loaded but not used)
          xp[k] = A[k * n + p];
          xq[k] = A[k * n + q];
          yp[k] = V[k * n + p];
          yq[k] = V[k * n + q];
        }
        const double theta = (aqq - app) / (2.0 * apq);
        const double at = theta < 0.0 ? -theta : theta;
        const double t = (theta < 0.0 ? -1.0 : 1.0) / (at + sqrt((theta * theta) + 1.0));
        const double c = 1.0 / sqrt((t * t) + 1.0), s = t * c;
        #pragma unroll
        for (int k = 0; k < n; ++k) {
          if (k == p || k == q) continue;
          const double x = (c * xp[k]) - (s * xq[k]), y = (s * xp[k]) + (c * xq[k]);
          A[k * n + p] = x;
          A[p * n + k] = x;
          A[k * n + q] = y;
          A[q * n + k] = y;
        }
        A[p * n + p] = app - (t * apq);
        A[q * n + q] = aqq + (t * apq);
        A[p * n + q] = 0.0;
        A[q * n + p] = 0.0;
        #pragma unroll
        for (int k = 0; k < n; ++k) {
          V[k * n + p] = (c * yp[k]) - (s * yq[k]);
          V[k * n + q] = (s * yp[k]) + (c * yq[k]);
        }
      }
  }
}

/* Least squares min |A x - b| for an m x k system (k <= 5) through the normal equations, Gaussian elimination
 * with partial pivoting.  A row-major with row stride lda.  Returns 0 on a vanishing pivot. */
static inline int orc_lsq_small(const double* A, int lda, const double* b, int m, int k, double* x) {
  double N[5 * 6];
  #pragma unroll
  for (int i = 0; i < k; ++i) {
    #pragma unroll
    for (int j = 0; j < k; ++j) {
      double s = 0.0;
      #pragma unroll
      for (int r = 0; r < m; ++r) s = s + (A[r * lda + i] * A[r * lda + j]);
      N[i * 6 + j] = s;
    }
    double s = 0.0;
    #pragma unroll
    for (int r = 0; r < m; ++r) s = s + (A[r * lda + i] * b[r]);
    N[i * 6 + 5] = s;
  }
  #pragma unroll
  for (int c = 0; c < k; ++c) {
    int piv = c;
    double best = N[c * 6 + c] < 0.0 ? -N[c * 6 + c] : N[c * 6 + c];
    #pragma unroll
    for (int r = c + 1; r < k; ++r) {
      const double v = N[r * 6 + c] < 0.0 ? -N[r * 6 + c] : N[r * 6 + c];
      if (v > best) {
        best = v;
        piv = r;
      }
    }
    if (!(best > 0.0)) return 0;
    #pragma unroll
    for (int r = c + 1; r < k; ++r) { /* swap rows c and piv; written over static row numbers (registers, not scratch) */
      const bool sw = piv == r;
      #pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double tc = N[c * 6 + j], tr = N[r * 6 + j];
        N[c * 6 + j] = sw ? tr : tc;
        N[r * 6 + j] = sw ? tc : tr;
      }
    }
    #pragma unroll
    for (int r = c + 1; r < k; ++r) {
      const double fct = N[r * 6 + c] / N[c * 6 + c];
      #pragma unroll
      for (int j = c; j < k; ++j) N[r * 6 + j] = N[r * 6 + j] - (fct * N[c * 6 + j]);
      N[r * 6 + 5] = N[r * 6 + 5] - (fct * N[c * 6 + 5]);
    }
  }
  #pragma unroll
  for (int c = k - 1; c >= 0; --c) {
    double s = N[c * 6 + 5];
    #pragma unroll
    for (int j = c + 1; j < k; ++j) s = s - (N[c * 6 + j] * x[j]);
    x[c] = s / N[c * 6 + c];
  }
  return 1;
}

/* world -> camera (Rcw, tcw) from the control-point coordinates in the camera frame; returns the mean
 * reprojection error (normalised image plane) or a negative number on failure. */
static inline double orc_epnp_pose_from_betas(const double* betas, const double* vv /*[4][12]*/, const double* alphas,
                                              const double* pw, const double* uv, int n, double* Rcw, double* tcw) {
  double ccs[12], pcs[3 * ORC_EPNP_MAXN];
  #pragma unroll
  for (int j = 0; j < 12; ++j)
    ccs[j] = (((betas[0] * vv[j]) + (betas[1] * vv[12 + j])) + (betas[2] * vv[24 + j])) + (betas[3] * vv[36 + j]);
  #pragma unroll
  for (int i = 0; i < n; ++i)
    #pragma unroll
    for (int k = 0; k < 3; ++k)
      pcs[3 * i + k] = (((alphas[4 * i] * ccs[k]) + (alphas[4 * i + 1] * ccs[3 + k])) + (alphas[4 * i + 2] * ccs[6 + k])) +
                       (alphas[4 * i + 3] * ccs[9 + k]);
  if (pcs[2] < 0.0) { /* the points must lie in front of the camera */
    #pragma unroll
    for (int j = 0; j < 12; ++j) ccs[j] = -ccs[j];
    #pragma unroll
    for (int j = 0; j < 3 * n; ++j) pcs[j] = -pcs[j];
  }
  /* absolute orientation: H = sum (pc - pc0)(pw - pw0)^T, R = U V^T of its SVD made proper */
  double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
  #pragma unroll
  for (int i = 0; i < n; ++i)
    #pragma unroll
    for (int k = 0; k < 3; ++k) {
      pc0[k] = pc0[k] + pcs[3 * i + k];
      pw0[k] = pw0[k] + pw[3 * i + k];
    }
  #pragma unroll
  for (int k = 0; k < 3; ++k) {
    pc0[k] = pc0[k] / (double)n;
    pw0[k] = pw0[k] / (double)n;
  }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  #pragma unroll
  for (int i = 0; i < n; ++i)
    #pragma unroll
    for (int r = 0; r < 3; ++r)
      #pragma unroll
      for (int c = 0; c < 3; ++c) H[3 * r + c] = H[3 * r + c] + ((pcs[3 * i + r] - pc0[r]) * (pw[3 * i + c] - pw0[c]));
  double S[9], V[9];
  #pragma unroll
  for (int r = 0; r < 3; ++r)
    #pragma unroll
    for (int c = 0; c < 3; ++c) S[3 * r + c] = ((H[r] * H[c]) + (H[3 + r] * H[3 + c])) + (H[6 + r] * H[6 + c]); /* H^T H */
  orc_jacobi_sym(S, 3, V);
  /* eigenvalues descending, with their eigenvectors (columns of V): a three-element sorting network on the values
   * instead of runtime indices into S and V (same comparisons, same outcome; keeps everything in registers) */
  double e0 = S[0], e1 = S[4], e2 = S[8];
  double v0[3] = {V[0], V[3], V[6]}, v1[3] = {V[1], V[4], V[7]}, v2[3] = {V[2], V[5], V[8]};
#define ORC_CSWAP(ea, va, eb, vb)                   \
  if (ea < eb) {                                   \
    double t_ = ea; ea = eb; eb = t_;              \
    t_ = va[0]; va[0] = vb[0]; vb[0] = t_;         \
    t_ = va[1]; va[1] = vb[1]; vb[1] = t_;         \
    t_ = va[2]; va[2] = vb[2]; vb[2] = t_;         \
  }
  ORC_CSWAP(e0, v0, e1, v1)
  ORC_CSWAP(e1, v1, e2, v2)
  ORC_CSWAP(e0, v0, e1, v1)
#undef ORC_CSWAP
  if (!(e1 > 0.0)) return -1.0; /* rank < 2: no orientation */
  v2[0] = (v0[1] * v1[2]) - (v0[2] * v1[1]);
  v2[1] = (v0[2] * v1[0]) - (v0[0] * v1[2]);
  v2[2] = (v0[0] * v1[1]) - (v0[1] * v1[0]);
  double u0[3], u1[3], u2[3];
  const double s0 = sqrt(e0), s1 = sqrt(e1);
  #pragma unroll
  for (int r = 0; r < 3; ++r) {
    u0[r] = (((H[3 * r] * v0[0]) + (H[3 * r + 1] * v0[1])) + (H[3 * r + 2] * v0[2])) / s0;
    u1[r] = (((H[3 * r] * v1[0]) + (H[3 * r + 1] * v1[1])) + (H[3 * r + 2] * v1[2])) / s1;
  }
  { /* re-orthonormalise u1 against u0 (they are orthogonal up to rounding) */
    const double d = ((u0[0] * u1[0]) + (u0[1] * u1[1])) + (u0[2] * u1[2]);
    #pragma unroll
    for (int r = 0; r < 3; ++r) u1[r] = u1[r] - (d * u0[r]);
    const double nn = sqrt(((u1[0] * u1[0]) + (u1[1] * u1[1])) + (u1[2] * u1[2]));
    if (!(nn > 0.0)) return -1.0;
    #pragma unroll
    for (int r = 0; r < 3; ++r) u1[r] = u1[r] / nn;
  }
  u2[0] = (u0[1] * u1[2]) - (u0[2] * u1[1]);
  u2[1] = (u0[2] * u1[0]) - (u0[0] * u1[2]);
  u2[2] = (u0[0] * u1[1]) - (u0[1] * u1[0]);
  #pragma unroll
  for (int r = 0; r < 3; ++r)
    #pragma unroll
    for (int c = 0; c < 3; ++c) Rcw[3 * r + c] = ((u0[r] * v0[c]) + (u1[r] * v1[c])) + (u2[r] * v2[c]);
  #pragma unroll
  for (int r = 0; r < 3; ++r)
    tcw[r] = pc0[r] - (((Rcw[3 * r] * pw0[0]) + (Rcw[3 * r + 1] * pw0[1])) + (Rcw[3 * r + 2] * pw0[2]));
  double err = 0.0;
  #pragma unroll
  for (int i = 0; i < n; ++i) {
    const double X = (((Rcw[0] * pw[3 * i]) + (Rcw[1] * pw[3 * i + 1])) + (Rcw[2] * pw[3 * i + 2])) + tcw[0];
    const double Y = (((Rcw[3] * pw[3 * i]) + (Rcw[4] * pw[3 * i + 1])) + (Rcw[5] * pw[3 * i + 2])) + tcw[1];
    const double Z = (((Rcw[6] * pw[3 * i]) + (Rcw[7] * pw[3 * i + 1])) + (Rcw[8] * pw[3 * i + 2])) + tcw[2];
    const double du = uv[2 * i] - (X / Z), dv = uv[2 * i + 1] - (Y / Z);
    err = err + sqrt((du * du) + (dv * dv));
  }
  err = err / (double)n;
  return (err == err) ? err : -1.0;
}

/* EPnP, first half: f, p: n rows of 3 (bearings in the camera, points in the world), 5 <= n <= ORC_EPNP_MAXN ->
 * normalised image coordinates uv, control points cw, barycentric coordinates alphas.  0 on failure. */
static inline int orc_epnp_front(const double* f, const double* p, int n, double* uv, double* cw, double* alphas) {
  if (n < 5 || n > ORC_EPNP_MAXN) return 0; /* 4 points leave a 4-dimensional null space: not handled */
  #pragma unroll
  for (int i = 0; i < n; ++i) {
    if (!(f[3 * i + 2] != 0.0)) return 0;
    uv[2 * i] = f[3 * i] / f[3 * i + 2];
    uv[2 * i + 1] = f[3 * i + 1] / f[3 * i + 2];
  }
  /* control points: centroid + principal axes scaled by sqrt(eigenvalue / n) */
#pragma unroll
  for (int k = 0; k < 12; ++k) cw[k] = 0.0;
  #pragma unroll
  for (int i = 0; i < n; ++i)
    #pragma unroll
    for (int k = 0; k < 3; ++k) cw[k] = cw[k] + p[3 * i + k];
  #pragma unroll
  for (int k = 0; k < 3; ++k) cw[k] = cw[k] / (double)n;
  double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, E[9];
  #pragma unroll
  for (int i = 0; i < n; ++i)
    #pragma unroll
    for (int r = 0; r < 3; ++r)
      #pragma unroll
      for (int c = 0; c < 3; ++c) C[3 * r + c] = C[3 * r + c] + ((p[3 * i + r] - cw[r]) * (p[3 * i + c] - cw[c]));
  orc_jacobi_sym(C, 3, E);
  #pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double lam = C[4 * a] > 0.0 ? C[4 * a] : 0.0;
    const double kk = sqrt(lam / (double)n);
    #pragma unroll
    for (int k = 0; k < 3; ++k) cw[3 * (a + 1) + k] = cw[k] + (kk * E[3 * k + a]);
  }
  /* barycentric coordinates: CC a = p - c0, CC columns = c_j - c_0 */
  double CC[9];
  #pragma unroll
  for (int r = 0; r < 3; ++r)
    #pragma unroll
    for (int c = 0; c < 3; ++c) CC[3 * r + c] = cw[3 * (c + 1) + r] - cw[r];
  const double det = ((CC[0] * ((CC[4] * CC[8]) - (CC[5] * CC[7]))) - (CC[1] * ((CC[3] * CC[8]) - (CC[5] * CC[6])))) +
                     (CC[2] * ((CC[3] * CC[7]) - (CC[4] * CC[6])));
  if (!(det != 0.0) || !(det == det)) return 0;
  double Ci[9];
  Ci[0] = ((CC[4] * CC[8]) - (CC[5] * CC[7])) / det;
  Ci[1] = ((CC[2] * CC[7]) - (CC[1] * CC[8])) / det;
  Ci[2] = ((CC[1] * CC[5]) - (CC[2] * CC[4])) / det;
  Ci[3] = ((CC[5] * CC[6]) - (CC[3] * CC[8])) / det;
  Ci[4] = ((CC[0] * CC[8]) - (CC[2] * CC[6])) / det;
  Ci[5] = ((CC[2] * CC[3]) - (CC[0] * CC[5])) / det;
  Ci[6] = ((CC[3] * CC[7]) - (CC[4] * CC[6])) / det;
  Ci[7] = ((CC[1] * CC[6]) - (CC[0] * CC[7])) / det;
  Ci[8] = ((CC[0] * CC[4]) - (CC[1] * CC[3])) / det;
  #pragma unroll
  for (int i = 0; i < n; ++i) {
    const double d0 = p[3 * i] - cw[0], d1 = p[3 * i + 1] - cw[1], d2 = p[3 * i + 2] - cw[2];
    #pragma unroll
    for (int j = 0; j < 3; ++j) alphas[4 * i + 1 + j] = ((Ci[3 * j] * d0) + (Ci[3 * j + 1] * d1)) + (Ci[3 * j + 2] * d2);
    alphas[4 * i] = ((1.0 - alphas[4 * i + 1]) - alphas[4 * i + 2]) - alphas[4 * i + 3];
  }
  return 1;
}

/* EPnP, second half: the four null-space vectors vv (vv[0] of the smallest eigenvalue) -> R, t: pose of the camera in
 * the world (points map by R^T (p - t)), as pyopengv returns it.  0 on failure. */
static inline int orc_epnp_back(const double* p, int n, const double* uv, const double* cw, const double* alphas,
                                   const double* vv, double* R, double* t) {
  /* L (6 x 10) and rho (6) over the control-point pairs */
  const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
  double L[60], rho[6];
  #pragma unroll
  for (int j = 0; j < 6; ++j) {
    double dv[4][3];
    #pragma unroll
    for (int a = 0; a < 4; ++a)
      #pragma unroll
      for (int k = 0; k < 3; ++k) dv[a][k] = vv[12 * a + 3 * pa[j] + k] - vv[12 * a + 3 * pb[j] + k];
#define ORC_D(a, b) (((dv[a][0] * dv[b][0]) + (dv[a][1] * dv[b][1])) + (dv[a][2] * dv[b][2]))
    L[10 * j + 0] = ORC_D(0, 0);
    L[10 * j + 1] = 2.0 * ORC_D(0, 1);
    L[10 * j + 2] = ORC_D(1, 1);
    L[10 * j + 3] = 2.0 * ORC_D(0, 2);
    L[10 * j + 4] = 2.0 * ORC_D(1, 2);
    L[10 * j + 5] = ORC_D(2, 2);
    L[10 * j + 6] = 2.0 * ORC_D(0, 3);
    L[10 * j + 7] = 2.0 * ORC_D(1, 3);
    L[10 * j + 8] = 2.0 * ORC_D(2, 3);
    L[10 * j + 9] = ORC_D(3, 3);
#undef ORC_D
    const double e0 = cw[3 * pa[j]] - cw[3 * pb[j]], e1 = cw[3 * pa[j] + 1] - cw[3 * pb[j] + 1],
                 e2 = cw[3 * pa[j] + 2] - cw[3 * pb[j] + 2];
    rho[j] = ((e0 * e0) + (e1 * e1)) + (e2 * e2);
  }
  double best_err = -1.0, Rb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tb[3] = {0, 0, 0};
  /* (a real loop on the device: the three initialisations share ONE copy of the Gauss-Newton steps and of the absolute
   * orientation -- a third of the code and of the registers of the unrolled form, so that two waves fit a SIMD and the
   * divisions / square roots of one hide behind the other's arithmetic) */
  #pragma unroll 1
  for (int variant = 0; variant < 3; ++variant) {
    double betas[4] = {0, 0, 0, 0}, A[6 * 5], x[5];
    int ok;
    if (variant == 0) { /* betas10 = [B11 B12 B13 B14] */
      #pragma unroll
      for (int j = 0; j < 6; ++j) {
        A[5 * j] = L[10 * j];
        A[5 * j + 1] = L[10 * j + 1];
        A[5 * j + 2] = L[10 * j + 3];
        A[5 * j + 3] = L[10 * j + 6];
      }
      ok = orc_lsq_small(A, 5, rho, 6, 4, x);
      if (ok) {
        if (x[0] < 0.0) {
          betas[0] = sqrt(-x[0]);
          betas[1] = -x[1] / betas[0];
          betas[2] = -x[2] / betas[0];
          betas[3] = -x[3] / betas[0];
        } else {
          betas[0] = sqrt(x[0]);
          betas[1] = x[1] / betas[0];
          betas[2] = x[2] / betas[0];
          betas[3] = x[3] / betas[0];
        }
      }
    } else { /* [B11 B12 B22] and [B11 B12 B22 B13 B23] */
      #pragma unroll
      for (int j = 0; j < 6; ++j)
        #pragma unroll
        for (int c = 0; c < 5; ++c) A[5 * j + c] = L[10 * j + c];
      x[3] = 0.0;
      x[4] = 0.0;
      if (variant == 1) ok = orc_lsq_small(A, 5, rho, 6, 3, x); /* (the system size stays a constant at each call) */
      else ok = orc_lsq_small(A, 5, rho, 6, 5, x);
      if (ok) {
        if (x[0] < 0.0) {
          betas[0] = sqrt(-x[0]);
          betas[1] = (x[2] < 0.0) ? sqrt(-x[2]) : 0.0;
        } else {
          betas[0] = sqrt(x[0]);
          betas[1] = (x[2] > 0.0) ? sqrt(x[2]) : 0.0;
        }
        if (x[1] < 0.0) betas[0] = -betas[0];
        betas[2] = variant == 2 ? (x[3] / betas[0]) : 0.0;
        betas[3] = 0.0;
      }
    }
    if (!ok) continue;
    #pragma unroll 1
    for (int itn = 0; itn < 5; ++itn) { /* Gauss-Newton on the six distance constraints */
      double J[6 * 5], r[6], dx[4];
      #pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double* l = L + 10 * j;
        J[5 * j] = (((2.0 * l[0]) * betas[0]) + (l[1] * betas[1])) + ((l[3] * betas[2]) + (l[6] * betas[3]));
        J[5 * j + 1] = ((l[1] * betas[0]) + ((2.0 * l[2]) * betas[1])) + ((l[4] * betas[2]) + (l[7] * betas[3]));
        J[5 * j + 2] = ((l[3] * betas[0]) + (l[4] * betas[1])) + (((2.0 * l[5]) * betas[2]) + (l[8] * betas[3]));
        J[5 * j + 3] = ((l[6] * betas[0]) + (l[7] * betas[1])) + ((l[8] * betas[2]) + ((2.0 * l[9]) * betas[3]));
        r[j] = rho[j] - (((((l[0] * betas[0]) * betas[0]) + ((l[1] * betas[0]) * betas[1])) +
                          (((l[2] * betas[1]) * betas[1]) + ((l[3] * betas[0]) * betas[2]))) +
                         ((((l[4] * betas[1]) * betas[2]) + ((l[5] * betas[2]) * betas[2])) +
                          ((((l[6] * betas[0]) * betas[3]) + ((l[7] * betas[1]) * betas[3])) +
                           (((l[8] * betas[2]) * betas[3]) + ((l[9] * betas[3]) * betas[3])))));
      }
      if (!orc_lsq_small(J, 5, r, 6, 4, dx)) break;
      #pragma unroll
      for (int a = 0; a < 4; ++a) betas[a] = betas[a] + dx[a];
    }
    double Rc[9], tc[3];
    const double err = orc_epnp_pose_from_betas(betas, vv, alphas, p, uv, n, Rc, tc);
    if (err >= 0.0 && (best_err < 0.0 || err < best_err)) {
      best_err = err;
      #pragma unroll
      for (int k = 0; k < 9; ++k) Rb[k] = Rc[k];
      #pragma unroll
      for (int k = 0; k < 3; ++k) tb[k] = tc[k];
    }
  }
  if (best_err < 0.0) return 0;
  /* camera pose in the world: R = Rcw^T, t = -Rcw^T tcw */
  #pragma unroll
  for (int r = 0; r < 3; ++r)
    #pragma unroll
    for (int c = 0; c < 3; ++c) R[3 * r + c] = Rb[3 * c + r];
  #pragma unroll
  for (int r = 0; r < 3; ++r) t[r] = -(((Rb[r] * tb[0]) + (Rb[3 + r] * tb[1])) + (Rb[6 + r] * tb[2]));
  #pragma unroll
  for (int k = 0; k < 9; ++k)
    if (!(R[k] == R[k])) return 0;
  #pragma unroll
  for (int k = 0; k < 3; ++k)
    if (!(t[k] == t[k])) return 0;
  return 1;
}

/* k distinct indices below n from the counter-based generator (draw numbers d0, d0 + 1, ...): the j-th draw picks
 * among the n - j values not taken yet (kept sorted).  Returns 0 if n < k. */
static inline int orc_sample_distinct(int32_t n, int k, uint64_t seed, uint64_t it, int32_t* s) {
  if (n < k || k > 9) return 0;
  int32_t taken[9]; /* ascending */
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    if (j >= k) break;
    int32_t v = (int32_t)orc_below(orc_mix64(seed, it, (uint64_t)j), (uint32_t)(n - j));
    bool skipping = true;
#pragma unroll
    for (int pos = 0; pos < j; ++pos) { /* skip the values already taken */
      if (skipping && v >= taken[pos]) v++;
      else skipping = false;
    }
    taken[j] = v;
#pragma unroll
    for (int q = j; q > 0; --q) /* keep `taken` ascending: the new value sinks to its place */
      if (taken[q - 1] > taken[q]) {
        const int32_t tmp = taken[q - 1];
        taken[q - 1] = taken[q];
        taken[q] = tmp;
      }
    s[j] = v;
  }
  return 1;
}

/* @oracle-only: begin -- the eigen-solvers of the 12 x 12 problem as plain loops, and the solver's driver */
/* A second, independent solver for the 12 x 12 symmetric eigen-problem -- used by the TESTS only, to cross-check the
 * Jacobi solver below (and both against numpy): Householder reduction to tridiagonal form followed by the QL algorithm
 * with implicit shifts, both accumulating the transformation (the classic EISPACK tred2 / tql2 pair as published in
 * Numerical Recipes ch. 11.2-11.3, restated).  ~10 k floating-point operations against ~70 k for Jacobi sweeps; on
 * the GPU it still lost to Jacobi (measured: 2.3 against 1.3 ms per 256 k matrices), see orc_jacobi12_rr.
 * A: 12 x 12 row-major, symmetric on entry; on success its COLUMNS are the eigenvectors and d the eigenvalues
 * (unordered).  e: 12 doubles of scratch.  Returns 0 if an eigenvalue needs more than 30 QL iterations.
 * Only + - * / sqrt, fabs, comparisons and EXPLICIT fused multiply-adds (fma(): one rounding, the same on both sides;
 * the inner products and the eigenvector updates are written with it, a third fewer operations), fully parenthesised:
 * the HIP side repeats every operation in this order. */
#define ORC_EIG_N 12
/* a sub-diagonal element counts as zero below this fraction of its neighbours' magnitude: QL converges cubically, the
 * last iteration of an eigenvalue would take it from ~1e-8 to ~1e-24 for nothing a RANSAC hypothesis can use */
#define ORC_EIG_EPS 1e-12
static inline int orc_symeig12(double* A, double* d, double* e) {
  const int n = ORC_EIG_N;
  /* ---- Householder reduction; A becomes the orthogonal matrix Q of the reduction ---- */
  for (int i = n - 1; i >= 1; --i) {
    const int l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0) {
      for (int k = 0; k <= l; ++k) scale = scale + fabs(A[i * n + k]);
      if (scale == 0.0) {
        e[i] = A[i * n + l];
      } else {
        for (int k = 0; k <= l; ++k) {
          A[i * n + k] = A[i * n + k] / scale;
          h = h + (A[i * n + k] * A[i * n + k]);
        }
        double f = A[i * n + l];
        double g = f >= 0.0 ? -sqrt(h) : sqrt(h);
        e[i] = scale * g;
        h = h - (f * g);
        A[i * n + l] = f - g;
        f = 0.0;
        for (int j = 0; j <= l; ++j) {
          A[j * n + i] = A[i * n + j] / h;
          g = 0.0;
          for (int k = 0; k <= j; ++k) g = fma(A[j * n + k], A[i * n + k], g);
          for (int k = j + 1; k <= l; ++k) g = fma(A[k * n + j], A[i * n + k], g);
          e[j] = g / h;
          f = f + (e[j] * A[i * n + j]);
        }
        const double hh = f / (h + h);
        for (int j = 0; j <= l; ++j) {
          f = A[i * n + j];
          g = e[j] - (hh * f);
          e[j] = g;
          for (int k = 0; k <= j; ++k) A[j * n + k] = A[j * n + k] - fma(f, e[k], g * A[i * n + k]);
        }
      }
    } else {
      e[i] = A[i * n + l];
    }
    d[i] = h;
  }
  d[0] = 0.0;
  e[0] = 0.0;
  for (int i = 0; i < n; ++i) {
    const int l = i - 1;
    if (d[i] != 0.0) {
      for (int j = 0; j <= l; ++j) {
        double g = 0.0;
        for (int k = 0; k <= l; ++k) g = fma(A[i * n + k], A[k * n + j], g);
        for (int k = 0; k <= l; ++k) A[k * n + j] = fma(-g, A[k * n + i], A[k * n + j]);
      }
    }
    d[i] = A[i * n + i];
    A[i * n + i] = 1.0;
    for (int j = 0; j <= l; ++j) {
      A[j * n + i] = 0.0;
      A[i * n + j] = 0.0;
    }
  }
  /* ---- QL with implicit shifts on (d, e), rotations accumulated into the columns of A ---- */
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= (ORC_EIG_EPS * dd)) break;
      }
      if (m != l) {
        if (iter++ == 30) return 0;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = sqrt((g * g) + 1.0);
        g = (d[m] - d[l]) + (e[l] / (g + (g >= 0.0 ? r : -r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          r = sqrt((f * f) + (g * g));
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] = d[i + 1] - p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = ((d[i] - g) * s) + ((2.0 * c) * b);
          p = s * r;
          d[i + 1] = g + p;
          g = (c * r) - b;
          for (int k = 0; k < n; ++k) {
            f = A[k * n + i + 1];
            const double zi = A[k * n + i];
            A[k * n + i + 1] = fma(s, zi, c * f);
            A[k * n + i] = fma(c, zi, -(s * f));
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] = d[l] - p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  return 1;
}

/* The eigen-solver EPnP actually uses for M^T M: Jacobi rotations in ROUND-ROBIN order (the circle method of a
 * tournament with 12 players: 11 rounds of 6 pairs whose index sets are disjoint; pair i of round r is (r, 11) for
 * i = 0 and ((r + i) mod 11, (r - i) mod 11) otherwise, smaller index first).  On a SIMD machine Jacobi beats the QL
 * solver above although it costs ~7x the arithmetic: every lane of a wave does the same work per sweep (QL's iteration
 * counts differ from lane to lane and the wave pays for the slowest), and the rotations of a round commute as far as
 * their angles go -- a rotation in the (p, q) plane leaves app, aqq, apq of a disjoint pair untouched -- so a lane
 * evaluates the six divide / square-root chains of a round side by side and then applies the rotations one after the
 * other, with exactly the result of this sequential loop.  A: 12 x 12 row-major symmetric (destroyed: its diagonal
 * ends as the eigenvalues), V: eigenvectors as columns.  Sweeps stop when the off-diagonal mass is below 1e-26 of the
 * diagonal's (squared norms): eigenvectors to ~1e-13, far inside what a RANSAC hypothesis needs. */
#define ORC_JACOBI12_TOL 1e-26
static inline int orc_rr_first(int idx) {
  const int r = idx / 6, i = idx % 6;
  const int a = i == 0 ? r : (r + i) % 11, b = i == 0 ? 11 : (r - i + 11) % 11;
  return a < b ? a : b;
}
static inline int orc_rr_second(int idx) {
  const int r = idx / 6, i = idx % 6;
  const int a = i == 0 ? r : (r + i) % 11, b = i == 0 ? 11 : (r - i + 11) % 11;
  return a < b ? b : a;
}
static inline void orc_jacobi12_rr(double* A, double* V) {
  const int n = 12;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < n; ++p) {
      diag = diag + (A[p * n + p] * A[p * n + p]);
      for (int q = p + 1; q < n; ++q) off = off + (A[p * n + q] * A[p * n + q]);
    }
    if (!(off > (ORC_JACOBI12_TOL * diag))) break;
    for (int idx = 0; idx < 66; ++idx) {
      const int p = orc_rr_first(idx), q = orc_rr_second(idx);
      const double apq = A[p * n + q];
      if (apq == 0.0) continue;
      const double app = A[p * n + p], aqq = A[q * n + q];
      const double theta = (aqq - app) / (2.0 * apq);
      const double at = theta < 0.0 ? -theta : theta;
      const double t = (theta < 0.0 ? -1.0 : 1.0) / (at + sqrt((theta * theta) + 1.0));
      const double c = 1.0 / sqrt((t * t) + 1.0), s = t * c;
      for (int k = 0; k < n; ++k) {
        if (k == p || k == q) continue;
        const double akp = A[k * n + p], akq = A[k * n + q];
        const double x = (c * akp) - (s * akq), y = (s * akp) + (c * akq);
        A[k * n + p] = x;
        A[p * n + k] = x;
        A[k * n + q] = y;
        A[q * n + k] = y;
      }
      A[p * n + p] = app - (t * apq);
      A[q * n + q] = aqq + (t * apq);
      A[p * n + q] = 0.0;
      A[q * n + p] = 0.0;
      for (int k = 0; k < n; ++k) {
        const double vkp = V[k * n + p], vkq = V[k * n + q];
        V[k * n + p] = (c * vkp) - (s * vkq);
        V[k * n + q] = (s * vkp) + (c * vkq);
      }
    }
  }
}

/* M^T M -> the four eigenvectors of the smallest eigenvalues, vv[0] the smallest (eigenvalue k has rank = the number of
 * eigenvalues below it; equal ones: those with a lower index). */
static inline int orc_epnp_null4(const double* alphas, const double* uv, int n, double* vv) {
  /* M^T M (12 x 12): two rows per point, [a_j, 0, -a_j u] and [0, a_j, -a_j v] for j = 0..3 */
  double MtM[144], evals[12], esub[12];
  for (int k = 0; k < 144; ++k) MtM[k] = 0.0;
  for (int i = 0; i < n; ++i) {
    double r1[12], r2[12];
    for (int j = 0; j < 4; ++j) {
      const double a = alphas[4 * i + j];
      r1[3 * j] = a;
      r1[3 * j + 1] = 0.0;
      r1[3 * j + 2] = -(a * uv[2 * i]);
      r2[3 * j] = 0.0;
      r2[3 * j + 1] = a;
      r2[3 * j + 2] = -(a * uv[2 * i + 1]);
    }
    for (int r = 0; r < 12; ++r)
      for (int c = 0; c < 12; ++c) MtM[12 * r + c] = (MtM[12 * r + c] + (r1[r] * r1[c])) + (r2[r] * r2[c]);
  }
  if (!orc_symeig12(MtM, evals, esub)) return 0; /* the columns of MtM are the eigenvectors now */
  for (int k = 0; k < 12; ++k) {
    int rank = 0;
    for (int j = 0; j < 12; ++j)
      if (evals[j] < evals[k] || (evals[j] == evals[k] && j < k)) rank++;
    if (rank < 4)
      for (int j = 0; j < 12; ++j) vv[12 * rank + j] = MtM[12 * j + k];
  }
  return 1;
}

/* f, p: n rows of 3 (bearings in the camera, points in the world), 5 <= n <= ORC_EPNP_MAXN.
 * -> R, t: pose of the camera in the world (points map by R^T (p - t)), as pyopengv returns it.  0 on failure. */
static inline int orc_epnp(const double* f, const double* p, int n, double* R, double* t) {
  double uv[2 * ORC_EPNP_MAXN], cw[12], alphas[4 * ORC_EPNP_MAXN], vv[48];
  if (!orc_epnp_front(f, p, n, uv, cw, alphas)) return 0;
  if (!orc_epnp_null4(alphas, uv, n, vv)) return 0;
  return orc_epnp_back(p, n, uv, cw, alphas, vv, R, t);
}
/* @oracle-only: end */
