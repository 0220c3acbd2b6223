// Device-side EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009) for the central absolute-pose RANSAC with algorithm
// "EPNP" (omnistereo/pose_est_tools.py:697, :915: OpenGV solves 6-point samples with EPnP): control points,
// barycentric coordinates, [the null space of M^T M: epnp_eig12_reg.h], the three beta initialisations with five
// Gauss-Newton steps each, absolute orientation, the candidate with the smallest reprojection error; plus the sampler
// of k distinct indices.  One lane per hypothesis, every array index a compile-time constant after unrolling.
// GENERATED from oracle/epnp_core.h by tests/gen_device_headers.py (the oracle's text minus its "oracle only" section, device
// prefixes): both sides evaluate the same operations in the same order; tests/test_abi.py checks the two files.
#pragma once
#include <hip/hip_runtime.h>
#include "ransac_core.h"

#define SV_EPNP_MAXN 8
#define SV_JACOBI_MAXN 12

/* Cyclic Jacobi on a symmetric n x n matrix A (row-major, destroyed: its diagonal ends as the eigenvalues), n <=
 * SV_JACOBI_MAXN; V (n x n) receives the eigenvectors as columns.  The classic symmetric update: a rotation in the
 * (p, q) plane changes rows / columns p and q only -- a'kp = c akp - s akq, a'kq = s akp + c akq for k != p, q
 * (mirrored), a'pp = app - t apq, a'qq = aqq + t apq, a'pq = 0 exactly.  The iterations of each inner loop touch
 * disjoint elements: all their loads come before the stores. */
__device__ static void sv_jacobi_sym(double* A, int n, double* V) {
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
        double xp[SV_JACOBI_MAXN], xq[SV_JACOBI_MAXN], yp[SV_JACOBI_MAXN], yq[SV_JACOBI_MAXN];
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
__device__ static int sv_lsq_small(const double* A, int lda, const double* b, int m, int k, double* x) {
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
__device__ static double sv_epnp_pose_from_betas(const double* betas, const double* vv /*[4][12]*/, const double* alphas,
                                              const double* pw, const double* uv, int n, double* Rcw, double* tcw) {
  double ccs[12], pcs[3 * SV_EPNP_MAXN];
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
  sv_jacobi_sym(S, 3, V);
  /* eigenvalues descending, with their eigenvectors (columns of V): a three-element sorting network on the values
   * instead of runtime indices into S and V (same comparisons, same outcome; keeps everything in registers) */
  double e0 = S[0], e1 = S[4], e2 = S[8];
  double v0[3] = {V[0], V[3], V[6]}, v1[3] = {V[1], V[4], V[7]}, v2[3] = {V[2], V[5], V[8]};
#define SV_CSWAP(ea, va, eb, vb)                   \
  if (ea < eb) {                                   \
    double t_ = ea; ea = eb; eb = t_;              \
    t_ = va[0]; va[0] = vb[0]; vb[0] = t_;         \
    t_ = va[1]; va[1] = vb[1]; vb[1] = t_;         \
    t_ = va[2]; va[2] = vb[2]; vb[2] = t_;         \
  }
  SV_CSWAP(e0, v0, e1, v1)
  SV_CSWAP(e1, v1, e2, v2)
  SV_CSWAP(e0, v0, e1, v1)
#undef SV_CSWAP
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

/* EPnP, first half: f, p: n rows of 3 (bearings in the camera, points in the world), 5 <= n <= SV_EPNP_MAXN ->
 * normalised image coordinates uv, control points cw, barycentric coordinates alphas.  0 on failure. */
__device__ static int sv_epnp_front(const double* f, const double* p, int n, double* uv, double* cw, double* alphas) {
  if (n < 5 || n > SV_EPNP_MAXN) return 0; /* 4 points leave a 4-dimensional null space: not handled */
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
  sv_jacobi_sym(C, 3, E);
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
__device__ static int sv_epnp_back(const double* p, int n, const double* uv, const double* cw, const double* alphas,
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
#define SV_D(a, b) (((dv[a][0] * dv[b][0]) + (dv[a][1] * dv[b][1])) + (dv[a][2] * dv[b][2]))
    L[10 * j + 0] = SV_D(0, 0);
    L[10 * j + 1] = 2.0 * SV_D(0, 1);
    L[10 * j + 2] = SV_D(1, 1);
    L[10 * j + 3] = 2.0 * SV_D(0, 2);
    L[10 * j + 4] = 2.0 * SV_D(1, 2);
    L[10 * j + 5] = SV_D(2, 2);
    L[10 * j + 6] = 2.0 * SV_D(0, 3);
    L[10 * j + 7] = 2.0 * SV_D(1, 3);
    L[10 * j + 8] = 2.0 * SV_D(2, 3);
    L[10 * j + 9] = SV_D(3, 3);
#undef SV_D
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
      ok = sv_lsq_small(A, 5, rho, 6, 4, x);
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
      if (variant == 1) ok = sv_lsq_small(A, 5, rho, 6, 3, x); /* (the system size stays a constant at each call) */
      else ok = sv_lsq_small(A, 5, rho, 6, 5, x);
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
      if (!sv_lsq_small(J, 5, r, 6, 4, dx)) break;
      #pragma unroll
      for (int a = 0; a < 4; ++a) betas[a] = betas[a] + dx[a];
    }
    double Rc[9], tc[3];
    const double err = sv_epnp_pose_from_betas(betas, vv, alphas, p, uv, n, Rc, tc);
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
__device__ static int sv_sample_distinct(int32_t n, int k, uint64_t seed, uint64_t it, int32_t* s) {
  if (n < k || k > 9) return 0;
  int32_t taken[9]; /* ascending */
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    if (j >= k) break;
    int32_t v = (int32_t)sv_below(sv_mix64(seed, it, (uint64_t)j), (uint32_t)(n - j));
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


#include "epnp_eig12_reg.h"
