// The 12 x 12 eigen-solver of EPnP on the device and the solver's driver: the ONE part of the EPnP restatement that is
// not the oracle's text (csrc/epnp_core.h is generated from oracle/epnp_core.h; this file is hand-written and included at its
// end).  Same operations in the same order as orc_symeig12 / orc_epnp_null4 -- the GPU parity tests compare every
// hypothesis bit for bit -- but worded for registers: the 12 x 12 matrix (which becomes the eigenvector matrix), the
// diagonal and the sub-diagonal in 144 + 24 doubles, every loop unrolled so that every index is a compile-time constant.
// (Until round 2 this was a round-robin Jacobi solver, ~7 x the arithmetic: 1.52 ms per 128 pairs x 2000 hypotheses
// against the figure in DESIGN.md section 10 now; the oracle keeps its loop form, orc_jacobi12_rr, as a cross-check.)
#pragma once
#include <type_traits>
#include <utility>
template <typename F, int... R>
__device__ __forceinline__ void sv_for_each_const(F& f, std::integer_sequence<int, R...>) {
  (f(std::integral_constant<int, R>{}), ...);
}

/* ---- Householder tridiagonalisation + QL with implicit shifts, register-resident --------------------------------------
 * The same operations in the same order as orc_symeig12 (oracle/epnp_core.h: the EISPACK tred2 / tql2 pair restated;
 * explicit fused multiply-adds where the oracle writes fma()), worded so that every array index is a compile-time
 * constant after unrolling: the loops of the reduction have static bounds; in the QL part the two data-dependent indices
 * -- m, where the matrix splits, and the start of the rotation chain -- become per-lane predicates over static ranges
 * ("rotation i runs if i < m"), and d[m] / e[m] are read and written through select chains.  ~1/7 of the arithmetic of
 * the Jacobi sweeps; lanes of a wave converge after different numbers of QL iterations and the wave pays for the
 * slowest, which still leaves it several times ahead.  A: symmetric on entry, its COLUMNS are the eigenvectors on exit,
 * d the eigenvalues (unordered).  Returns 0 if an eigenvalue needs more than 30 iterations. */
#define SV_EIG_EPS 1e-12
__device__ __forceinline__ static int sv_symeig12_reg(double (&A)[144], double (&d)[12], double (&e)[12]) {
  constexpr int n = 12;
  // (the outer loops are unrolled by a fold: a constexpr index makes the triangular inner loops' bounds constants at once)
  auto reduce_step = [&](auto i_tag) __attribute__((always_inline)) {
    constexpr int i = n - 1 - decltype(i_tag)::value;
    constexpr int l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0) {
#pragma unroll
      for (int k = 0; k <= l; ++k) scale = scale + fabs(A[i * n + k]);
      if (scale == 0.0) {
        e[i] = A[i * n + l];
      } else {
#pragma unroll
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
#pragma unroll
        for (int j = 0; j <= l; ++j) {
          A[j * n + i] = A[i * n + j] / h;
          g = 0.0;
#pragma unroll
          for (int k = 0; k <= j; ++k) g = fma(A[j * n + k], A[i * n + k], g);
#pragma unroll
          for (int k = j + 1; k <= l; ++k) g = fma(A[k * n + j], A[i * n + k], g);
          e[j] = g / h;
          f = f + (e[j] * A[i * n + j]);
        }
        const double hh = f / (h + h);
#pragma unroll
        for (int j = 0; j <= l; ++j) {
          f = A[i * n + j];
          g = e[j] - (hh * f);
          e[j] = g;
#pragma unroll
          for (int k = 0; k <= j; ++k) A[j * n + k] = A[j * n + k] - fma(f, e[k], g * A[i * n + k]);
        }
      }
    } else {
      e[i] = A[i * n + l];
    }
    d[i] = h;
  };
  sv_for_each_const(reduce_step, std::make_integer_sequence<int, n - 1>{});
  d[0] = 0.0;
  e[0] = 0.0;
  auto accumulate_step = [&](auto i_tag) __attribute__((always_inline)) {
    constexpr int i = decltype(i_tag)::value;
    constexpr int l = i - 1;
    if (d[i] != 0.0) {
#pragma unroll
      for (int j = 0; j <= l; ++j) {
        double g = 0.0;
#pragma unroll
        for (int k = 0; k <= l; ++k) g = fma(A[i * n + k], A[k * n + j], g);
#pragma unroll
        for (int k = 0; k <= l; ++k) A[k * n + j] = fma(-g, A[k * n + i], A[k * n + j]);
      }
    }
    d[i] = A[i * n + i];
    A[i * n + i] = 1.0;
#pragma unroll
    for (int j = 0; j <= l; ++j) {
      A[j * n + i] = 0.0;
      A[i * n + j] = 0.0;
    }
  };
  sv_for_each_const(accumulate_step, std::make_integer_sequence<int, n>{});
#pragma unroll
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  /* QL: eigenvalue after eigenvalue.  Written once, for "l = 0": after each eigenvalue the arrays d, e and the columns
   * of A are shifted one place to the left (cyclically), so that the active block always starts at position 0 and ends at
   * len - 1; after the twelfth shift everything is back in its original place.  Same arithmetic as the oracle's loop over
   * l -- only the place of the operands moves -- and one copy of the rotation chain (~20 KB of code) that every wave of
   * the chip runs, instead of twelve specialised ones (the waves drift apart with their iteration counts and would evict
   * each other's part of the instruction cache: measured 5.1 against 1.5 ms). */
  int ok = 1;
  for (int len = n; len >= 1; --len) {
    int iter = 0;
    while (ok) {
      /* m = the first position whose sub-diagonal element is negligible (len - 1 if none) */
      int m = len - 1;
#pragma unroll
      for (int mm = n - 2; mm >= 0; --mm) {
        const double dd = fabs(d[mm]) + fabs(d[mm + 1]);
        if (mm <= len - 2 && fabs(e[mm]) <= (SV_EIG_EPS * dd)) m = mm;
      }
      if (m == 0) break;
      if (iter++ == 30) {
        ok = 0;
        break;
      }
      double dm = d[n - 1];  // d[m]
#pragma unroll
      for (int mm = n - 2; mm > 0; --mm) dm = (m == mm) ? d[mm] : dm;
      double g = (d[1] - d[0]) / (2.0 * e[0]);
      double r = sqrt((g * g) + 1.0);
      g = (dm - d[0]) + (e[0] / (g + (g >= 0.0 ? r : -r)));
      double s = 1.0, c = 1.0, p = 0.0;
      bool broke = false;
#pragma unroll
      for (int i = n - 2; i >= 0; --i) {
        if (i < m && !broke) {
          double f = s * e[i];
          const double b = c * e[i];
          r = sqrt((f * f) + (g * g));
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] = d[i + 1] - p;
#pragma unroll
            for (int mm = 0; mm < n; ++mm)
              if (m == mm) e[mm] = 0.0;
            broke = true;
          } else {
            s = f / r;
            c = g / r;
            g = d[i + 1] - p;
            r = ((d[i] - g) * s) + ((2.0 * c) * b);
            p = s * r;
            d[i + 1] = g + p;
            g = (c * r) - b;
#pragma unroll
            for (int k = 0; k < n; ++k) {
              f = A[k * n + i + 1];
              const double zi = A[k * n + i];
              A[k * n + i + 1] = fma(s, zi, c * f);
              A[k * n + i] = fma(c, zi, -(s * f));
            }
          }
        }
      }
      if (!broke) {
        d[0] = d[0] - p;
        e[0] = g;
#pragma unroll
        for (int mm = 0; mm < n; ++mm)
          if (m == mm) e[mm] = 0.0;
      }
    }
    { /* position 0 is done: everything one place to the left, position 0 to the end */
      const double d0 = d[0], e0 = e[0];
#pragma unroll
      for (int q = 0; q < n - 1; ++q) {
        d[q] = d[q + 1];
        e[q] = e[q + 1];
      }
      d[n - 1] = d0;
      e[n - 1] = e0;
#pragma unroll
      for (int k = 0; k < n; ++k) {
        const double a0 = A[k * n];
#pragma unroll
        for (int q = 0; q < n - 1; ++q) A[k * n + q] = A[k * n + q + 1];
        A[k * n + n - 1] = a0;
      }
    }
  }
  return ok;
}

/* M^T M of the sample (two rows of M per point, [a_j, 0, -a_j u] and [0, a_j, -a_j v]) -> the four eigenvectors of the
 * smallest eigenvalues, vv[0] the smallest (eigenvalue k has rank = the number of eigenvalues below it; equal ones:
 * those with a lower index); orc_epnp_null4's operations.  Returns 0 if the eigen-solver gives up. */
__device__ static int sv_epnp_null4_ql(const double* alphas, const double* uv, int n, double* vv) {
  double A[144], d[12], e[12];
#pragma unroll
  for (int k = 0; k < 144; ++k) A[k] = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double r1[12], r2[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double al = alphas[4 * i + j];
      r1[3 * j] = al;
      r1[3 * j + 1] = 0.0;
      r1[3 * j + 2] = -(al * uv[2 * i]);
      r2[3 * j] = 0.0;
      r2[3 * j + 1] = al;
      r2[3 * j + 2] = -(al * uv[2 * i + 1]);
    }
#pragma unroll
    for (int r = 0; r < 12; ++r)
#pragma unroll
      for (int c = 0; c < 12; ++c) A[12 * r + c] = (A[12 * r + c] + (r1[r] * r1[c])) + (r2[r] * r2[c]);
  }
  if (!sv_symeig12_reg(A, d, e)) return 0;
  int rank[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    int rk = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) rk += (d[j] < d[k] || (d[j] == d[k] && j < k)) ? 1 : 0;
    rank[k] = rk;
  }
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 12; ++k) v = rank[k] == r4 ? A[12 * j + k] : v;
      vv[12 * r4 + j] = v;
    }
  return 1;
}

/* The whole solver on one lane. */
__device__ static int sv_epnp(const double* f, const double* p, int n, double* R, double* t) {
  double uv[2 * SV_EPNP_MAXN], cw[12], alphas[4 * SV_EPNP_MAXN], vv[48];
  if (!sv_epnp_front(f, p, n, uv, cw, alphas)) return 0;
  if (!sv_epnp_null4_ql(alphas, uv, n, vv)) return 0;
  return sv_epnp_back(p, n, uv, cw, alphas, vv, R, t);
}
