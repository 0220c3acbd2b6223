// The 12 x 12 eigen-solver of EPnP on the device and the solver's driver: the ONE part of the EPnP restatement that is
// not the oracle's text (csrc/epnp_core.h is generated from oracle/epnp_core.h; this file is hand-written and included at its
// end).  Same operations in the same order as orc_jacobi12_rr / orc_epnp_null4 -- the GPU parity tests compare every
// hypothesis bit for bit -- but worded for registers: upper triangle + eigenvector matrix in 78 + 144 doubles, the eleven
// rounds of a sweep unrolled by a fold so that every index is a compile-time constant.
#pragma once
#include <type_traits>
#include <utility>

/* The 12 x 12 symmetric eigen-problem of EPnP (M^T M): Jacobi rotations in round-robin order (oracle/epnp_core.h,
 * orc_jacobi12_rr: the same operations in the same order).  Pair i of round r is (r, 11) for i = 0 and
 * ((r + i) mod 11, (r - i) mod 11) otherwise, smaller index first. */
#define SV_JACOBI12_TOL 1e-26
__device__ constexpr int sv_rr_first(int idx) {
  const int r = idx / 6, i = idx % 6;
  const int a = i == 0 ? r : (r + i) % 11, b = i == 0 ? 11 : (r - i + 11) % 11;
  return a < b ? a : b;
}
__device__ constexpr int sv_rr_second(int idx) {
  const int r = idx / 6, i = idx % 6;
  const int a = i == 0 ? r : (r + i) % 11, b = i == 0 ? 11 : (r - i + 11) % 11;
  return a < b ? b : a;
}
/* index of element (i, j) of a symmetric 12 x 12 matrix kept as its upper triangle (78 entries) */
__device__ constexpr int sv_tri(int i, int j) { return i <= j ? i * 12 - i * (i - 1) / 2 + (j - i) : j * 12 - j * (j - 1) / 2 + (i - j); }
template <typename F, int... R>
__device__ __forceinline__ void sv_for_each_round(F& f, std::integer_sequence<int, R...>) {
  (f(std::integral_constant<int, R>{}), ...);
}

/* EVERYTHING IN REGISTERS: the symmetric matrix as its upper triangle (a, 78 doubles; built from the barycentric
 * coordinates: two rows of M per point, [a_j, 0, -a_j u] and [0, a_j, -a_j v]) and the eigenvector matrix V (144
 * doubles) -- a wave that has a SIMD to itself owns 512 registers per lane.  Every index is a compile-time constant: the
 * 11 rounds of a sweep are unrolled by a fold.  The six rotations of a round have disjoint index pairs, so their angles
 * -- two divisions and two square roots in a chain, the longest dependency of the solver -- are evaluated side by side
 * before the rotations are applied one after the other.  -> vv: the four eigenvectors of the smallest eigenvalues, vv[0]
 * the smallest (eigenvalue k has rank = the number of eigenvalues below it; equal ones: those with a lower index). */
__device__ static void sv_epnp_null4_reg(const double* alphas, const double* uv, int n, double* vv) {
  double a[78], V[144];
#pragma unroll
  for (int k = 0; k < 78; ++k) a[k] = 0.0;
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
      for (int c = r; c < 12; ++c) a[sv_tri(r, c)] = (a[sv_tri(r, c)] + (r1[r] * r1[c])) + (r2[r] * r2[c]);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int j = 0; j < 12; ++j) V[i * 12 + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
#pragma unroll
    for (int p = 0; p < 12; ++p) {
      diag = diag + (a[sv_tri(p, p)] * a[sv_tri(p, p)]);
#pragma unroll
      for (int q = p + 1; q < 12; ++q) off = off + (a[sv_tri(p, q)] * a[sv_tri(p, q)]);
    }
    if (!(off > (SV_JACOBI12_TOL * diag))) break;
    auto round = [&](auto r_tag) __attribute__((always_inline)) {
      constexpr int R = decltype(r_tag)::value;
      double tt[6], cc[6], ss[6];
      bool on[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {  // the six angle chains of the round, independent of each other
        const int P = sv_rr_first(6 * R + i), Q = sv_rr_second(6 * R + i);
        const double apq = a[sv_tri(P, Q)], app = a[sv_tri(P, P)], aqq = a[sv_tri(Q, Q)];
        on[i] = apq != 0.0;
        const double theta = (aqq - app) / (2.0 * apq);
        const double at = theta < 0.0 ? -theta : theta;
        tt[i] = (theta < 0.0 ? -1.0 : 1.0) / (at + sqrt((theta * theta) + 1.0));
        cc[i] = 1.0 / sqrt((tt[i] * tt[i]) + 1.0);
        ss[i] = tt[i] * cc[i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {  // the rotations, one after the other
        const int P = sv_rr_first(6 * R + i), Q = sv_rr_second(6 * R + i);
        if (!on[i]) continue;
        const double c = cc[i], s = ss[i], t = tt[i];
        const double apq = a[sv_tri(P, Q)], app = a[sv_tri(P, P)], aqq = a[sv_tri(Q, Q)];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          if (k == P || k == Q) continue;
          const double akp = a[sv_tri(k, P)], akq = a[sv_tri(k, Q)];
          a[sv_tri(k, P)] = (c * akp) - (s * akq);
          a[sv_tri(k, Q)] = (s * akp) + (c * akq);
        }
        a[sv_tri(P, P)] = app - (t * apq);
        a[sv_tri(Q, Q)] = aqq + (t * apq);
        a[sv_tri(P, Q)] = 0.0;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const double vkp = V[k * 12 + P], vkq = V[k * 12 + Q];
          V[k * 12 + P] = (c * vkp) - (s * vkq);
          V[k * 12 + Q] = (s * vkp) + (c * vkq);
        }
      }
    };
    sv_for_each_round(round, std::make_integer_sequence<int, 11>{});
  }
  int rank[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    int rk = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const double ej = a[sv_tri(j, j)], ek = a[sv_tri(k, k)];
      rk += (ej < ek || (ej == ek && j < k)) ? 1 : 0;
    }
    rank[k] = rk;
  }
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 12; ++k) v = rank[k] == r4 ? V[12 * j + k] : v;
      vv[12 * r4 + j] = v;
    }
}

/* The whole solver on one lane. */
__device__ static int sv_epnp(const double* f, const double* p, int n, double* R, double* t) {
  double uv[2 * SV_EPNP_MAXN], cw[12], alphas[4 * SV_EPNP_MAXN], vv[48];
  if (!sv_epnp_front(f, p, n, uv, cw, alphas)) return 0;
  sv_epnp_null4_reg(alphas, uv, n, vv);
  return sv_epnp_back(p, n, uv, cw, alphas, vv, R, t);
}

