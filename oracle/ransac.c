/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product
 * path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * CPU restatement of the 3D-2D absolute-pose RANSAC + non-linear refinement stage.
 *
 * The arithmetic lives in OpenGV (pyopengv, fork ubuntuslave/opengv branch
 * non_central-python, commit unpinned: reference README.md:253-258), a third-party dependency
 * that is NOT under the reference tree.  The restatement follows the published algorithms and
 * is anchored on the reference's own call sites and on its in-repo restatement of the score:
 *   omnistereo/pose_est_tools.py:785   absolute_pose_noncentral_ransac(b, cam, p, offsets, rots, thr, iters)
 *   omnistereo/pose_est_tools.py:915   absolute_pose_ransac(b, p, "EPNP"|"KNEIP"|..., thr, iters)
 *   omnistereo/pose_est_tools.py:830/:937  *_optimize_nonlinear(...)
 *   omnistereo/pose_est_tools.py:150-203   get_selected_distances_to_model: score = 1 - f . normalize(R^T (p - t))
 *   omnistereo/pose_est_tools.py:181-185   non-central form: R_c^T (R^T (p - t) - o_c)
 *   omnistereo/pose_est_tools.py:672-720   threshold 1 - cos(5 deg), iteration budget
 * Published algorithms restated: Kneip, Scaramuzza, Siegwart, "A novel parametrization of the
 * P3P problem", CVPR 2011 (minimal solver); Fischler & Bolles RANSAC with the adaptive stop
 * k = log(1 - 0.99) / log(1 - w^s) as in OpenGV's sac::Ransac; Levenberg-Marquardt on
 * (t, Cayley(R)) with residual 1 - f . f_hat and forward-difference Jacobian as in OpenGV's
 * optimize_nonlinear.
 * Deliberate, documented deviations (DESIGN.md "RANSAC"): (1) OpenGV seeds its sampler from the
 * clock, so the reference does not reproduce its own inlier sets; the sampler here is a
 * counter-based hash of (seed, iteration, draw) shared with the HIP path.  (2) The non-central
 * minimal solver draws its three solve points from ONE camera of the rig (central P3P in that
 * camera, moved to the body frame) and disambiguates/scores non-centrally, instead of OpenGV's
 * gP3P Groebner template, which cannot be restated from the reference tree.  (3) Real quartic
 * roots only (OpenGV takes real parts of complex roots; those candidates never survive the
 * 4th-point check).
 * Parity status: UNPINNED against OpenGV binaries (no golden vectors exist in the reference);
 * pinned by noise-free known-pose tests (tests/test_oracle_ransac.py) and, for the score, by a
 * fixture generated from the reference's own get_selected_distances_to_model
 * (tests/golden/make_fixtures.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ransac_core.h"
#include "epnp_core.h"
#include "gp3p_core.h"
#include "relpose_core.h"

/* ---- scoring of all points under one pose --------------------------------------------- */
void orc_score_points(const double* f, const double* p, const int32_t* cam, const double* cam_off,
                      const double* cam_rot, int32_t n, const double* T, double* scores) {
  double R[9], t[3];
  orc_T_to_Rt(T, R, t);
  const double zero3[3] = {0, 0, 0};
  const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < n; ++i) {
    int c = cam ? cam[i] : 0;
    const double* o = cam ? cam_off + 3 * c : zero3;
    const double* Rc = cam ? cam_rot + 9 * c : eye;
    scores[i] = orc_score(R, t, f + 3 * i, p + 3 * i, o, Rc);
  }
}

/* ---- RANSAC ------------------------------------------------------------------------------ */
/* cam == NULL -> central problem (one camera at the body origin).
 * T_out: 3x4 row-major [R|t].  Returns 0 on success, 1 if no valid hypothesis was found. */
int32_t orc_ransac_abs_pose(const double* f, const double* p, const int32_t* cam, const double* cam_off,
                            const double* cam_rot, int32_t n, int32_t ncam, double thr, int32_t max_iter,
                            int32_t adaptive, uint64_t seed, double* T_out, uint8_t* inlier_mask,
                            int32_t* n_inliers, int32_t* best_iter, int32_t* iters_used,
                            int32_t* counts_out /* [max_iter] or NULL */) {
  const double zero3[3] = {0, 0, 0};
  const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  /* `adaptive` bit 1 selects the EPnP hypothesis generator (central problems only): 6-point samples solved by
   * orc_epnp, as OpenGV's AbsolutePoseSacProblem does for algorithm EPNP (pose_est_tools.py:697, :915). */
  const int use_epnp = ((adaptive >> 1) & 1) && !cam;
  /* bit 2 selects the generalised P3P generator (oracle/gp3p_core.h): four correspondences out of ALL cameras, as
   * OpenGV's non-central problem does ("will ALWAYS use GP3P", pose_est_tools.py:696); without it the three solve points
   * come from one camera (central P3P moved to the body frame). */
  const int use_gp3p = ((adaptive >> 2) & 1) && !use_epnp;
  /* bit 3: TWOPT (central problems): 2-point samples, translation only, rotation = identity */
  const int use_twopt = ((adaptive >> 3) & 1) && !cam && !use_epnp && !use_gp3p;
  adaptive &= 1;
  if (!cam) {
    ncam = 1;
    cam_off = zero3;
    cam_rot = eye;
  }
  /* stable partition of point indices by camera */
  int32_t* perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  int32_t* cstart = (int32_t*)calloc((size_t)ncam + 1, sizeof(int32_t));
  int32_t* ccount = (int32_t*)calloc((size_t)ncam, sizeof(int32_t));
  for (int i = 0; i < n; ++i) ccount[cam ? cam[i] : 0]++;
  for (int c = 0; c < ncam; ++c) cstart[c + 1] = cstart[c] + ccount[c];
  {
    int32_t* fill = (int32_t*)calloc((size_t)ncam, sizeof(int32_t));
    for (int i = 0; i < n; ++i) {
      int c = cam ? cam[i] : 0;
      perm[cstart[c] + fill[c]++] = i;
    }
    free(fill);
  }

  int best_count = -1, best_it = -1;
  double best_R[9], best_t[3];
  double base = 1.0; /* (1 - w^4) of the best model so far, see orc_ransac_continue */
  int iterations = 0, used = 0;
  for (int it = 0; it < max_iter; ++it) {
    if (adaptive && iterations > 0 && !orc_ransac_continue(base, iterations)) break;
    used = it + 1;
    double R[9], t[3];
    int ok;
    if (use_epnp) {
      int32_t s6[6];
      double f6[18], p6[18];
      ok = orc_sample_distinct(n, 6, seed, (uint64_t)it, s6);
      if (ok) {
        for (int k = 0; k < 6; ++k)
          for (int c = 0; c < 3; ++c) {
            f6[3 * k + c] = f[3 * s6[k] + c];
            p6[3 * k + c] = p[3 * s6[k] + c];
          }
        ok = orc_epnp(f6, p6, 6, R, t);
      }
    } else if (use_twopt) {
      ok = orc_hypothesis_twopt(f, p, n, seed, (uint64_t)it, R, t);
    } else if (use_gp3p) {
      ok = orc_hypothesis_gp3p(f, p, cam, cam_off, cam_rot, n, seed, (uint64_t)it, R, t);
    } else {
      ok = orc_hypothesis(f, p, cam, cam_off, cam_rot, n, perm, cstart, ccount, seed, (uint64_t)it, R, t);
    }
    if (counts_out) counts_out[it] = -1;
    if (!ok) continue; /* failed solve: skipped, does not count as an iteration */
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
      int c = cam ? cam[i] : 0;
      if (orc_score(R, t, f + 3 * i, p + 3 * i, cam_off + 3 * c, cam_rot + 9 * c) < thr) cnt++;
    }
    if (counts_out) counts_out[it] = cnt;
    if (cnt > best_count) {
      best_count = cnt;
      best_it = it;
      memcpy(best_R, R, sizeof(R));
      memcpy(best_t, t, sizeof(t));
      base = use_epnp ? orc_adaptive_base6(cnt, n) : (use_twopt ? orc_adaptive_base2(cnt, n) : orc_adaptive_base(cnt, n));
    }
    iterations++;
  }
  for (int it = used; counts_out && it < max_iter; ++it) counts_out[it] = -2; /* never drawn */

  int status = 0;
  if (best_it < 0) {
    status = 1;
    memcpy(best_R, eye, sizeof(eye));
    memcpy(best_t, zero3, sizeof(zero3));
    best_count = 0;
  }
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    int c = cam ? cam[i] : 0;
    uint8_t in = (status == 0) && (orc_score(best_R, best_t, f + 3 * i, p + 3 * i, cam_off + 3 * c, cam_rot + 9 * c) < thr);
    inlier_mask[i] = in;
    cnt += in;
  }
  orc_Rt_to_T(best_R, best_t, T_out);
  *n_inliers = cnt;
  *best_iter = best_it;
  *iters_used = used;
  free(perm);
  free(cstart);
  free(ccount);
  return status;
}

/* One hypothesis (for unit tests of the minimal solver and the sampler). */
int32_t orc_hypothesis_once(const double* f, const double* p, const int32_t* cam, const double* cam_off,
                            const double* cam_rot, int32_t n, int32_t ncam, uint64_t seed, int32_t it,
                            double* T_out, int32_t* sample4) {
  const double zero3[3] = {0, 0, 0};
  const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (!cam) {
    ncam = 1;
    cam_off = zero3;
    cam_rot = eye;
  }
  int32_t* perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  int32_t* cstart = (int32_t*)calloc((size_t)ncam + 1, sizeof(int32_t));
  int32_t* ccount = (int32_t*)calloc((size_t)ncam, sizeof(int32_t));
  for (int i = 0; i < n; ++i) ccount[cam ? cam[i] : 0]++;
  for (int c = 0; c < ncam; ++c) cstart[c + 1] = cstart[c] + ccount[c];
  int32_t* fill = (int32_t*)calloc((size_t)ncam, sizeof(int32_t));
  for (int i = 0; i < n; ++i) {
    int c = cam ? cam[i] : 0;
    perm[cstart[c] + fill[c]++] = i;
  }
  free(fill);
  double R[9], t[3];
  int32_t s[4] = {-1, -1, -1, -1};
  int ok = orc_sample4(cam, n, perm, cstart, ccount, seed, (uint64_t)it, s);
  if (sample4) memcpy(sample4, s, sizeof(s));
  if (ok) ok = orc_hypothesis(f, p, cam, cam_off, cam_rot, n, perm, cstart, ccount, seed, (uint64_t)it, R, t);
  if (ok) orc_Rt_to_T(R, t, T_out);
  free(perm);
  free(cstart);
  free(ccount);
  return ok;
}

/* All solutions of the central P3P for three correspondences (unit tests). */
int32_t orc_p3p_kneip(const double* f3x3, const double* p3x3, double* R_out /*[4][9]*/, double* C_out /*[4][3]*/) {
  return orc_p3p(f3x3, f3x3 + 3, f3x3 + 6, p3x3, p3x3 + 3, p3x3 + 6, R_out, C_out);
}

int32_t orc_quartic_real_roots(const double* a5, double* roots4) { return orc_quartic(a5, roots4); }

/* ---- non-linear refinement ------------------------------------------------------------------ */
/* Levenberg-Marquardt on x = (t, cayley(R)), residual_i = 1 - f_i . f_hat_i(x), forward
 * differences.  idx == NULL -> all n points, else the m listed points.  T_io: 3x4 in/out. */
int32_t orc_refine_abs_pose(const double* f, const double* p, const int32_t* cam, const double* cam_off,
                            const double* cam_rot, int32_t n, const int32_t* idx, int32_t m, double* T_io,
                            int32_t max_lm_iter, double* final_cost, int32_t* lm_iters) {
  const double zero3[3] = {0, 0, 0};
  const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (!cam) {
    cam_off = zero3;
    cam_rot = eye;
  }
  int cnt = idx ? m : n;
  double R[9], t[3], x[6];
  orc_T_to_Rt(T_io, R, t);
  x[0] = t[0];
  x[1] = t[1];
  x[2] = t[2];
  orc_rot2cayley(R, x + 3);
  double lambda = ORC_LM_LAMBDA0;
  double cost = 0.0;
  int it_done = 0;
  /* Sums over correspondences are DEFINED as 256-way interleaved partial sums (term q goes to
   * partial q mod 256, in increasing q) combined by orc_tree_sum256; the HIP path evaluates the
   * same order (one partial per lane, wave shuffle tree, four wave partials), which makes the
   * LM trajectory reproducible bit for bit. */
  double(*part)[256] = (double(*)[256])malloc(sizeof(double) * 28 * 256);
  for (int it = 0; it < max_lm_iter; ++it) {
    double A[21], g[6];
    for (int a = 0; a < 28; ++a)
      for (int l = 0; l < 256; ++l) part[a][l] = 0.0;
    for (int q = 0; q < cnt; ++q) {
      int i = idx ? idx[q] : q;
      int c = cam ? cam[i] : 0;
      int l = q & 255;
      double r, J[6], Hq[21];
      orc_residual_jac(x, f + 3 * i, p + 3 * i, cam_off + 3 * c, cam_rot + 9 * c, &r, J, Hq);
      part[27][l] += r * r;
      int a = 0;
      for (int u = 0; u < 6; ++u) {
        part[21 + u][l] += J[u] * r;
        for (int v = u; v < 6; ++v) {
          part[a][l] += (J[u] * J[v]) + Hq[a];
          a++;
        }
      }
    }
    cost = orc_tree_sum256(part[27]);
    for (int a = 0; a < 21; ++a) A[a] = orc_tree_sum256(part[a]);
    for (int a = 0; a < 6; ++a) g[a] = orc_tree_sum256(part[21 + a]);
    it_done = it;
    /* inner loop: raise lambda until the step reduces the cost */
    int accepted = 0, converged = 0;
    for (int tries = 0; tries < ORC_LM_MAX_TRIES; ++tries) {
      double dx[6], xn[6];
      if (!orc_solve_damped(A, g, lambda, dx)) {
        lambda *= 10.0;
        continue;
      }
      for (int u = 0; u < 6; ++u) xn[u] = x[u] + dx[u];
      {
        /* a step below the resolution we care about ends the refinement before it is evaluated */
        double dxn = 0.0, xnn = 0.0;
        for (int u = 0; u < 6; ++u) {
          dxn += dx[u] * dx[u];
          xnn += xn[u] * xn[u];
        }
        if (sqrt(dxn) <= ORC_LM_XTOL * (sqrt(xnn) + ORC_LM_XTOL)) {
          converged = 1;
          break;
        }
      }
      for (int l = 0; l < 256; ++l) part[0][l] = 0.0;
      for (int q = 0; q < cnt; ++q) {
        int i = idx ? idx[q] : q;
        int c = cam ? cam[i] : 0;
        double r = orc_residual(xn, f + 3 * i, p + 3 * i, cam_off + 3 * c, cam_rot + 9 * c);
        part[0][q & 255] += r * r;
      }
      const double cn = orc_tree_sum256(part[0]);
      if (cn < cost) {
        double dxn = 0.0, xnn = 0.0;
        for (int u = 0; u < 6; ++u) {
          dxn += dx[u] * dx[u];
          xnn += xn[u] * xn[u];
        }
        converged = ((cost - cn) <= ORC_LM_FTOL * cost);
        (void)dxn;
        (void)xnn;
        for (int u = 0; u < 6; ++u) x[u] = xn[u];
        cost = cn;
        lambda *= 0.1;
        if (lambda < 1e-15) lambda = 1e-15;
        accepted = 1;
        break;
      }
      lambda *= 10.0;
    }
    it_done = it + 1;
    if (!accepted || converged) break;
  }
  orc_cayley2rot(x + 3, R);
  t[0] = x[0];
  t[1] = x[1];
  t[2] = x[2];
  orc_Rt_to_T(R, t, T_io);
  free(part);
  if (final_cost) *final_cost = cost;
  if (lm_iters) *lm_iters = it_done;
  return 0;
}

/* EPnP on n (5..8) correspondences (unit tests): T_out = [R | t], pose of the camera in the world. */
int32_t orc_epnp_solve(const double* f, const double* p, int32_t n, double* T_out) {
  double R[9], t[3];
  const int ok = orc_epnp(f, p, n, R, t);
  if (ok) orc_Rt_to_T(R, t, T_out);
  return ok;
}

/* The 12 x 12 symmetric eigen-solver of EPnP on its own (unit tests): A [144] in / eigenvectors (columns) out, d [12]. */
int32_t orc_symeig12_solve(double* A, double* d) {
  double e[12];
  return orc_symeig12(A, d, e);
}

/* ---- 2D-2D relative-pose RANSAC (pyopengv.relative_pose_ransac, pose_est_tools.py:78) with the eight-point solver ----
 * T_out: 3x4 [R|t], pose of viewpoint 2 in frame 1, |t| = 1.  Same sequential semantics as orc_ransac_abs_pose
 * (strictly-better update, failed solves skipped, adaptive stop of sac::Ransac on 8-point samples). */
int32_t orc_ransac_rel_pose(const double* f1, const double* f2, int32_t n, int32_t algorithm, double thr, int32_t max_iter, int32_t adaptive,
                            uint64_t seed, double* T_out, uint8_t* inlier_mask, int32_t* n_inliers, int32_t* best_iter,
                            int32_t* iters_used, int32_t* counts_out) {
  int best_count = -1, best_it = -1;
  double best_R[9], best_t[3];
  double base = 1.0;
  int iterations = 0, used = 0;
  for (int it = 0; it < max_iter; ++it) {
    if (adaptive && iterations > 0 && !orc_ransac_continue(base, iterations)) break;
    used = it + 1;
    double R[9], t[3];
    const int ok = orc_rel_hypothesis(f1, f2, n, algorithm, seed, (uint64_t)it, R, t);
    if (counts_out) counts_out[it] = -1;
    if (!ok) continue;
    int cnt = 0;
    for (int i = 0; i < n; ++i)
      if (orc_rel_score(R, t, f1 + 3 * i, f2 + 3 * i) < thr) cnt++;
    if (counts_out) counts_out[it] = cnt;
    if (cnt > best_count) {
      best_count = cnt;
      best_it = it;
      memcpy(best_R, R, sizeof(R));
      memcpy(best_t, t, sizeof(t));
      base = orc_adaptive_base_k(cnt, n, algorithm == ORC_REL_SEVENPT ? 9 : 8);
    }
    iterations++;
  }
  for (int it = used; counts_out && it < max_iter; ++it) counts_out[it] = -2;
  int status = 0;
  if (best_it < 0) {
    status = 1;
    const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z[3] = {0, 0, 0};
    memcpy(best_R, eye, sizeof(eye));
    memcpy(best_t, z, sizeof(z));
  }
  orc_Rt_to_T(best_R, best_t, T_out);
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    const int in = status == 0 && orc_rel_score(best_R, best_t, f1 + 3 * i, f2 + 3 * i) < thr;
    inlier_mask[i] = (uint8_t)in;
    cnt += in;
  }
  *n_inliers = cnt;
  *best_iter = best_it;
  *iters_used = used;
  return status;
}

double orc_rel_score_once(const double* T, const double* f1, const double* f2) {
  double R[9], t[3];
  orc_T_to_Rt(T, R, t);
  return orc_rel_score(R, t, f1, f2);
}

/* the essential matrices of the five- and seven-point solvers on their own (unit tests): E_out [10][9] / [3][9] */
int32_t orc_fivept_solve(const double* f1, const double* f2, double* E_out) { return orc_fivept(f1, f2, E_out); }
int32_t orc_sevenpt_solve(const double* f1, const double* f2, double* E_out) { return orc_sevenpt(f1, f2, E_out); }

int32_t orc_eightpt_solve(const double* f1, const double* f2, double* T_out) {
  double R[9], t[3];
  const int ok = orc_eightpt(f1, f2, R, t);
  if (ok) orc_Rt_to_T(R, t, T_out);
  return ok;
}

/* Generalised P3P on its own (unit tests): fb, o, P [9] -> up to 8 poses T_out [8][12] = [R | t]; returns their number.
 * oct_out (optional): the 9 coefficients of the octic in the first depth, in units of the largest world side. */
int32_t orc_gp3p_solve(const double* fb, const double* o, const double* P, double* T_out, double* oct_out) {
  double Rs[72], ts[24];
  const int ns = orc_gp3p(fb, o, P, Rs, ts);
  for (int k = 0; k < ns; ++k) orc_Rt_to_T(Rs + 9 * k, ts + 3 * k, T_out + 12 * k);
  if (oct_out) {
    double L = 0.0;
    for (int i = 0; i < 3; ++i) {
      const int j = (i + 1) % 3;
      const double d0 = P[3 * i] - P[3 * j], d1 = P[3 * i + 1] - P[3 * j + 1], d2 = P[3 * i + 2] - P[3 * j + 2];
      const double d = sqrt(((d0 * d0) + (d1 * d1)) + (d2 * d2));
      if (d > L) L = d;
    }
    double os[9], Ps[9], e12[4], e13[4], e23[4], Ap[4], Bp[5];
    for (int k = 0; k < 9; ++k) {
      os[k] = o[k] / L;
      Ps[k] = P[k] / L;
    }
    orc_gp3p_pair(fb, fb + 3, os, os + 3, Ps, Ps + 3, e12);
    orc_gp3p_pair(fb, fb + 6, os, os + 6, Ps, Ps + 6, e13);
    orc_gp3p_pair(fb + 3, fb + 6, os + 3, os + 6, Ps + 3, Ps + 6, e23);
    orc_gp3p_octic(e12, e13, e23, oct_out, Ap, Bp);
  }
  return ns;
}

/* ... and the Jacobi solver EPnP uses: A [144] in (destroyed), d [12] eigenvalues, V [144] eigenvectors as columns. */
int32_t orc_jacobi12_solve(double* A, double* d, double* V) {
  orc_jacobi12_rr(A, V);
  for (int k = 0; k < 12; ++k) d[k] = A[13 * k];
  return 1;
}

int32_t orc_sample_distinct_once(int32_t n, int32_t k, uint64_t seed, int32_t it, int32_t* s) {
  return orc_sample_distinct(n, k, seed, (uint64_t)it, s);
}
