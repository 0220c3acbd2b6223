// K8/K10 -- 3D-2D absolute-pose RANSAC (central and non-central rig) and K9 -- LM refinement.
//
// Reference call sites replaced (the arithmetic itself lives in OpenGV, outside the tree):
//   omnistereo/pose_est_tools.py:785  absolute_pose_noncentral_ransac(b, cam, p, offsets, rots, thr, iters)
//   omnistereo/pose_est_tools.py:915  absolute_pose_ransac(b, p, algo, thr, iters)
//   omnistereo/pose_est_tools.py:830 / :937  *_optimize_nonlinear
//
// Mapping to CDNA4 (no MFMA: there is no contraction, only per-point FP64 geometry):
//   1. prepare     one workgroup per problem: stable partition of point indices by camera.
//   2. hypotheses  one lane per (problem, iteration): counter-based sample, Kneip P3P in the
//                  sampled camera, 4th-point disambiguation -> (R, t, -R^T t) in HBM (128 B each).
//   3. score       the iters x N hot loop.  A lane keeps its point(s) in VGPRs; the hypothesis is
//                  wave-uniform, fetched with scalar loads (SGPR operands); the inlier decision is
//                  a 64-bit __ballot + popcount per wave, summed in LDS, one integer atomic per
//                  (workgroup, hypothesis).  A squared-cosine test with a 1e-9 guard band decides
//                  all but near-threshold points without FP64 sqrt/div; the guard band falls back
//                  to the exact reference formula, so counts are bit-identical to it.
//   4. select      sequential semantics of sac::Ransac (strictly-better update, adaptive stop)
//                  replayed over the counts; exact inlier mask + ascending index list of the winner.
//   5. refine      one workgroup per problem, Levenberg-Marquardt on (t, Cayley) with fixed-order
//                  reductions (deterministic run to run).
#include <type_traits>
#include "common.h"
#include "ransac_core.h"
#include "epnp_core.h"
#include "gp3p_core.h"

namespace {

constexpr int kMaxCam = 8;
constexpr int kHypDoubles = 16;  // R[9], t[3], -R^T t [3], pad
constexpr int kThreads = 256;
constexpr int kScoreHypChunkMax = 128;

__device__ const double kEye9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
__device__ const double kZero3[3] = {0, 0, 0};

__device__ __forceinline__ uint64_t problem_seed(uint64_t seed, int b) { return seed + (uint64_t)b; }

// ---- 1. prepare -----------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void ransac_prepare_kernel(const int32_t* __restrict__ cam,
                                                                  const int32_t* __restrict__ n_arr, int stride,
                                                                  int ncam, int32_t* __restrict__ perm,
                                                                  int32_t* __restrict__ cinfo) {
  SOSVO_LATENCY_BOUND_PRIO();
  __shared__ int cnt[kThreads][kMaxCam];
  __shared__ int cstart[kMaxCam + 1];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int n = min(n_arr[b], stride);
  const int chunk = (n + kThreads - 1) / kThreads;
  const int begin = min(n, tid * chunk), end = min(n, begin + chunk);
  const int32_t* camb = cam ? cam + (size_t)b * stride : nullptr;
  int local[kMaxCam];
#pragma unroll
  for (int c = 0; c < kMaxCam; ++c) local[c] = 0;
  for (int i = begin; i < end; ++i) {
    int c = camb ? camb[i] : 0;
    c = max(0, min(ncam - 1, c));
#pragma unroll
    for (int k = 0; k < kMaxCam; ++k) local[k] += (k == c);
  }
#pragma unroll
  for (int c = 0; c < kMaxCam; ++c) cnt[tid][c] = local[c];
  __syncthreads();
  if (tid < kMaxCam) {
    int running = 0;
    for (int t = 0; t < kThreads; ++t) {
      const int v = cnt[t][tid];
      cnt[t][tid] = running;
      running += v;
    }
    cinfo[(size_t)b * (2 * kMaxCam + 1) + kMaxCam + 1 + tid] = running;  // ccount
    cstart[tid + 1] = running;
  }
  __syncthreads();
  if (tid == 0) {
    cstart[0] = 0;
    for (int c = 0; c < kMaxCam; ++c) cstart[c + 1] += cstart[c];
    for (int c = 0; c <= kMaxCam; ++c) cinfo[(size_t)b * (2 * kMaxCam + 1) + c] = cstart[c];
  }
  __syncthreads();
  int pos[kMaxCam];
#pragma unroll
  for (int c = 0; c < kMaxCam; ++c) pos[c] = cstart[c] + cnt[tid][c];
  for (int i = begin; i < end; ++i) {
    int c = camb ? camb[i] : 0;
    c = max(0, min(ncam - 1, c));
    int dst = 0;
#pragma unroll
    for (int k = 0; k < kMaxCam; ++k) {
      if (k == c) {
        dst = pos[k];
        pos[k] += 1;
      }
    }
    perm[(size_t)b * stride + dst] = i;
  }
}

// ---- 2. hypotheses --------------------------------------------------------------------------
// Workgroup -> (problem, hypothesis chunk) for the hypothesis generators: workgroups are dealt round-robin over the 8
// XCDs by linear id, so with the chunk as the fastest index the ~30 workgroups of one problem land on all eight L2s and
// every one of them fetches the problem's correspondences (measured: 0.54 MB of HBM fetches per pair for 37 KB of
// bearings and points).  Here XCD x takes the problems x, x + 8, ... with all their chunks (1-D grid of
// 8 * ceil(nprob / 8) * chunks workgroups; the padding ones exit).
__device__ __forceinline__ bool hyp_block(int nprob, int chunks, int* b, int* chunk) {
  const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
  *b = (slot / chunks) * 8 + xcd;
  *chunk = slot - (slot / chunks) * chunks;
  return *b < nprob;
}
static inline unsigned hyp_grid(int nprob, int chunks) { return (unsigned)(8 * cdiv(nprob, 8) * chunks); }

// The 64 hypothesis records of a wave (R[9], t[3], -R^T t[3], pad: 128 B each) leave through LDS: a lane storing its own
// record issues 15 eight-byte stores 128 B apart from its neighbours' (partial sectors: 2.6 x the bytes at the memory
// interface); transposed, every store instruction of the wave writes 512 contiguous bytes.  One-wave workgroups; all 64
// lanes call (ok = false for a failed solve or a lane beyond H: NaN in R[0], the rest zero).
__device__ __forceinline__ void hyp_store_wave(double* __restrict__ hyp_b, int it0, int H, bool ok, const double* R,
                                               const double* t) {
  __shared__ double sh[64 * 17];
  const int lane = threadIdx.x;
  double* row = sh + lane * 17;
#pragma unroll
  for (int k = 0; k < 9; ++k) row[k] = ok ? R[k] : (k == 0 ? __longlong_as_double(0x7FF8000000000000LL) : 0.0);
#pragma unroll
  for (int k = 0; k < 3; ++k) row[9 + k] = ok ? t[k] : 0.0;
  row[12] = ok ? -(((R[0] * t[0]) + (R[3] * t[1])) + (R[6] * t[2])) : 0.0;
  row[13] = ok ? -(((R[1] * t[0]) + (R[4] * t[1])) + (R[7] * t[2])) : 0.0;
  row[14] = ok ? -(((R[2] * t[0]) + (R[5] * t[1])) + (R[8] * t[2])) : 0.0;
  row[15] = 0.0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int idx = j * 64 + lane, rec = idx >> 4, e = idx & 15;
    if (it0 + rec < H) hyp_b[(size_t)(it0 + rec) * kHypDoubles + e] = sh[rec * 17 + e];
  }
}

__global__ __launch_bounds__(64) void ransac_hyp_kernel(const double* __restrict__ f, const double* __restrict__ p,
                                                        const int32_t* __restrict__ cam,
                                                        const double* __restrict__ cam_off,
                                                        const double* __restrict__ cam_rot,
                                                        const int32_t* __restrict__ n_arr, int nprob, int stride, int H,
                                                        uint64_t seed, const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ cinfo, double* __restrict__ hyp,
                                                        int32_t* __restrict__ counts) {
  SOSVO_LATENCY_BOUND_PRIO();
  int b, chunk;
  if (!hyp_block(nprob, (H + 63) >> 6, &b, &chunk)) return;  // uniform
  const int it = chunk * 64 + threadIdx.x;
  const int n = min(n_arr[b], stride);
  const size_t base = (size_t)b * stride;
  const int32_t* ci = cinfo + (size_t)b * (2 * kMaxCam + 1);
  const double* off = cam ? cam_off : kZero3;
  const double* rot = cam ? cam_rot : kEye9;
  double R[9], t[3];
  int ok = 0;
  if (it < H)
    ok = sv_hypothesis(f + 3 * base, p + 3 * base, cam ? cam + base : nullptr, off, rot, n, perm + base, ci, ci + kMaxCam + 1,
                       problem_seed(seed, b), (uint64_t)it, R, t);
  hyp_store_wave(hyp + (size_t)b * H * kHypDoubles, chunk * 64, H, ok != 0, R, t);
  if (it < H) counts[(size_t)b * H + it] = ok ? 0 : -1;
}

// TWOPT (SOSVO_FLAG_TWOPT, central problems): two distinct correspondences, translation only (sv_hypothesis_twopt).
__global__ __launch_bounds__(64) void ransac_hyp_twopt_kernel(const double* __restrict__ f, const double* __restrict__ p,
                                                              const int32_t* __restrict__ n_arr, int nprob, int stride,
                                                              int H, uint64_t seed, double* __restrict__ hyp,
                                                              int32_t* __restrict__ counts) {
  int b, chunk;
  if (!hyp_block(nprob, (H + 63) >> 6, &b, &chunk)) return;  // uniform
  const int it = chunk * 64 + threadIdx.x;
  const int n = min(n_arr[b], stride);
  const size_t base = (size_t)b * stride;
  double R[9], t[3];
  int ok = 0;
  if (it < H) ok = sv_hypothesis_twopt(f + 3 * base, p + 3 * base, n, problem_seed(seed, b), (uint64_t)it, R, t);
  hyp_store_wave(hyp + (size_t)b * H * kHypDoubles, chunk * 64, H, ok != 0, R, t);
  if (it < H) counts[(size_t)b * H + it] = ok ? 0 : -1;
}

// The generalised-P3P hypothesis generator (SOSVO_FLAG_GP3P): one lane per (problem, iteration): four distinct
// correspondences out of all cameras, sv_hypothesis_gp3p (gp3p_core.h, the oracle's text).  Latency-bound (SQ counters:
// 32.5 k VALU instructions per wave at 7.2 SIMD-cycles each: chains of FP64 divisions and square roots), so the register
// budget is capped for three waves per SIMD (measured 2 / 3 / 4 / 5 / 6 waves: 0.785 / 0.700 / 0.715 / 0.806 / 0.781 ms).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void ransac_hyp_gp3p_kernel(const double* __restrict__ f, const double* __restrict__ p,
                                                             const int32_t* __restrict__ cam,
                                                             const double* __restrict__ cam_off,
                                                             const double* __restrict__ cam_rot,
                                                             const int32_t* __restrict__ n_arr, int nprob, int stride,
                                                             int H, uint64_t seed, double* __restrict__ hyp,
                                                             int32_t* __restrict__ counts) {
  SOSVO_LATENCY_BOUND_PRIO();
  int b, chunk;
  if (!hyp_block(nprob, (H + 63) >> 6, &b, &chunk)) return;  // uniform
  const int it = chunk * 64 + threadIdx.x;
  const int n = min(n_arr[b], stride);
  const size_t base = (size_t)b * stride;
  const double* off = cam ? cam_off : kZero3;
  const double* rot = cam ? cam_rot : kEye9;
  double R[9], t[3];
  int ok = 0;
  if (it < H)
    ok = sv_hypothesis_gp3p(f + 3 * base, p + 3 * base, cam ? cam + base : nullptr, off, rot, n, problem_seed(seed, b),
                            (uint64_t)it, R, t);
  hyp_store_wave(hyp + (size_t)b * H * kHypDoubles, chunk * 64, H, ok != 0, R, t);
  if (it < H) counts[(size_t)b * H + it] = ok ? 0 : -1;
}

// The EPnP hypothesis generator (central problems, SOSVO_FLAG_EPNP), one hypothesis per lane, as TWO kernels:
//   ransac_epnp_eigen_kernel  sample -> control points, barycentric coordinates -> M^T M -> the 12 x 12 eigen-solver
//       entirely in registers (sv_epnp_null4_ql: Householder tridiagonalisation + implicit QL, 144 + 24 doubles, one wave
//       per SIMD, no LDS) -> the four null-space vectors, 48 doubles per hypothesis, to the workspace;
//   ransac_epnp_pose_kernel   recomputes the (cheap) first half, reads the four vectors, runs the three beta
//       initialisations + Gauss-Newton + absolute orientation.  On its own this half needs a fraction of the registers, so
//       several waves per SIMD hide its divide / square-root chains.
// One monolithic kernel held 512 registers per lane for its whole life and its eigen-solver's arrays in LDS (96 hypotheses
// per CU in flight, every LDS round trip exposed): 5.5 ms per 128 pairs x 2000 hypotheses.
__device__ __forceinline__ int epnp_sample_front(const double* __restrict__ f, const double* __restrict__ p, int n, size_t base,
                                                 uint64_t seed, int it, double (&p6)[18], double (&uv)[2 * SV_EPNP_MAXN],
                                                 double (&cw)[12], double (&alphas)[4 * SV_EPNP_MAXN]) {
  int32_t s6[6];
  if (!sv_sample_distinct(n, 6, seed, (uint64_t)it, s6)) return 0;
  double f6[18];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f6[3 * k + c] = f[3 * (base + s6[k]) + c];
      p6[3 * k + c] = p[3 * (base + s6[k]) + c];
    }
  return sv_epnp_front(f6, p6, 6, uv, cw, alphas);
}

constexpr int kEpnpVecDoubles = 48;
__global__ __launch_bounds__(64) void ransac_epnp_eigen_kernel(const double* __restrict__ f, const double* __restrict__ p,
                                                               const int32_t* __restrict__ n_arr, int nprob, int stride,
                                                               int H, uint64_t seed, double* __restrict__ vvbuf,
                                                               int32_t* __restrict__ counts) {
  int b, chunk;
  if (!hyp_block(nprob, (H + 63) >> 6, &b, &chunk)) return;  // uniform
  const int it = chunk * 64 + threadIdx.x;
  if (it >= H) return;
  const int n = min(n_arr[b], stride);
  double p6[18], uv[2 * SV_EPNP_MAXN], cw[12], alphas[4 * SV_EPNP_MAXN];
  int ok = epnp_sample_front(f, p, n, (size_t)b * stride, problem_seed(seed, b), it, p6, uv, cw, alphas);
  double* out = vvbuf + ((size_t)b * H + it) * kEpnpVecDoubles;
  if (ok) {
    double vv[48];
    ok = sv_epnp_null4_ql(alphas, uv, 6, vv);
    if (ok) {
#pragma unroll
      for (int k = 0; k < 48; ++k) out[k] = vv[k];
    }
  }
  counts[(size_t)b * H + it] = ok ? 0 : -1;
}

__global__ __launch_bounds__(64) void ransac_epnp_pose_kernel(const double* __restrict__ f, const double* __restrict__ p,
                                                              const int32_t* __restrict__ n_arr, int nprob, int stride,
                                                              int H, uint64_t seed, const double* __restrict__ vvbuf,
                                                              double* __restrict__ hyp, int32_t* __restrict__ counts) {
  SOSVO_LATENCY_BOUND_PRIO();
  int b, chunk;
  if (!hyp_block(nprob, (H + 63) >> 6, &b, &chunk)) return;  // uniform
  const int it = chunk * 64 + threadIdx.x;
  const int n = min(n_arr[b], stride);
  double R[9], t[3];
  int ok = it < H && counts[(size_t)b * H + it] == 0;
  if (ok) {
    double p6[18], uv[2 * SV_EPNP_MAXN], cw[12], alphas[4 * SV_EPNP_MAXN], vv[48];
    ok = epnp_sample_front(f, p, n, (size_t)b * stride, problem_seed(seed, b), it, p6, uv, cw, alphas);  // (same bits as before)
    const double* in = vvbuf + ((size_t)b * H + it) * kEpnpVecDoubles;
#pragma unroll
    for (int k = 0; k < 48; ++k) vv[k] = in[k];
    if (ok) ok = sv_epnp_back(p6, 6, uv, cw, alphas, vv, R, t);
  }
  hyp_store_wave(hyp + (size_t)b * H * kHypDoubles, chunk * 64, H, ok != 0, R, t);
  if (it < H) counts[(size_t)b * H + it] = ok ? 0 : -1;
}

// ---- 3. score -------------------------------------------------------------------------------
// Exact reference decision for one point given (R, it = -R^T t): sv_score(...) < thr, with the
// translation term already folded (bitwise the same value sv_score computes internally).
template <bool IDENT>
__device__ __forceinline__ bool inlier_exact(const double* __restrict__ h, const double fx, const double fy,
                                             const double fz, const double px, const double py, const double pz,
                                             const double ox, const double oy, const double oz, const double* Rc,
                                             const double thr) {
  const double vx = (((h[0] * px) + (h[3] * py)) + (h[6] * pz)) + h[12];
  const double vy = (((h[1] * px) + (h[4] * py)) + (h[7] * pz)) + h[13];
  const double vz = (((h[2] * px) + (h[5] * py)) + (h[8] * pz)) + h[14];
  const double wx = vx - ox, wy = vy - oy, wz = vz - oz;
  double ux = wx, uy = wy, uz = wz;
  if (!IDENT) {
    ux = ((Rc[0] * wx) + (Rc[3] * wy)) + (Rc[6] * wz);
    uy = ((Rc[1] * wx) + (Rc[4] * wy)) + (Rc[7] * wz);
    uz = ((Rc[2] * wx) + (Rc[5] * wy)) + (Rc[8] * wz);
  }
  const double nrm = sqrt(((ux * ux) + (uy * uy)) + (uz * uz));
  const double gx = ux / nrm, gy = uy / nrm, gz = uz / nrm;
  return (1.0 - (((fx * gx) + (fy * gy)) + (fz * gz))) < thr;
}

// ---- tier 1 of the scoring decision: single precision with a proven error term ------------------------------
// The reference decides  1 - f.u/|u| < thr  with u = R p + (i - o), i.e. (thr < 0.5)  s > 0 and s^2 > c2 q  with
// s = f.u, q = u.u, c2 = (1 - thr)^2.  Tier 1 evaluates s~, q~ in single precision (inputs rounded to float, the same
// fma chains as the double-precision fast tier) and decides only where the outcome cannot depend on the rounding:
//   every component of u~ is off by at most  E = 5.1 eps (rho P1 + |i|_inf + |o|_inf)   (eps = 2^-24, rho = max |R_k|,
//   P1 = |px| + |py| + |pz|: two input roundings + at most three fma roundings per term), so |u~ - u| <= sqrt(3) E,
//   |s~ - s| <= sqrt(3) E + 4.1 eps n,  |q~ - n^2| <= 3 eps n^2,  | |u|^2 - n^2 | <= 2 sqrt(3) n E + 3 E^2  (n = |u~|),
//   and with  n E <= (K E^2 + n^2 / K) / 2  for any K > 0:
//     s~^2 > (c2 + 3.47 / K + 20 eps) q~ + (3.47 K + 6) E^2   =>  s^2 > c2 (1 + 1e-9) q  and s > 0   (inlier),
//     s~ |s~| < (c2 - 3.47 / K - 20 eps) q~ - (3.47 K + 6) E^2   =>  s <= 0 or s^2 < c2 (1 - 1e-9) q   (not one)
//   (the 1e-9 margins are the double-precision fast tier's: beyond them its evaluation and the reference's agree; the
//   kernel compares the SIGNED square s~ |s~| on both lines.  For s~ < 0 the second line reads s~^2 + (c2 - ..) q~ > (..) E^2:
//   either s~ <= -(sqrt(3) E + 4.1 eps n), then s <= 0; or s~^2 is below that bound's square, which leaves
//   (c2 - ..) q~ > 3.47 K E^2 - 34 eps^2 n^2, i.e. E < n / 1900, and s < 2 (sqrt(3) E + 4.1 eps n) < 2e-3 n gives
//   s^2 < 4e-6 n^2 < c2 (1 - 1e-9) |u|^2 with c2 > 1/4.)
// E^2 <= 78.03 eps^2 (rho^2 P1^2 + |i|_inf^2 + |o|_inf^2) <= 78.03 eps^2 (max(rho^2, 1) (P1^2 + |o|_inf^2) + |i|_inf^2): one
// fma of a per-hypothesis pair (ea, eb) with a per-lane constant.  K = 2^20 balances the two terms for |u| ~ |p|: the
// band is ~6e-6 of c2 wide, 3e-5 rad around the threshold angle at 5 degrees.  The bound assumes float arithmetic without
// overflow: hypotheses and points with a magnitude above 1e9, NaNs, and bearing vectors that are not of unit length get a
// NaN / +inf error coefficient, every comparison with it is false and the lane stays undecided.
constexpr double kTier1K = 1048576.0;
constexpr double kTier1Eps = 5.9604644775390625e-08;
constexpr double kTier1QBand = 3.47 / kTier1K + 20.0 * kTier1Eps;
constexpr double kTier1Max = 1e9;
constexpr double kTier1E2 = (3.47 * kTier1K + 6.0) * 78.03 * kTier1Eps * kTier1Eps * (1.0 + 1e-4);

__device__ __forceinline__ float score_tier1_lane(double fx, double fy, double fz, double px, double py, double pz, double ox,
                                                  double oy, double oz) {
  const double p1 = fabs(px) + fabs(py) + fabs(pz);
  const double om = fmax(fabs(ox), fmax(fabs(oy), fabs(oz)));
  const double fn = fx * fx + fy * fy + fz * fz;
  const double l = (p1 * p1 + om * om) * (1.0 + 1e-6);
  // magnitudes up to 1e9 on both sides (kTier1Max): |u_k| <= 1e18 + 2e9, q <= 3.1e36 -- nothing overflows in float
  return (fabs(fn - 1.0) <= 1e-6 && p1 <= kTier1Max && om <= kTier1Max) ? (float)l : __builtin_inff();
}

// h: R[9], -R^T t [3] in double precision -> hs[0..11] the same as floats, hs[12] = ea, hs[13] = eb (rounded up)
__device__ __forceinline__ void score_tier1_hyp(const double* __restrict__ h, float* __restrict__ hs) {
  double rho = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    hs[k] = (float)h[k];
    rho = fmax(rho, fabs(h[k]));
  }
  const double im = fmax(fabs(h[9]), fmax(fabs(h[10]), fabs(h[11])));
  hs[9] = (float)h[9], hs[10] = (float)h[10], hs[11] = (float)h[11];
  // (fmax drops a NaN operand: test the sums, which keep it)
  const double chk = (((((((h[0] + h[1]) + h[2]) + h[3]) + h[4]) + h[5]) + h[6]) + h[7]) + h[8] + h[9] + h[10] + h[11];
  const double ea = kTier1E2 * fmax(rho * rho, 1.0) * (1.0 + 1e-6);
  // the floor of eb keeps every lane with q below ~1e-24 undecided (float underflow in the squares is then < 1e-14 of q)
  const double eb = kTier1E2 * (im * im) * (1.0 + 1e-6) + 1e-24;
  const bool fin = chk == chk && rho <= kTier1Max && im <= kTier1Max;
  hs[12] = fin ? (float)ea : __builtin_nanf("");
  hs[13] = fin ? (float)eb : __builtin_nanf("");
  hs[14] = 0.0f, hs[15] = 0.0f;
}

// T1 = false: double-precision tiers only (thr >= 0.5, or camera rotations that are not the identity) and the lane's
// constants in double precision in registers.  T1 = true: tier 1 in front; the lane keeps the float copies and its
// row, and re-reads bearing, point and camera offset in the rare fallback (128 VGPRs, four waves per SIMD).
template <bool IDENT, int PPT, bool T1>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(T1 ? 4 : 1))) void ransac_score_kernel(
    const double* __restrict__ f, const double* __restrict__ p, const int32_t* __restrict__ cam,
    const double* __restrict__ cam_off, const double* __restrict__ cam_rot, const int32_t* __restrict__ n_arr,
    int stride, int H, int hchunk, int hspan, double thr, int fast_ok, const double* __restrict__ hyp,
    int32_t* __restrict__ counts) {
  static_assert(!T1 || (IDENT && PPT <= 4), "tier 1 is written for the identity-rotation form");
  SOSVO_LATENCY_BOUND_PRIO();
  constexpr bool KEEP = !T1;
  __shared__ int lcnt[kScoreHypChunkMax];  // by position in the compacted list
  __shared__ double shyp[kScoreHypChunkMax][12];
  __shared__ unsigned long long sok[(kScoreHypChunkMax + 63) / 64];  // bit = hypothesis of the chunk is a solved one
  __shared__ int sidx[kScoreHypChunkMax];   // compacted list of the solved hypotheses of the chunk
  // tier 1: the solved hypotheses once more in single precision, R, -R^T t and the two coefficients of the error term
  // (see score_tier1_hyp), in list order
  __shared__ float shyp32[T1 ? kScoreHypChunkMax : 1][16];
  const int tid = threadIdx.x, b = blockIdx.z;
  const int n = min(n_arr[b], stride);
  // XCD-aware grid: workgroups are dealt round-robin over the 8 XCDs by linear id, so the dimension with
  // early-exiting workgroups (point blocks beyond n) must NOT be the fastest one: x = hypothesis chunk.
  // Point sets of 64 are split EVENLY over the point blocks this problem needs (nb = sets / 16 rounded up, each block
  // takes ceil(sets / nb) consecutive sets): a ragged problem (say 22 sets) runs as 11 + 11 instead of 16 + 6, so
  // every block amortises its per-hypothesis work (LDS reads, the count atomic) over about the same number of sets.
  const int nsets = (n + 63) >> 6;
  const int nb = (nsets + (kThreads / 64) * PPT - 1) / ((kThreads / 64) * PPT);
  if ((int)blockIdx.y >= nb) return;
  const int spb = (nsets + nb - 1) / nb;
  const int set0 = blockIdx.y * spb, set1 = min(set0 + spb, nsets);
  const int p0 = set0 * 64;

  double fx[KEEP ? PPT : 1], fy[KEEP ? PPT : 1], fz[KEEP ? PPT : 1];
  double px[KEEP ? PPT : 1], py[KEEP ? PPT : 1], pz[KEEP ? PPT : 1];
  double ox[KEEP ? PPT : 1], oy[KEEP ? PPT : 1], oz[KEEP ? PPT : 1];
  float fxs[T1 ? PPT : 1], fys[T1 ? PPT : 1], fzs[T1 ? PPT : 1], pxs[T1 ? PPT : 1], pys[T1 ? PPT : 1], pzs[T1 ? PPT : 1];
  float oxs[T1 ? PPT : 1], oys[T1 ? PPT : 1], ozs[T1 ? PPT : 1], lerr[T1 ? PPT : 1];
  int rowi[PPT];  // the lane's row of set r, relative to the problem's first row
  double Rc[IDENT ? 1 : 9];
  // point sets of 64 are dealt round-robin over the 4 waves (set = r * 4 + wave): the live sets of a ragged
  // problem spread evenly, and a wave skips its sets beyond n altogether.  Everything that steers the hot loop
  // is wave-uniform and kept in SGPRs (wave index by readfirstlane, ballots, popcounts): the loop has scalar
  // branches only, no exec-mask juggling.
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // wave w of a workgroup sits on SIMD w: which wave takes the short column of a ragged block rotates with the
  // workgroup, or SIMD 0 would always carry the most sets and SIMD 3 the fewest
  const int wslot = (wave + blockIdx.x + blockIdx.y + blockIdx.z) & (kThreads / 64 - 1);
  const size_t base = (size_t)b * stride;
  unsigned long long vmask[PPT];
  int nlive_v = 0;
#pragma unroll
  for (int r = 0; r < PPT; ++r) {
    const int set = set0 + r * (kThreads / 64) + wslot;
    const int i0 = set * 64;
    const int i = i0 + lane;
    nlive_v += set < set1 ? 1 : 0;
    const bool valid = i < n;
    vmask[r] = __ballot(valid);
    rowi[r] = valid ? i : p0;
    const size_t row = base + rowi[r];
    const double lfx = f[3 * row + 0], lfy = f[3 * row + 1], lfz = f[3 * row + 2];
    const double lpx = p[3 * row + 0], lpy = p[3 * row + 1], lpz = p[3 * row + 2];
    const int c = cam ? cam[row] : 0;
    const double* o = cam ? cam_off + 3 * c : kZero3;
    const double lox = o[0], loy = o[1], loz = o[2];
    if (KEEP) {
      fx[r] = lfx, fy[r] = lfy, fz[r] = lfz;
      px[r] = lpx, py[r] = lpy, pz[r] = lpz;
      ox[r] = lox, oy[r] = loy, oz[r] = loz;
    }
    if (!IDENT) {
      const double* rc = cam ? cam_rot + 9 * c : kEye9;
#pragma unroll
      for (int k = 0; k < 9; ++k) Rc[k] = rc[k];
    }
    if (T1) {
      fxs[r] = (float)lfx, fys[r] = (float)lfy, fzs[r] = (float)lfz;
      pxs[r] = (float)lpx, pys[r] = (float)lpy, pzs[r] = (float)lpz;
      oxs[r] = (float)lox, oys[r] = (float)loy, ozs[r] = (float)loz;
      lerr[r] = score_tier1_lane(lfx, lfy, lfz, lpx, lpy, lpz, lox, loy, loz);
    }
  }
  // the live sets of a wave are its first nlive ones (set grows with r); scalar
  const int nlive = __builtin_amdgcn_readfirstlane(nlive_v);
  const double c1 = 1.0 - thr;
  const double c2 = c1 * c1;
  const double c2hi = c2 * (1.0 + 1e-9), c2lo = c2 * (1.0 - 1e-9);
  const float c2hs = (float)((c2 + kTier1QBand) * (1.0 + 1e-6)), c2ls = (float)((c2 - kTier1QBand) * (1.0 - 1e-6));
  // the workgroup's hypotheses [blockIdx.x * hspan, + hspan) go through LDS in chunks of hchunk: the lanes' points are
  // loaded and converted once for all of them
  const int hend = min(H, ((int)blockIdx.x + 1) * hspan);
  for (int h0 = blockIdx.x * hspan; h0 < hend; h0 += hchunk) {
  const int h1 = min(hend, h0 + hchunk);
  for (int k = tid; k < kScoreHypChunkMax; k += kThreads) lcnt[k] = 0;
  const int nh = h1 - h0;
  const double* hb = hyp + ((size_t)b * H + h0) * kHypDoubles;
  static_assert(kScoreHypChunkMax == 128, "two words of solved-hypothesis flags");
  if (wave < 2) {  // wave w flags hypotheses [64 w, 64 w + 64): a failed minimal solve left a NaN in R[0]
    const int hh = wave * 64 + lane;
    const double r00 = hh < nh ? hb[(size_t)hh * kHypDoubles] : 0.0;
    const unsigned long long okb = __ballot(hh < nh && r00 == r00);
    if (lane == 0) sok[wave] = okb;
  }
  __syncthreads();
  // the solved hypotheses go to LDS as a compact list (position = rank among the solved ones, coalesced loads): the
  // loop below walks it without bit scans and reads an entry as wave-wide LDS broadcasts
  const unsigned long long ok0 = sok[0], ok1 = sok[1];
  const int nok = __builtin_amdgcn_readfirstlane(__popcll(ok0) + __popcll(ok1));
  for (int k = tid; k < nh * 12; k += kThreads) {
    const int hh = k / 12, e = k - hh * 12;
    const unsigned long long w = hh < 64 ? ok0 : ok1;
    if ((w >> (hh & 63)) & 1ULL) {
      const int pos = __popcll(w & ((1ULL << (hh & 63)) - 1ULL)) + (hh < 64 ? 0 : __popcll(ok0));
      shyp[pos][e] = hb[(size_t)hh * kHypDoubles + (e < 9 ? e : e + 3)];
      if (e == 0) sidx[pos] = hh;
    }
  }
  __syncthreads();
  if (T1) {
    for (int pos = tid; pos < nok; pos += kThreads) score_tier1_hyp(shyp[pos], shyp32[pos]);
    __syncthreads();
  }

  // the double-precision tiers for the lanes `und1[r]` of one hypothesis; returns what they add to its count
  auto tier2 = [&](auto nl, const int j, const unsigned long long* und1) -> int {
    constexpr int NL = decltype(nl)::value;
    int total = 0;
    const double* hp = shyp[j];
    const double r0 = hp[0], r1 = hp[1], r2 = hp[2], r3 = hp[3], r4 = hp[4], r5 = hp[5], r6 = hp[6], r7 = hp[7], r8 = hp[8];
    const double ix = hp[9], iy = hp[10], iz = hp[11];
    // Fast decision with fused multiply-adds (21 instead of 34 FP64 ops): squared-cosine test with a relative
    // guard band of 1e-9, five orders of magnitude above the rounding difference between this evaluation and
    // the exact one for |p| / |u| < 1e6.  Lanes inside the band are left to the oracle's formula, operation for
    // operation (no contraction), so the counts are identical to the sequential reference.  The predicates are
    // combined without branches; one wave-wide test per set tells whether any lane is undecided.
#pragma unroll
    for (int r = 0; r < NL && r < PPT; ++r) {
      if (!und1[r]) continue;  // scalar
      double Fx, Fy, Fz, Px, Py, Pz, Ox, Oy, Oz;
      if (KEEP) {
        Fx = fx[r], Fy = fy[r], Fz = fz[r], Px = px[r], Py = py[r], Pz = pz[r];
        Ox = ox[r], Oy = oy[r], Oz = oz[r];
      } else {
        const size_t row = base + rowi[r];
        Fx = f[3 * row + 0], Fy = f[3 * row + 1], Fz = f[3 * row + 2];
        Px = p[3 * row + 0], Py = p[3 * row + 1], Pz = p[3 * row + 2];
        const double* o = cam ? cam_off + 3 * cam[row] : kZero3;
        Ox = o[0], Oy = o[1], Oz = o[2];
      }
      unsigned long long und = und1[r];
      if (fast_ok) {
        double ux = fma(r0, Px, fma(r3, Py, fma(r6, Pz, ix - Ox)));
        double uy = fma(r1, Px, fma(r4, Py, fma(r7, Pz, iy - Oy)));
        double uz = fma(r2, Px, fma(r5, Py, fma(r8, Pz, iz - Oz)));
        if (!IDENT) {
          const double wx = ux, wy = uy, wz = uz;
          ux = fma(Rc[0], wx, fma(Rc[IDENT ? 0 : 3], wy, Rc[IDENT ? 0 : 6] * wz));
          uy = fma(Rc[IDENT ? 0 : 1], wx, fma(Rc[IDENT ? 0 : 4], wy, Rc[IDENT ? 0 : 7] * wz));
          uz = fma(Rc[IDENT ? 0 : 2], wx, fma(Rc[IDENT ? 0 : 5], wy, Rc[IDENT ? 0 : 8] * wz));
        }
        const double s = fma(Fx, ux, fma(Fy, uy, Fz * uz));
        const double q = fma(ux, ux, fma(uy, uy, uz * uz));
        const double lhs = s * s;
        // thr < 0.5: an inlier has cosine > 0.5, far from any rounding of s (NaN: not an inlier)
        const unsigned long long pos = __ballot(s > 0.0) & und1[r];
        const unsigned long long hi = __ballot(lhs > c2hi * q);
        const unsigned long long lo = __ballot(lhs < c2lo * q);
        total += __popcll(pos & hi);
        und = pos & ~hi & ~lo;
      }
      if (und) {  // scalar; rare when fast_ok
        const double vx = (((r0 * Px) + (r3 * Py)) + (r6 * Pz)) + ix;
        const double vy = (((r1 * Px) + (r4 * Py)) + (r7 * Pz)) + iy;
        const double vz = (((r2 * Px) + (r5 * Py)) + (r8 * Pz)) + iz;
        const double wx = vx - Ox, wy = vy - Oy, wz = vz - Oz;
        double ux = wx, uy = wy, uz = wz;
        if (!IDENT) {
          ux = ((Rc[0] * wx) + (Rc[IDENT ? 0 : 3] * wy)) + (Rc[IDENT ? 0 : 6] * wz);
          uy = ((Rc[IDENT ? 0 : 1] * wx) + (Rc[IDENT ? 0 : 4] * wy)) + (Rc[IDENT ? 0 : 7] * wz);
          uz = ((Rc[IDENT ? 0 : 2] * wx) + (Rc[IDENT ? 0 : 5] * wy)) + (Rc[IDENT ? 0 : 8] * wz);
        }
        const double q = ((ux * ux) + (uy * uy)) + (uz * uz);
        const double nrm = sqrt(q);
        const double gx = ux / nrm, gy = uy / nrm, gz = uz / nrm;
        total += __popcll(__ballot((1.0 - (((Fx * gx) + (Fy * gy)) + (Fz * gz))) < thr) & und);
      }
    }
    return total;
  };

  // Tier 1 for one hypothesis of the list: the decision in single precision with a per-lane, per-hypothesis error
  // term (score_tier1_lane, score_tier1_hyp: a proven bound on what single precision can do to s^2 - c2 q); 23
  // fast-rate operations and two compares per set instead of 21 double-precision operations and three.  Lanes it
  // cannot decide (~1e-4 of them; one wave-step in ~30 has any) are returned in und1 and go through tier2, so the
  // counts stay those of the sequential reference.  Straight-line code over the wave's NL live sets.
  auto tier1 = [&](auto nl, const int j, unsigned long long* und1, unsigned long long& any1) -> int {
    constexpr int NL = decltype(nl)::value;
    int total = 0;
    const float* hs = shyp32[T1 ? j : 0];
    const float q0 = hs[0], q1 = hs[1], q2 = hs[2], q3 = hs[3], q4 = hs[4], q5 = hs[5], q6 = hs[6], q7 = hs[7], q8 = hs[8];
    const float jx = hs[9], jy = hs[10], jz = hs[11], ea = hs[12], eb = hs[13];
#pragma unroll
    for (int r = 0; r < NL && r < PPT; ++r) {
      const float ux = fmaf(q0, pxs[T1 ? r : 0], fmaf(q3, pys[T1 ? r : 0], fmaf(q6, pzs[T1 ? r : 0], jx - oxs[T1 ? r : 0])));
      const float uy = fmaf(q1, pxs[T1 ? r : 0], fmaf(q4, pys[T1 ? r : 0], fmaf(q7, pzs[T1 ? r : 0], jy - oys[T1 ? r : 0])));
      const float uz = fmaf(q2, pxs[T1 ? r : 0], fmaf(q5, pys[T1 ? r : 0], fmaf(q8, pzs[T1 ? r : 0], jz - ozs[T1 ? r : 0])));
      const float sv = fmaf(fxs[T1 ? r : 0], ux, fmaf(fys[T1 ? r : 0], uy, fzs[T1 ? r : 0] * uz));
      const float qv = fmaf(ux, ux, fmaf(uy, uy, uz * uz));
      const float lhs = sv * fabsf(sv);  // the signed square (one fast-rate multiply with |.| on an operand; v_max_f32 is slow-rate)
      const float e2 = fmaf(ea, lerr[T1 ? r : 0], eb);
      const unsigned long long hi = __ballot(lhs > fmaf(c2hs, qv, e2));
      const unsigned long long lo = __ballot(lhs < fmaf(c2ls, qv, -e2));
      total += __popcll(hi & vmask[r]);
      und1[r] = ~hi & ~lo & vmask[r];
      any1 |= und1[r];
    }
    return total;
  };

  // the wave's count of live sets is fixed: the hypothesis loop exists once per count (no switch inside it).  The
  // count of list entry j stays in lane j & 63 of the wave until 64 entries are through.
  auto hyp_loop = [&](auto nl) {
    constexpr int NL = decltype(nl)::value;
    int cntv = 0;
    for (int j = 0; j < nok; ++j) {
      unsigned long long und[PPT], any = 0ULL;
      int tot = 0;
      if (T1) {
        tot = tier1(nl, j, und, any);
      } else {
#pragma unroll
        for (int r = 0; r < NL && r < PPT; ++r) {
          und[r] = vmask[r];
          any |= vmask[r];
        }
      }
      if (any) tot += tier2(nl, j, und);  // scalar; with tier 1 in front: rare
      if (lane == (j & 63)) cntv += tot;
      if ((j & 63) == 63) {
        if (cntv) atomicAdd(&lcnt[(j & ~63) + lane], cntv);
        cntv = 0;
      }
    }
    if (cntv) atomicAdd(&lcnt[((nok - 1) & ~63) + lane], cntv);
  };
  if (nlive >= 4) hyp_loop(std::integral_constant<int, 4>{});
  else if (nlive == 3) hyp_loop(std::integral_constant<int, 3>{});
  else if (nlive == 2) hyp_loop(std::integral_constant<int, 2>{});
  else if (nlive == 1) hyp_loop(std::integral_constant<int, 1>{});
  __syncthreads();
  for (int k = tid; k < nok; k += kThreads) {
    const int v = lcnt[k];
    if (v) atomicAdd(&counts[(size_t)b * H + h0 + sidx[k]], v);
  }
  }  // chunks of the workgroup
}

// ---- 4. select ------------------------------------------------------------------------------
template <bool IDENT>
__global__ __launch_bounds__(kThreads) void ransac_select_kernel(
    const double* __restrict__ f, const double* __restrict__ p, const int32_t* __restrict__ cam,
    const double* __restrict__ cam_off, const double* __restrict__ cam_rot, const int32_t* __restrict__ n_arr,
    int stride, int H, double thr, int adaptive, const double* __restrict__ hyp, const int32_t* __restrict__ counts,
    double* __restrict__ T_out, uint8_t* __restrict__ mask, int32_t* __restrict__ inl_idx,
    int32_t* __restrict__ n_inl, int32_t* __restrict__ info) {
  SOSVO_LATENCY_BOUND_PRIO();
  __shared__ long long red[kThreads / 64];
  __shared__ int redv[kThreads / 64];
  __shared__ int s_best_it, s_used, s_nvalid;
  __shared__ int wave_off[kThreads / 64 + 1];
  __shared__ int s_running;
  const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = min(n_arr[b], stride);
  const int32_t* cb = counts + (size_t)b * H;

  // number of valid hypotheses (diagnostic) and, when not adaptive, the arg-max
  long long best_key = -1;
  int nvalid = 0;
  for (int it = tid; it < H; it += kThreads) {
    const int c = cb[it];
    if (c >= 0) {
      nvalid++;
      const long long key = ((long long)c << 32) | (long long)(0x7FFFFFFF - it);  // max count, then min it
      best_key = key > best_key ? key : best_key;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const long long other = __shfl_down(best_key, o);
    best_key = other > best_key ? other : best_key;
    nvalid += __shfl_down(nvalid, o);
  }
  if (lane == 0) {
    red[wid] = best_key;
    redv[wid] = nvalid;
  }
  __syncthreads();
  if (tid == 0) {
    long long k = red[0];
    int nv = redv[0];
    for (int w = 1; w < kThreads / 64; ++w) {
      k = red[w] > k ? red[w] : k;
      nv += redv[w];
    }
    s_nvalid = nv;
    if (!(adaptive & 1)) {
      s_best_it = k < 0 ? -1 : (0x7FFFFFFF - (int)(k & 0xFFFFFFFFLL));
      s_used = H;
    } else {
      // replay of the sequential loop: strictly-better update, stop when iterations >= k
      int best_count = -1, best_it = -1, iterations = 0, used = 0;
      double base = 1.0;
      for (int it = 0; it < H; ++it) {
        if (iterations > 0 && !sv_ransac_continue(base, iterations)) break;
        used = it + 1;
        const int c = cb[it];
        if (c < 0) continue;
        if (c > best_count) {
          best_count = c;
          best_it = it;
          base = (adaptive & 2) ? sv_adaptive_base6(c, n) : ((adaptive & 4) ? sv_adaptive_base2(c, n) : sv_adaptive_base(c, n));
        }
        iterations++;
      }
      s_best_it = best_it;
      s_used = used;
    }
    s_running = 0;
  }
  __syncthreads();
  const int best_it = s_best_it;
  const size_t base = (size_t)b * stride;
  if (best_it < 0) {
    for (int i = tid; i < n; i += kThreads) mask[base + i] = 0;
    if (tid < 12) T_out[(size_t)b * 12 + tid] = (tid == 0 || tid == 5 || tid == 10) ? 1.0 : 0.0;
    if (tid == 0) {
      n_inl[b] = 0;
      info[4 * b + 0] = -1;
      info[4 * b + 1] = s_used;
      info[4 * b + 2] = 1;
      info[4 * b + 3] = s_nvalid;
    }
    return;
  }
  const double* hp = hyp + ((size_t)b * H + best_it) * kHypDoubles;
  // exact inlier mask of the winner + ascending index list (stable compaction)
  for (int i0 = 0; i0 < n; i0 += kThreads) {
    const int i = i0 + tid;
    bool inl = false;
    if (i < n) {
      const size_t row = base + i;
      const int c = cam ? cam[row] : 0;
      const double* o = cam ? cam_off + 3 * c : kZero3;
      const double* rc = cam ? cam_rot + 9 * c : kEye9;
      inl = inlier_exact<IDENT>(hp, f[3 * row], f[3 * row + 1], f[3 * row + 2], p[3 * row], p[3 * row + 1],
                                p[3 * row + 2], o[0], o[1], o[2], rc, thr);
      mask[row] = inl ? 1 : 0;
    }
    const unsigned long long bal = __ballot(inl);
    if (lane == 0) wave_off[wid + 1] = __popcll(bal);
    __syncthreads();
    if (tid == 0) {
      wave_off[0] = s_running;
      for (int w = 0; w < kThreads / 64; ++w) wave_off[w + 1] += wave_off[w];
      s_running = wave_off[kThreads / 64];
    }
    __syncthreads();
    if (inl) {
      const int pos = wave_off[wid] + __popcll(bal & ((1ULL << lane) - 1ULL));
      inl_idx[base + pos] = i;
    }
    __syncthreads();
  }
  if (tid < 12) {
    const int r = tid >> 2, c = tid & 3;
    T_out[(size_t)b * 12 + tid] = (c < 3) ? hp[3 * r + c] : hp[9 + r];
  }
  if (tid == 0) {
    n_inl[b] = s_running;
    info[4 * b + 0] = best_it;
    info[4 * b + 1] = s_used;
    info[4 * b + 2] = 0;
    info[4 * b + 3] = s_nvalid;
  }
}

// ---- 5. refine ------------------------------------------------------------------------------
// Fixed-order workgroup sum: wave shuffles, then the four wave partials in order.
__device__ __forceinline__ double block_sum(double v, double* scratch /*[4]*/, int tid) {
  for (int o = 32; o > 0; o >>= 1) v = v + __shfl_down(v, o);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  return ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
}

__global__ __launch_bounds__(kThreads) void refine_kernel(
    const double* __restrict__ f, const double* __restrict__ p, const int32_t* __restrict__ cam,
    const double* __restrict__ cam_off, const double* __restrict__ cam_rot, const int32_t* __restrict__ n_arr,
    int stride, const int32_t* __restrict__ idx, const int32_t* __restrict__ m_arr, int max_lm_iter,
    double* __restrict__ T_io, double* __restrict__ cost_out, int32_t* __restrict__ iters_out) {
  SOSVO_LATENCY_BOUND_PRIO();
  __shared__ double scratch[4];
  __shared__ double sred[28][4];
  __shared__ double sx[6], sxn[6], sdx[6];
  __shared__ double sA[21], sg[6];
  __shared__ int s_flag;  // 0 continue tries, 1 accepted, 2 accepted+converged, 3 give up
  const int tid = threadIdx.x, b = blockIdx.x;
  const int n = min(n_arr[b], stride);
  const int cnt = idx ? min(m_arr[b], n) : n;
  const size_t base = (size_t)b * stride;
  double* T = T_io + (size_t)b * 12;
  if (tid == 0) {
    double R[9], t[3];
    sv_T_to_Rt(T, R, t);
    sx[0] = t[0];
    sx[1] = t[1];
    sx[2] = t[2];
    sv_rot2cayley(R, sx + 3);
  }
  __syncthreads();
  double lambda = SV_LM_LAMBDA0;
  double cost = 0.0;
  int it_done = 0;
  for (int it = 0; it < max_lm_iter; ++it) {
    double x[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) x[u] = sx[u];
    double A[21], g[6], c = 0.0;
#pragma unroll
    for (int a = 0; a < 21; ++a) A[a] = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) g[a] = 0.0;
    for (int q = tid; q < cnt; q += kThreads) {
      const int i = idx ? idx[base + q] : q;
      const size_t row = base + i;
      const int cc = cam ? cam[row] : 0;
      const double* o = cam ? cam_off + 3 * cc : kZero3;
      const double* rc = cam ? cam_rot + 9 * cc : kEye9;
      double r, J[6], Hq[21];
      sv_residual_jac(x, f + 3 * row, p + 3 * row, o, rc, &r, J, Hq);
      c += r * r;
      int a = 0;
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        g[u] += J[u] * r;
#pragma unroll
        for (int v = u; v < 6; ++v) {
          A[a] += (J[u] * J[v]) + Hq[a];
          a++;
        }
      }
    }
    // one fused reduction of the 28 sums (same order as block_sum: wave tree, then w0 + w1 + w2 + w3)
    {
      double wv[28];
      wv[27] = c;
#pragma unroll
      for (int a = 0; a < 21; ++a) wv[a] = A[a];
#pragma unroll
      for (int a = 0; a < 6; ++a) wv[21 + a] = g[a];
#pragma unroll
      for (int a = 0; a < 28; ++a)
        for (int o = 32; o > 0; o >>= 1) wv[a] = wv[a] + __shfl_down(wv[a], o);
      __syncthreads();
      if ((tid & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 28; ++a) sred[a][tid >> 6] = wv[a];
      }
      __syncthreads();
      if (tid < 21) sA[tid] = ((sred[tid][0] + sred[tid][1]) + sred[tid][2]) + sred[tid][3];
      if (tid >= 21 && tid < 27) sg[tid - 21] = ((sred[tid][0] + sred[tid][1]) + sred[tid][2]) + sred[tid][3];
      cost = ((sred[27][0] + sred[27][1]) + sred[27][2]) + sred[27][3];
    }
    __syncthreads();
    int accepted = 0, converged = 0;
    for (int tries = 0; tries < SV_LM_MAX_TRIES; ++tries) {
      if (tid == 0) {
        double dx[6];
        if (sv_solve_damped(sA, sg, lambda, dx)) {
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            sdx[u] = dx[u];
            sxn[u] = sx[u] + dx[u];
          }
          // a step below the resolution we care about ends the refinement before it is evaluated
          double dxn = 0.0, xnn = 0.0;
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            dxn += dx[u] * dx[u];
            xnn += sxn[u] * sxn[u];
          }
          s_flag = (sqrt(dxn) <= SV_LM_XTOL * (sqrt(xnn) + SV_LM_XTOL)) ? 4 : 0;
        } else {
          s_flag = 3;
        }
      }
      __syncthreads();
      if (s_flag == 3) {  // singular: raise lambda and retry (uniform)
        lambda *= 10.0;
        __syncthreads();
        continue;
      }
      if (s_flag == 4) {  // uniform
        converged = 1;
        __syncthreads();
        break;
      }
      double xn[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) xn[u] = sxn[u];
      double cn = 0.0;
      for (int q = tid; q < cnt; q += kThreads) {
        const int i = idx ? idx[base + q] : q;
        const size_t row = base + i;
        const int cc = cam ? cam[row] : 0;
        const double* o = cam ? cam_off + 3 * cc : kZero3;
        const double* rc = cam ? cam_rot + 9 * cc : kEye9;
        const double r = sv_residual(xn, f + 3 * row, p + 3 * row, o, rc);
        cn += r * r;
      }
      cn = block_sum(cn, scratch, tid);
      if (cn < cost) {
        converged = ((cost - cn) <= SV_LM_FTOL * cost);
        __syncthreads();
        if (tid == 0) {
#pragma unroll
          for (int u = 0; u < 6; ++u) sx[u] = xn[u];
        }
        cost = cn;
        lambda *= 0.1;
        if (lambda < 1e-15) lambda = 1e-15;
        accepted = 1;
        __syncthreads();
        break;
      }
      lambda *= 10.0;
      __syncthreads();
    }
    it_done = it + 1;
    if (!accepted || converged) break;
  }
  __syncthreads();
  if (tid == 0) {
    double R[9], t[3];
    sv_cayley2rot(sx + 3, R);
    t[0] = sx[0];
    t[1] = sx[1];
    t[2] = sx[2];
    sv_Rt_to_T(R, t, T);
    if (cost_out) cost_out[b] = cost;
    if (iters_out) iters_out[b] = it_done;
  }
}

}  // namespace

extern "C" {

int32_t sosvo_ransac_abs_pose(sosvo_ctx* ctx, const double* f, const double* p, const int32_t* cam,
                              const double* cam_off, const double* cam_rot, int32_t ncam, int32_t flags,
                              const int32_t* n, int32_t nprob, int32_t stride, double thr, int32_t max_iter,
                              int32_t adaptive, uint64_t seed, double* T_out, uint8_t* inlier_mask,
                              int32_t* inlier_idx, int32_t* n_inliers, int32_t* info, int32_t* hyp_counts) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, f && p && n && T_out && inlier_mask && inlier_idx && n_inliers && info, "null pointer");
  SOSVO_REQUIRE(ctx, cam == nullptr || (cam_off && cam_rot), "cam given without cam_off / cam_rot");
  SOSVO_REQUIRE(ctx, ncam >= 1 && ncam <= kMaxCam, "ncam out of range (1..8)");
  SOSVO_REQUIRE(ctx, nprob >= 0 && nprob <= 65535, "nprob out of range");
  SOSVO_REQUIRE(ctx, stride > 0 && stride <= (1 << 20), "stride out of range");
  SOSVO_REQUIRE(ctx, max_iter > 0 && max_iter <= (1 << 20), "max_iter out of range");
  SOSVO_REQUIRE(ctx, thr > 0.0, "threshold must be positive");
  if (nprob == 0) return SOSVO_OK;
  const int H = max_iter;
  // workspace carve-up
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t o_perm = carve(sizeof(int32_t) * (size_t)nprob * stride);
  const size_t o_cinfo = carve(sizeof(int32_t) * (size_t)nprob * (2 * kMaxCam + 1));
  const size_t o_hyp = carve(sizeof(double) * (size_t)nprob * H * kHypDoubles);
  const size_t o_counts = carve(sizeof(int32_t) * (size_t)nprob * H);
  const bool epnp_vec = (flags & SOSVO_FLAG_EPNP) != 0;
  const size_t o_vv = carve(epnp_vec ? sizeof(double) * (size_t)nprob * H * kEpnpVecDoubles : 0);
  int32_t rc = sosvo_ws_reserve(ctx, off);
  if (rc != SOSVO_OK) return rc;
  char* ws = (char*)ctx->ws;
  int32_t* perm = (int32_t*)(ws + o_perm);
  int32_t* cinfo = (int32_t*)(ws + o_cinfo);
  double* hyp = (double*)(ws + o_hyp);
  int32_t* counts = hyp_counts ? hyp_counts : (int32_t*)(ws + o_counts);

  SOSVO_LAUNCH(ctx,ransac_prepare_kernel, dim3(nprob), dim3(kThreads), 0, ctx->stream, cam, n, stride,
                     cam ? ncam : 1, perm, cinfo);
  SOSVO_LAUNCH_CHECK(ctx);
  const int epnp = (flags & SOSVO_FLAG_EPNP) != 0;
  SOSVO_REQUIRE(ctx, !epnp || cam == nullptr, "SOSVO_FLAG_EPNP is for central problems (cam == NULL)");
  const int twopt = (flags & SOSVO_FLAG_TWOPT) != 0 && !epnp && !(flags & SOSVO_FLAG_GP3P);
  SOSVO_REQUIRE(ctx, !twopt || cam == nullptr, "SOSVO_FLAG_TWOPT is for central problems (cam == NULL)");
  if (epnp) adaptive = (adaptive ? 1 : 0) | 2;  // bit 1: the adaptive stop uses 6-point samples
  if (twopt) adaptive = (adaptive ? 1 : 0) | 4;  // bit 2: ... 2-point samples
  if (epnp) {
    double* vvbuf = (double*)(ws + o_vv);
    SOSVO_LAUNCH(ctx, ransac_epnp_eigen_kernel, dim3(hyp_grid(nprob, cdiv(H, 64))), dim3(64), 0, ctx->stream, f, p, n, nprob, stride, H,
                 seed, vvbuf, counts);
    SOSVO_LAUNCH(ctx, ransac_epnp_pose_kernel, dim3(hyp_grid(nprob, cdiv(H, 64))), dim3(64), 0, ctx->stream, f, p, n, nprob, stride, H,
                 seed, vvbuf, hyp, counts);
  } else if (twopt) {
    SOSVO_LAUNCH(ctx, ransac_hyp_twopt_kernel, dim3(hyp_grid(nprob, cdiv(H, 64))), dim3(64), 0, ctx->stream, f, p, n, nprob, stride, H, seed,
                 hyp, counts);
  } else if (flags & SOSVO_FLAG_GP3P) {
    SOSVO_LAUNCH(ctx, ransac_hyp_gp3p_kernel, dim3(hyp_grid(nprob, cdiv(H, 64))), dim3(64), 0, ctx->stream, f, p, cam, cam_off, cam_rot,
                 n, nprob, stride, H, seed, hyp, counts);
  } else
    SOSVO_LAUNCH(ctx, ransac_hyp_kernel, dim3(hyp_grid(nprob, cdiv(H, 64))), dim3(64), 0, ctx->stream, f, p, cam, cam_off, cam_rot, n,
                 nprob, stride, H, seed, perm, cinfo, hyp, counts);
  SOSVO_LAUNCH_CHECK(ctx);

  const bool ident = (flags & SOSVO_FLAG_CAM_ROT_IDENTITY) != 0 || cam == nullptr;
  const int fast_ok = thr < 0.5 ? 1 : 0;
  // hypothesis chunks: enough workgroups to fill 256 CUs a few times over
  // (point blocks beyond n exit at once, so the live workgroups are ~hchunks * nprob * cdiv(n, points per block))
  const int ppt = ident ? 4 : 1;
  const int gx = cdiv(stride, kThreads * ppt);
  // tier 1 in front: identity camera rotations and a threshold below 0.5 (the squared-cosine form of the test)
  const bool t1 = ident && fast_ok && !ctx->hint_score_fp64_only;
  // workgroups per launch: enough to fill the chip several times over; the tier-1 form amortises its per-lane set-up
  // (loads, conversions, error constants) over twice the hypotheses (measured: 0.50 -> 0.47 ms per 256 x 1285 x 2000)
  const int wg_target = t1 ? 4096 : 8192;
  int hchunks = cdiv(wg_target, nprob);
  if (hchunks < 1) hchunks = 1;
  int hspan = cdiv(H, hchunks);  // hypotheses per workgroup, in LDS chunks of at most kScoreHypChunkMax
  if (hspan < 32) hspan = H < 32 ? H : 32;
  hchunks = cdiv(H, hspan);
  const int hchunk = cdiv(hspan, cdiv(hspan, kScoreHypChunkMax));
  dim3 grid(hchunks, gx, nprob);
#define SOSVO_SCORE_ARGS grid, dim3(kThreads), 0, ctx->stream, f, p, cam, cam_off, cam_rot, n, stride, H, hchunk, hspan, thr, fast_ok, hyp, counts
  if (t1)
    SOSVO_LAUNCH(ctx,(ransac_score_kernel<true, 4, true>), SOSVO_SCORE_ARGS);
  else if (ident)
    SOSVO_LAUNCH(ctx,(ransac_score_kernel<true, 4, false>), SOSVO_SCORE_ARGS);
  else
    SOSVO_LAUNCH(ctx,(ransac_score_kernel<false, 1, false>), SOSVO_SCORE_ARGS);
#undef SOSVO_SCORE_ARGS
  SOSVO_LAUNCH_CHECK(ctx);
  if (ident)
    SOSVO_LAUNCH(ctx,(ransac_select_kernel<true>), dim3(nprob), dim3(kThreads), 0, ctx->stream, f, p, cam,
                       cam_off, cam_rot, n, stride, H, thr, adaptive, hyp, counts, T_out, inlier_mask, inlier_idx,
                       n_inliers, info);
  else
    SOSVO_LAUNCH(ctx,(ransac_select_kernel<false>), dim3(nprob), dim3(kThreads), 0, ctx->stream, f, p, cam,
                       cam_off, cam_rot, n, stride, H, thr, adaptive, hyp, counts, T_out, inlier_mask, inlier_idx,
                       n_inliers, info);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_refine_abs_pose(sosvo_ctx* ctx, const double* f, const double* p, const int32_t* cam,
                              const double* cam_off, const double* cam_rot, int32_t ncam, const int32_t* n,
                              int32_t nprob, int32_t stride, const int32_t* idx, const int32_t* m,
                              int32_t max_lm_iter, double* T_io, double* cost_out, int32_t* iters_out) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, f && p && n && T_io, "null pointer");
  SOSVO_REQUIRE(ctx, cam == nullptr || (cam_off && cam_rot), "cam given without cam_off / cam_rot");
  SOSVO_REQUIRE(ctx, (idx == nullptr) == (m == nullptr), "idx and m go together");
  SOSVO_REQUIRE(ctx, ncam >= 1 && ncam <= kMaxCam, "ncam out of range (1..8)");
  SOSVO_REQUIRE(ctx, nprob >= 0 && nprob <= 65535, "nprob out of range");
  SOSVO_REQUIRE(ctx, stride > 0 && stride <= (1 << 20), "stride out of range");
  SOSVO_REQUIRE(ctx, max_lm_iter > 0 && max_lm_iter <= 10000, "max_lm_iter out of range");
  if (nprob == 0) return SOSVO_OK;
  SOSVO_LAUNCH(ctx,refine_kernel, dim3(nprob), dim3(kThreads), 0, ctx->stream, f, p, cam, cam_off, cam_rot, n,
                     stride, idx, m, max_lm_iter, T_io, cost_out, iters_out);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

}  // extern "C"
