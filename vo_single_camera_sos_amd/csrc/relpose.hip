// 2D-2D relative-pose RANSAC between two sets of bearing vectors (not reached by the VO drivers).
//
// Reference call site replaced (the arithmetic lives in OpenGV, outside the tree; the SCORE is restated by the reference
// itself):   omnistereo/pose_est_tools.py:78   pyopengv.relative_pose_ransac(b1, b2, algorithm, threshold, max_iterations)
//            omnistereo/pose_est_tools.py:150-203  score of a correspondence under a relative pose
//
// Same decomposition as the absolute-pose RANSAC (ransac.hip): hypotheses for every (problem, iteration) in parallel, one
// lane each (relpose_core.h: the oracle's text) -> the iters x N scoring loop, a lane keeps its correspondence in VGPRs and
// reads the chunk's hypotheses as LDS broadcasts, the inlier decision is a wave ballot + scalar popcount -> the sequential
// semantics of sac::Ransac replayed over the counts, exact mask + ascending index list of the winner.  The score needs a
// triangulation, two square roots and three divisions per (correspondence, hypothesis); it is evaluated with the oracle's
// operations in the oracle's order (no contraction), so counts, winner, mask and pose bits equal the sequential run.
#include "common.h"
#include "ransac_core.h"
#include "epnp_core.h"
#include "relpose_core.h"

namespace {

constexpr int kRelHypDoubles = 12;  // R[9], t[3]
constexpr int kRelThreads = 256;
constexpr int kRelChunkMax = 128;

__device__ __forceinline__ uint64_t rel_problem_seed(uint64_t seed, int b) { return seed + (uint64_t)b; }

__global__ __launch_bounds__(64) void relpose_hyp_kernel(const double* __restrict__ f1, const double* __restrict__ f2,
                                                         const int32_t* __restrict__ n_arr, int stride, int H, int algorithm,
                                                         uint64_t seed, double* __restrict__ hyp,
                                                         int32_t* __restrict__ counts) {
  const int b = blockIdx.y;
  const int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= H) return;
  const int n = min(n_arr[b], stride);
  const size_t base = (size_t)b * stride;
  double R[9], t[3];
  const int ok = sv_rel_hypothesis(f1 + 3 * base, f2 + 3 * base, n, algorithm, rel_problem_seed(seed, b), (uint64_t)it, R, t);
  double* h = hyp + ((size_t)b * H + it) * kRelHypDoubles;
  if (ok) {
#pragma unroll
    for (int k = 0; k < 9; ++k) h[k] = R[k];
    h[9] = t[0];
    h[10] = t[1];
    h[11] = t[2];
  } else {
    h[0] = __longlong_as_double(0x7FF8000000000000LL);
  }
  counts[(size_t)b * H + it] = ok ? 0 : -1;
}

// grid (hypothesis chunk, point block, problem): the dimension whose workgroups exit early (point blocks beyond n) is
// not the fastest one, so the live workgroups spread over the XCDs.
__global__ __launch_bounds__(kRelThreads) void relpose_score_kernel(const double* __restrict__ f1,
                                                                    const double* __restrict__ f2,
                                                                    const int32_t* __restrict__ n_arr, int stride, int H,
                                                                    int hchunk, double thr, const double* __restrict__ hyp,
                                                                    int32_t* __restrict__ counts) {
  __shared__ int lcnt[kRelChunkMax];
  __shared__ double shyp[kRelChunkMax][kRelHypDoubles];
  const int tid = threadIdx.x, b = blockIdx.z, lane = tid & 63;
  const int n = min(n_arr[b], stride);
  const int i = blockIdx.y * kRelThreads + tid;
  if ((int)blockIdx.y * kRelThreads >= n) return;
  const int h0 = blockIdx.x * hchunk, h1 = min(H, h0 + hchunk);
  for (int k = tid; k < hchunk; k += kRelThreads) lcnt[k] = 0;
  const double* hb = hyp + ((size_t)b * H + h0) * kRelHypDoubles;
  for (int k = tid; k < (h1 - h0) * kRelHypDoubles; k += kRelThreads) shyp[k / kRelHypDoubles][k % kRelHypDoubles] = hb[k];
  const bool valid = i < n;
  const size_t row = (size_t)b * stride + (valid ? i : 0);
  double a[3], c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    a[k] = f1[3 * row + k];
    c[k] = f2[3 * row + k];
  }
  __syncthreads();
  for (int hh = 0; hh < h1 - h0; ++hh) {
    const double* hp = shyp[hh];
    if (!(hp[0] == hp[0])) continue;  // failed minimal solve (uniform)
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = hp[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = hp[9 + k];
    const bool inl = valid && (sv_rel_score(R, t, a, c) < thr);
    const int cnt = __popcll(__ballot(inl));
    if (cnt && lane == 0) atomicAdd(&lcnt[hh], cnt);
  }
  __syncthreads();
  for (int k = tid; k < h1 - h0; k += kRelThreads) {
    const int v = lcnt[k];
    if (v) atomicAdd(&counts[(size_t)b * H + h0 + k], v);
  }
}

__global__ __launch_bounds__(kRelThreads) void relpose_select_kernel(const double* __restrict__ f1,
                                                                     const double* __restrict__ f2,
                                                                     const int32_t* __restrict__ n_arr, int stride, int H,
                                                                     double thr, int adaptive, int sample_size,
                                                                     const double* __restrict__ hyp,
                                                                     const int32_t* __restrict__ counts,
                                                                     double* __restrict__ T_out, uint8_t* __restrict__ mask,
                                                                     int32_t* __restrict__ inl_idx, int32_t* __restrict__ n_inl,
                                                                     int32_t* __restrict__ info) {
  __shared__ int s_best_it, s_used, s_nvalid, s_running;
  __shared__ int wave_off[kRelThreads / 64 + 1];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int n = min(n_arr[b], stride);
  const int32_t* cb = counts + (size_t)b * H;
  if (tid == 0) {
    // the sequential loop of sac::Ransac over the counts: strictly-better update, failed solves skipped, adaptive stop
    int best_count = -1, best_it = -1, iterations = 0, used = 0, nvalid = 0;
    double base = 1.0;
    for (int it = 0; it < H; ++it) {
      if (adaptive && iterations > 0 && !sv_ransac_continue(base, iterations)) break;
      used = it + 1;
      const int c = cb[it];
      if (c < 0) continue;
      nvalid++;
      if (c > best_count) {
        best_count = c;
        best_it = it;
        base = sv_adaptive_base_k(c, n, sample_size);
      }
      iterations++;
    }
    s_best_it = best_it;
    s_used = used;
    s_nvalid = nvalid;
    s_running = 0;
  }
  __syncthreads();
  const int best_it = s_best_it;
  const size_t base = (size_t)b * stride;
  if (best_it < 0) {
    for (int i = tid; i < n; i += kRelThreads) mask[base + i] = 0;
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
  const double* hp = hyp + ((size_t)b * H + best_it) * kRelHypDoubles;
  double R[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = hp[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = hp[9 + k];
  for (int i0 = 0; i0 < n; i0 += kRelThreads) {
    const int i = i0 + tid;
    bool inl = false;
    if (i < n) {
      const size_t row = base + i;
      inl = sv_rel_score(R, t, f1 + 3 * row, f2 + 3 * row) < thr;
      mask[row] = inl ? 1 : 0;
    }
    const int pos = sosvo_block_compact_pos(inl, wave_off, &s_running, tid);
    if (inl) inl_idx[base + pos] = i;
  }
  __syncthreads();
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

}  // namespace

extern "C" int32_t sosvo_ransac_rel_pose(sosvo_ctx* ctx, const double* f1, const double* f2, const int32_t* n, int32_t nprob,
                                         int32_t stride, int32_t algorithm, double thr, int32_t max_iter, int32_t adaptive,
                                         uint64_t seed, double* T_out, uint8_t* inlier_mask, int32_t* inlier_idx,
                                         int32_t* n_inliers, int32_t* info, int32_t* hyp_counts) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, f1 && f2 && n && T_out && inlier_mask && inlier_idx && n_inliers && info, "null pointer");
  SOSVO_REQUIRE(ctx, algorithm == SOSVO_REL_EIGHTPT || algorithm == SOSVO_REL_SEVENPT || algorithm == SOSVO_REL_FIVEPT,
                "unknown algorithm");
  SOSVO_REQUIRE(ctx, nprob >= 0 && nprob <= 65535, "nprob out of range");
  SOSVO_REQUIRE(ctx, stride > 0 && stride <= (1 << 20), "stride out of range");
  SOSVO_REQUIRE(ctx, max_iter > 0 && max_iter <= (1 << 20), "max_iter out of range");
  SOSVO_REQUIRE(ctx, thr > 0.0, "threshold must be positive");
  if (nprob == 0) return SOSVO_OK;
  const int H = max_iter;
  const size_t hyp_bytes = (sizeof(double) * (size_t)nprob * H * kRelHypDoubles + 255) & ~(size_t)255;
  const size_t cnt_bytes = sizeof(int32_t) * (size_t)nprob * H;
  int32_t rc = sosvo_ws_reserve(ctx, hyp_bytes + cnt_bytes);
  if (rc != SOSVO_OK) return rc;
  double* hyp = (double*)ctx->ws;
  int32_t* counts = hyp_counts ? hyp_counts : (int32_t*)((char*)ctx->ws + hyp_bytes);
  SOSVO_LAUNCH(ctx, relpose_hyp_kernel, dim3(cdiv(H, 64), nprob), dim3(64), 0, ctx->stream, f1, f2, n, stride, H, algorithm, seed,
               hyp, counts);
  SOSVO_LAUNCH_CHECK(ctx);
  int hchunk = cdiv(H, cdiv(4096, nprob));
  if (hchunk > kRelChunkMax) hchunk = kRelChunkMax;
  if (hchunk < 16) hchunk = H < 16 ? H : 16;
  SOSVO_LAUNCH(ctx, relpose_score_kernel, dim3(cdiv(H, hchunk), cdiv(stride, kRelThreads), nprob), dim3(kRelThreads), 0,
               ctx->stream, f1, f2, n, stride, H, hchunk, thr, hyp, counts);
  SOSVO_LAUNCH_CHECK(ctx);
  SOSVO_LAUNCH(ctx, relpose_select_kernel, dim3(nprob), dim3(kRelThreads), 0, ctx->stream, f1, f2, n, stride, H, thr,
               adaptive ? 1 : 0, algorithm == SOSVO_REL_SEVENPT ? 9 : 8, hyp, counts, T_out, inlier_mask, inlier_idx, n_inliers, info);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}
