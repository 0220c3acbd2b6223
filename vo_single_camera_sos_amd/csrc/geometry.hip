// Geometry stages of the hot path: pano pixel -> bearing (a7, a8), midpoint triangulation (a9),
// range filter (a10), RGB-D back-projection (a12), and the two fused "assemble" kernels that
// replace the Python list/fancy-indexing glue between matching and RANSAC:
//   stereo_assemble: OmniStereoModel.match_features_panoramic_top_bottom (camera_models.py:3027-3101)
//                    + StereoPanoramicFrame.establish_stereo_correspondences (pose_est_tools.py:339-397)
//   f2f_assemble:    match_features_frame_to_frame (pose_est_tools.py:211-269)
//                    + the stacking of TrackerStereoSE3.track_frame (pose_est_tools.py:752-778)
// Both keep the reference's ordering (bucket order, then ascending distance with ties in query
// order; top view before bottom view) by stable compaction: wave ballot prefix + a running base.
// All are HBM/latency-bound elementwise work: one lane per correspondence, coalesced rows.
#include "common.h"
#include "geom_core.h"

namespace {

constexpr int kThreads = 256;

__global__ void pano_to_bearing_kernel(const double* __restrict__ uv, int n, double cols, double rows, double px,
                                       double hmax, double* __restrict__ az, double* __restrict__ el,
                                       double* __restrict__ b3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double pano[4] = {cols, rows, px, hmax};
  double a, e, b[3];
  sv_pano_angles(uv[2 * i], uv[2 * i + 1], pano, &a, &e);
  sv_bearing(a, e, b);
  if (az) az[i] = a;
  if (el) el[i] = e;
  if (b3) {
    b3[3 * i] = b[0];
    b3[3 * i + 1] = b[1];
    b3[3 * i + 2] = b[2];
  }
}

struct F3x2 {
  double F1[3], F2[3];
};

__global__ void triangulate_kernel(const double* __restrict__ az1, const double* __restrict__ el1,
                                   const double* __restrict__ az2, const double* __restrict__ el2, int n, F3x2 foci,
                                   double* __restrict__ X) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x[3];
  sv_triangulate(az1[i], el1[i], az2[i], el2[i], foci.F1, foci.F2, x);
  X[3 * i] = x[0];
  X[3 * i + 1] = x[1];
  X[3 * i + 2] = x[2];
}

struct Rt12 {
  double t[3], R[9];
};

__global__ void triangulate2_kernel(const double* __restrict__ b1, const double* __restrict__ b2, int n, Rt12 rt,
                                    double* __restrict__ X) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double f1[3] = {b1[3 * i], b1[3 * i + 1], b1[3 * i + 2]}, f2[3] = {b2[3 * i], b2[3 * i + 1], b2[3 * i + 2]};
  double x[3];
  sv_triangulate2(f1, f2, rt.t, rt.R, x);
  X[3 * i] = x[0];
  X[3 * i + 1] = x[1];
  X[3 * i + 2] = x[2];
}

__global__ void range_filter_kernel(const double* __restrict__ X, int n, double min_range, double max_range,
                                    uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x[3] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]};
  ok[i] = sv_range_ok(x, min_range, max_range) ? 1 : 0;
}

struct Intr {
  double fx, fy, cx, cy, fl;
};

// camera_models.py:781-799 (radial depth -> Z; focal_length * depth stays float32), :845-860, :203-212
__global__ void rgbd_backproject_kernel(const float* __restrict__ depth, int rows, int cols,
                                        const int32_t* __restrict__ u, const int32_t* __restrict__ v, int n, Intr k,
                                        int depth_is_Z, double* __restrict__ xyz, double* __restrict__ bearing) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ui = min(max(u[i], 0), cols - 1), vi = min(max(v[i], 0), rows - 1);
  const float dv = depth[(size_t)vi * cols + ui];
  double d = (double)dv;
  if (!depth_is_Z) {
    const double xi = (k.fl / k.fx) * ((double)ui - k.cx), yi = (k.fl / k.fy) * ((double)vi - k.cy), zi = k.fl;
    const float fd = (float)k.fl * dv;
    d = (double)fd / sqrt(xi * xi + yi * yi + zi * zi);
  }
  const double Z = (d != 0.0) ? d : sv_nan();
  const double X = ((double)ui - k.cx) * Z / k.fx, Y = ((double)vi - k.cy) * Z / k.fy;
  xyz[3 * i] = X;
  xyz[3 * i + 1] = Y;
  xyz[3 * i + 2] = Z;
  const double nrm = sqrt(X * X + Y * Y + Z * Z);
  bearing[3 * i] = X / nrm;
  bearing[3 * i + 1] = Y / nrm;
  bearing[3 * i + 2] = Z / nrm;
}

// Stable position of a lane's element among the workgroup's valid elements of this round.
// Returns the position (valid lanes only meaningful) and advances *s_running by the round total.
__device__ __forceinline__ int block_compact_pos(bool valid, int* wave_off /*[5]*/, int* s_running, int tid) {
  const int lane = tid & 63, wid = tid >> 6;
  const unsigned long long bal = __ballot(valid);
  __syncthreads();
  if (lane == 0) wave_off[wid + 1] = __popcll(bal);
  __syncthreads();
  if (tid == 0) {
    wave_off[0] = *s_running;
    for (int w = 0; w < kThreads / 64; ++w) wave_off[w + 1] += wave_off[w];
    *s_running = wave_off[kThreads / 64];
  }
  __syncthreads();
  return wave_off[wid] + __popcll(bal & ((1ULL << lane) - 1ULL));
}

__global__ __launch_bounds__(kThreads) void stereo_assemble_kernel(
    sosvo_rig rig, const float* __restrict__ kp_top, const float* __restrict__ kp_bot,
    const uint4* __restrict__ desc_top, const uint4* __restrict__ desc_bot, const int32_t* __restrict__ n_top,
    const int32_t* __restrict__ n_bot, const uint32_t* __restrict__ keys, const int32_t* __restrict__ order, int nmask,
    int cap, int out_cap, float* __restrict__ m_top, float* __restrict__ m_bot, uint4* __restrict__ d_top,
    uint4* __restrict__ d_bot, double* __restrict__ X, double* __restrict__ b_top, double* __restrict__ b_bot,
    int32_t* __restrict__ M, int32_t* __restrict__ n_cand) {
  SOSVO_LATENCY_BOUND_PRIO();
  __shared__ int wave_off[kThreads / 64 + 1];
  __shared__ int s_running;
  const int tid = threadIdx.x, fr = blockIdx.x;
  if (tid == 0) s_running = 0;
  __syncthreads();
  int cand_total = 0;
  for (int m = 0; m < nmask; ++m) {
    const int p = fr * nmask + m;
    const int nq = min(n_bot[p], cap), nt = min(n_top[p], cap);
    if (nq == 0 || nt == 0) continue;                                 // camera_models.py:3039
    const int good = (int)(rig.pct_good_matches * (double)nq);        // camera_models.py:3045
    cand_total += good;
    const size_t pb = (size_t)p * cap;
    for (int r0 = 0; r0 < good; r0 += kThreads) {
      const int r = r0 + tid;
      bool valid = false;
      int q = 0, t = 0;
      double xyz[3], bt[3], bb[3];
      float ut = 0, vt = 0, ub = 0, vb = 0;
      if (r < good) {
        q = order[pb + r];                                            // query = bottom (:3042)
        t = (int)(keys[pb + q] & SOSVO_KEY_IDX_MASK);                 // train = top
        ut = kp_top[2 * (pb + t)];
        vt = kp_top[2 * (pb + t) + 1];
        ub = kp_bot[2 * (pb + q)];
        vb = kp_bot[2 * (pb + q) + 1];
        valid = sv_pixel_gate((double)ut, (double)vt, (double)ub, (double)vb, rig.stereo_min_disp, rig.stereo_max_hdiff);
        double az1, el1, az2, el2;
        sv_pano_angles((double)ut, (double)vt, rig.pano_top, &az1, &el1);
        sv_pano_angles((double)ub, (double)vb, rig.pano_bot, &az2, &el2);
        sv_bearing(az1, el1, bt);
        sv_bearing(az2, el2, bb);
        sv_triangulate(az1, el1, az2, el2, rig.F_top, rig.F_bot, xyz);
        valid = valid && sv_range_ok(xyz, rig.min_range, rig.max_range);
      }
      const int pos = block_compact_pos(valid, wave_off, &s_running, tid);
      if (valid && pos < out_cap) {
        const size_t o = (size_t)fr * out_cap + pos;
        m_top[2 * o] = ut;
        m_top[2 * o + 1] = vt;
        m_bot[2 * o] = ub;
        m_bot[2 * o + 1] = vb;
        d_top[2 * o] = desc_top[2 * (pb + t)];
        d_top[2 * o + 1] = desc_top[2 * (pb + t) + 1];
        d_bot[2 * o] = desc_bot[2 * (pb + q)];
        d_bot[2 * o + 1] = desc_bot[2 * (pb + q) + 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          X[3 * o + k] = xyz[k];
          b_top[3 * o + k] = bt[k];
          b_bot[3 * o + k] = bb[k];
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    M[fr] = min(s_running, out_cap);
    if (n_cand) n_cand[fr] = cand_total;
  }
}

__global__ __launch_bounds__(kThreads) void f2f_assemble_kernel(
    sosvo_rig rig, const float* __restrict__ m_top, const float* __restrict__ m_bot, const double* __restrict__ X,
    const double* __restrict__ b_top, const double* __restrict__ b_bot, const int32_t* __restrict__ M, int frame_cap,
    const int32_t* __restrict__ ref_frame, const int32_t* __restrict__ cur_frame,
    const uint32_t* __restrict__ keys_top, const int32_t* __restrict__ order_top,
    const uint32_t* __restrict__ keys_bot, const int32_t* __restrict__ order_bot, int corr_cap,
    double* __restrict__ f, double* __restrict__ p, int32_t* __restrict__ cam, int32_t* __restrict__ corr_q,
    int32_t* __restrict__ corr_t, int32_t* __restrict__ n, int32_t* __restrict__ n_topview) {
  SOSVO_LATENCY_BOUND_PRIO();
  __shared__ int wave_off[kThreads / 64 + 1];
  __shared__ int s_running;
  const int tid = threadIdx.x, pr = blockIdx.x;
  if (tid == 0) s_running = 0;
  __syncthreads();
  const int fr_ref = ref_frame[pr], fr_cur = cur_frame[pr];
  const int nq = min(M[fr_cur], frame_cap), nt = min(M[fr_ref], frame_cap);
  const int good = (nt > 0) ? (int)(rig.pct_good_matches * (double)nq) : 0;  // pose_est_tools.py:225
  const size_t kb = (size_t)pr * frame_cap;
  for (int view = 0; view < 2; ++view) {
    const uint32_t* keys = view == 0 ? keys_top : keys_bot;
    const int32_t* order = view == 0 ? order_top : order_bot;
    const float* mm = view == 0 ? m_top : m_bot;
    const double* bb = view == 0 ? b_top : b_bot;
    for (int r0 = 0; r0 < good; r0 += kThreads) {
      const int r = r0 + tid;
      bool valid = false;
      int q = 0, t = 0;
      if (r < good) {
        q = order[kb + r];                                  // query = current frame (:215)
        t = (int)(keys[kb + q] & SOSVO_KEY_IDX_MASK);       // train = reference (key)frame
        valid = true;
        if (rig.f2f_max_hdiff >= 0) {                       // :245-247, gate inside common_cv.py:177
          const double u_train = (double)mm[2 * ((size_t)fr_ref * frame_cap + t)];
          const double u_query = (double)mm[2 * ((size_t)fr_cur * frame_cap + q)];
          valid = sv_pixel_gate(u_train, 0.0, u_query, 0.0, -1.0, rig.f2f_max_hdiff);
        }
      }
      const int pos = block_compact_pos(valid, wave_off, &s_running, tid);
      if (valid && pos < corr_cap) {
        const size_t o = (size_t)pr * corr_cap + pos;
        const size_t qrow = (size_t)fr_cur * frame_cap + q, trow = (size_t)fr_ref * frame_cap + t;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          f[3 * o + k] = bb[3 * qrow + k];                  // bearings of the CURRENT frame (:752,:755)
          p[3 * o + k] = X[3 * trow + k];                   // 3-D points of the REFERENCE frame (:753,:756)
        }
        cam[o] = view;
        corr_q[o] = q;
        corr_t[o] = t;
      }
    }
    __syncthreads();
    if (view == 0 && tid == 0 && n_topview) n_topview[pr] = min(s_running, corr_cap);
  }
  __syncthreads();
  if (tid == 0) n[pr] = min(s_running, corr_cap);
}

}  // namespace

extern "C" {

int32_t sosvo_pano_to_bearing(sosvo_ctx* ctx, const double* uv, int32_t n, double cols, double rows,
                              double pixel_size, double cyl_height_max, double* az, double* el, double* bearing) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, uv && n >= 0, "bad arguments");
  if (n == 0) return SOSVO_OK;
  SOSVO_LAUNCH(ctx,pano_to_bearing_kernel, dim3(cdiv(n, kThreads)), dim3(kThreads), 0, ctx->stream, uv, n, cols, rows,
                     pixel_size, cyl_height_max, az, el, bearing);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_triangulate_midpoint(sosvo_ctx* ctx, const double* az_top, const double* el_top, const double* az_bot,
                                   const double* el_bot, int32_t n, const double* F_top_host,
                                   const double* F_bot_host, double* X) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, az_top && el_top && az_bot && el_bot && F_top_host && F_bot_host && X && n >= 0, "bad arguments");
  if (n == 0) return SOSVO_OK;
  F3x2 foci;
  for (int k = 0; k < 3; ++k) {
    foci.F1[k] = F_top_host[k];
    foci.F2[k] = F_bot_host[k];
  }
  SOSVO_LAUNCH(ctx,triangulate_kernel, dim3(cdiv(n, kThreads)), dim3(kThreads), 0, ctx->stream, az_top, el_top, az_bot,
                     el_bot, n, foci, X);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_triangulate2(sosvo_ctx* ctx, const double* b1, const double* b2, int32_t n, const double* t12_host,
                           const double* R12_host, double* X) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, b1 && b2 && t12_host && R12_host && X && n >= 0, "bad arguments");
  if (n == 0) return SOSVO_OK;
  Rt12 rt;
  for (int k = 0; k < 3; ++k) rt.t[k] = t12_host[k];
  for (int k = 0; k < 9; ++k) rt.R[k] = R12_host[k];
  SOSVO_LAUNCH(ctx, triangulate2_kernel, dim3(cdiv(n, kThreads)), dim3(kThreads), 0, ctx->stream, b1, b2, n, rt, X);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_range_filter(sosvo_ctx* ctx, const double* X, int32_t n, double min_range, double max_range,
                           uint8_t* ok) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, X && ok && n >= 0, "bad arguments");
  if (n == 0) return SOSVO_OK;
  SOSVO_LAUNCH(ctx,range_filter_kernel, dim3(cdiv(n, kThreads)), dim3(kThreads), 0, ctx->stream, X, n, min_range,
                     max_range, ok);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_rgbd_backproject(sosvo_ctx* ctx, const float* depth, int32_t rows, int32_t cols, const int32_t* u,
                               const int32_t* v, int32_t n, double fx, double fy, double cx, double cy,
                               double focal_length_m, int32_t depth_is_Z, double* xyz, double* bearing) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, depth && u && v && xyz && bearing && n >= 0 && rows > 0 && cols > 0, "bad arguments");
  if (n == 0) return SOSVO_OK;
  Intr k{fx, fy, cx, cy, focal_length_m};
  SOSVO_LAUNCH(ctx,rgbd_backproject_kernel, dim3(cdiv(n, kThreads)), dim3(kThreads), 0, ctx->stream, depth, rows, cols,
                     u, v, n, k, depth_is_Z, xyz, bearing);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_stereo_assemble(sosvo_ctx* ctx, const sosvo_rig* rig_host, const float* kp_top, const float* kp_bot,
                              const uint8_t* desc_top, const uint8_t* desc_bot, const int32_t* n_top,
                              const int32_t* n_bot, const uint32_t* keys, const int32_t* order, int32_t nframes,
                              int32_t nmask, int32_t cap, int32_t out_cap, float* m_top, float* m_bot, uint8_t* d_top,
                              uint8_t* d_bot, double* X, double* b_top, double* b_bot, int32_t* M, int32_t* n_cand) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, rig_host && kp_top && kp_bot && desc_top && desc_bot && n_top && n_bot && keys && order,
                "null input pointer");
  SOSVO_REQUIRE(ctx, m_top && m_bot && d_top && d_bot && X && b_top && b_bot && M, "null output pointer");
  SOSVO_REQUIRE(ctx, nframes >= 0 && nframes <= (1 << 20) && nmask > 0 && cap > 0 && out_cap > 0, "sizes out of range");
  SOSVO_REQUIRE(ctx, cap <= (1 << SOSVO_KEY_SHIFT), "cap out of range");
  if (nframes == 0) return SOSVO_OK;
  SOSVO_LAUNCH(ctx,stereo_assemble_kernel, dim3(nframes), dim3(kThreads), 0, ctx->stream, *rig_host, kp_top, kp_bot,
                     reinterpret_cast<const uint4*>(desc_top), reinterpret_cast<const uint4*>(desc_bot), n_top, n_bot,
                     keys, order, nmask, cap, out_cap, m_top, m_bot, reinterpret_cast<uint4*>(d_top),
                     reinterpret_cast<uint4*>(d_bot), X, b_top, b_bot, M, n_cand);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_f2f_assemble(sosvo_ctx* ctx, const sosvo_rig* rig_host, const float* m_top, const float* m_bot,
                           const double* X, const double* b_top, const double* b_bot, const int32_t* M,
                           int32_t frame_cap, const int32_t* ref_frame, const int32_t* cur_frame,
                           const uint32_t* keys_top, const int32_t* order_top, const uint32_t* keys_bot,
                           const int32_t* order_bot, int32_t npairs, int32_t corr_cap, double* f, double* p,
                           int32_t* cam, int32_t* corr_q, int32_t* corr_t, int32_t* n, int32_t* n_topview) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, rig_host && m_top && m_bot && X && b_top && b_bot && M && ref_frame && cur_frame,
                "null input pointer");
  SOSVO_REQUIRE(ctx, keys_top && order_top && keys_bot && order_bot && f && p && cam && corr_q && corr_t && n,
                "null pointer");
  SOSVO_REQUIRE(ctx, npairs >= 0 && npairs <= (1 << 20) && frame_cap > 0 && corr_cap > 0, "sizes out of range");
  if (npairs == 0) return SOSVO_OK;
  SOSVO_LAUNCH(ctx,f2f_assemble_kernel, dim3(npairs), dim3(kThreads), 0, ctx->stream, *rig_host, m_top, m_bot, X,
                     b_top, b_bot, M, frame_cap, ref_frame, cur_frame, keys_top, order_top, keys_bot, order_bot,
                     corr_cap, f, p, cam, corr_q, corr_t, n, n_topview);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

}  // extern "C"
