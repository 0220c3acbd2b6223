// The RGB-D (perspective camera) variant of the per-frame / per-pair glue, batched:
//
//   rgbd_assemble      RGBDFrame.establish_keypoints (omnistereo/pose_est_tools.py:600-623) after the detector:
//                      RGBDCamModel.get_XYZ (camera_models.py:835-860, get_depth_Z :781-799) at the keypoints'
//                      integer pixels, NaN test (:613), range test on Z (:615, filter_3D_points_due_to_range
//                      :570-592), get_normalized_points (camera_models.py:203-212), stable compaction (:620-622)
//   f2f_assemble_central
//                      TrackerRGBDSE3.track_frame steps 1-2 (pose_est_tools.py:896-913): the sorted frame-to-frame
//                      matches (query = current, train = reference) cut at percentage_good_matches (:225), gated
//                      on |du| (:245-247), bearings of the current frame / 3-D points of the reference frame
//
// One workgroup per frame (pair); FP64 with the oracle's operation order (+ - * / sqrt only), stable order by a
// ballot prefix + running base.
#include "common.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void rgbd_assemble_kernel(sosvo_rgbd_cam cam, const float* __restrict__ kp,
                                                                 const uint4* __restrict__ desc,
                                                                 const int32_t* __restrict__ n_kp,
                                                                 const float* __restrict__ depth, int rows, int cols,
                                                                 int cap, int out_cap, float* __restrict__ m,
                                                                 uint4* __restrict__ d, double* __restrict__ X,
                                                                 double* __restrict__ b, int32_t* __restrict__ M) {
  __shared__ int wave_off[kThreads / 64 + 1];
  __shared__ int s_running;
  const int tid = threadIdx.x, fr = blockIdx.x;
  if (tid == 0) s_running = 0;
  __syncthreads();
  const int n = min(n_kp[fr], cap);
  const float* dep = depth + (size_t)fr * rows * cols;
  const double nan = __longlong_as_double(0x7FF8000000000000LL);
  for (int i0 = 0; i0 < n; i0 += kThreads) {
    const int i = i0 + tid;
    bool valid = false;
    float x = 0.f, y = 0.f;
    double P[3] = {0, 0, 0};
    if (i < n) {
      x = kp[2 * ((size_t)fr * cap + i)];
      y = kp[2 * ((size_t)fr * cap + i) + 1];
      const int ui = min(max((int)x, 0), cols - 1), vi = min(max((int)y, 0), rows - 1);  // .astype(np.uint) (:611)
      const float dv = dep[(size_t)vi * cols + ui];
      double dd = (double)dv;
      if (!cam.depth_is_Z) {  // radial depth -> Z (camera_models.py:781-799; focal_length * depth stays float32)
        const double xi = (cam.focal_length_m / cam.fx) * ((double)ui - cam.cx);
        const double yi = (cam.focal_length_m / cam.fy) * ((double)vi - cam.cy), zi = cam.focal_length_m;
        const float fd = (float)cam.focal_length_m * dv;
        dd = (double)fd / sqrt(xi * xi + yi * yi + zi * zi);
      }
      const double Z = (dd != 0.0) ? dd : nan;  // :846
      P[0] = ((double)ui - cam.cx) * Z / cam.fx;
      P[1] = ((double)vi - cam.cy) * Z / cam.fy;
      P[2] = Z;
      const double az = fabs(Z);                // norm over the single Z row (:583), NaN -> 0 (:584)
      valid = Z == Z;
      if (cam.min_range > 0) valid = valid && az >= cam.min_range;
      if (cam.max_range > 0) valid = valid && az <= cam.max_range;
    }
    const int pos = sosvo_block_compact_pos(valid, wave_off, &s_running, tid);
    if (valid && pos < out_cap) {
      const size_t o = (size_t)fr * out_cap + pos;
      m[2 * o] = x;
      m[2 * o + 1] = y;
      d[2 * o] = desc[2 * ((size_t)fr * cap + i)];
      d[2 * o + 1] = desc[2 * ((size_t)fr * cap + i) + 1];
      const double nrm = sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        X[3 * o + k] = P[k];
        b[3 * o + k] = P[k] / nrm;
      }
    }
  }
  __syncthreads();
  if (tid == 0) M[fr] = min(s_running, out_cap);
}

__global__ __launch_bounds__(kThreads) void f2f_assemble_central_kernel(
    double pct_good_matches, double max_hdiff, const float* __restrict__ m, const double* __restrict__ X,
    const double* __restrict__ b, const int32_t* __restrict__ M, int frame_cap, const int32_t* __restrict__ ref_frame,
    const int32_t* __restrict__ cur_frame, const uint32_t* __restrict__ keys, const int32_t* __restrict__ order,
    int corr_cap, double* __restrict__ f, double* __restrict__ p, int32_t* __restrict__ corr_q,
    int32_t* __restrict__ corr_t, int32_t* __restrict__ n) {
  __shared__ int wave_off[kThreads / 64 + 1];
  __shared__ int s_running;
  const int tid = threadIdx.x, pr = blockIdx.x;
  if (tid == 0) s_running = 0;
  __syncthreads();
  const int fr_ref = ref_frame[pr], fr_cur = cur_frame[pr];
  const int nq = min(M[fr_cur], frame_cap), nt = min(M[fr_ref], frame_cap);
  const int good = (nt > 0) ? (int)(pct_good_matches * (double)nq) : 0;  // pose_est_tools.py:225
  const size_t kb = (size_t)pr * frame_cap;
  for (int r0 = 0; r0 < good; r0 += kThreads) {
    const int r = r0 + tid;
    bool valid = false;
    int q = 0, t = 0;
    if (r < good) {
      q = order[kb + r];                              // query = current frame (:215)
      t = (int)(keys[kb + q] & SOSVO_KEY_IDX_MASK);   // train = reference (key)frame
      valid = true;
      if (max_hdiff > 0) {                            // :245-247 -> common_cv.py:177 (only |du| is tested)
        const double u_train = (double)m[2 * ((size_t)fr_ref * frame_cap + t)];
        const double u_query = (double)m[2 * ((size_t)fr_cur * frame_cap + q)];
        valid = fabs(u_train - u_query) <= max_hdiff;
      }
    }
    const int pos = sosvo_block_compact_pos(valid, wave_off, &s_running, tid);
    if (valid && pos < corr_cap) {
      const size_t o = (size_t)pr * corr_cap + pos;
      const size_t qrow = (size_t)fr_cur * frame_cap + q, trow = (size_t)fr_ref * frame_cap + t;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        f[3 * o + k] = b[3 * qrow + k];               // bearings of the CURRENT frame (:909)
        p[3 * o + k] = X[3 * trow + k];               // 3-D points of the REFERENCE frame (:910)
      }
      corr_q[o] = q;
      corr_t[o] = t;
    }
  }
  __syncthreads();
  if (tid == 0) n[pr] = min(s_running, corr_cap);
}

}  // namespace

extern "C" {

int32_t sosvo_rgbd_assemble(sosvo_ctx* ctx, const sosvo_rgbd_cam* cam_host, const float* kp, const uint8_t* desc,
                            const int32_t* n, const float* depth, int32_t nframes, int32_t rows, int32_t cols,
                            int32_t cap, int32_t out_cap, float* m, uint8_t* d, double* X, double* b, int32_t* M) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, cam_host && kp && desc && n && depth && m && d && X && b && M, "null pointer");
  SOSVO_REQUIRE(ctx, nframes >= 0 && nframes <= (1 << 20) && rows > 0 && cols > 0 && cap > 0 && out_cap > 0,
                "sizes out of range");
  SOSVO_REQUIRE(ctx, cam_host->fx != 0 && cam_host->fy != 0, "focal lengths must be non-zero");
  SOSVO_REQUIRE(ctx, (((uintptr_t)desc | (uintptr_t)d) & 15) == 0, "descriptor buffers must be 16-byte aligned");
  if (nframes == 0) return SOSVO_OK;
  SOSVO_LAUNCH(ctx, rgbd_assemble_kernel, dim3(nframes), dim3(kThreads), 0, ctx->stream, *cam_host, kp,
               reinterpret_cast<const uint4*>(desc), n, depth, rows, cols, cap, out_cap, m, reinterpret_cast<uint4*>(d), X, b,
               M);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_f2f_assemble_central(sosvo_ctx* ctx, double pct_good_matches, double max_hdiff, const float* m,
                                   const double* X, const double* b, const int32_t* M, int32_t frame_cap,
                                   const int32_t* ref_frame, const int32_t* cur_frame, const uint32_t* keys,
                                   const int32_t* order, int32_t npairs, int32_t corr_cap, double* f, double* p,
                                   int32_t* corr_q, int32_t* corr_t, int32_t* n) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, m && X && b && M && ref_frame && cur_frame && keys && order, "null input pointer");
  SOSVO_REQUIRE(ctx, f && p && corr_q && corr_t && n, "null output pointer");
  SOSVO_REQUIRE(ctx, npairs >= 0 && npairs <= (1 << 20) && frame_cap > 0 && corr_cap > 0, "sizes out of range");
  if (npairs == 0) return SOSVO_OK;
  SOSVO_LAUNCH(ctx, f2f_assemble_central_kernel, dim3(npairs), dim3(kThreads), 0, ctx->stream, pct_good_matches, max_hdiff,
               m, X, b, M, frame_cap, ref_frame, cur_frame, keys, order, corr_cap, f, p, corr_q, corr_t, n);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

}  // extern "C"
