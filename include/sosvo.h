/*
 * sosvo.h -- C ABI of libsosvo.so, the MI355X (gfx950) front end of the single-camera
 * SOS visual-odometry hot path.
 *
 * The reference (ubuntuslave/vo_single_camera_sos) is pure Python and has no FFI of its
 * own: its "native boundary" is the set of Python call sites into cv2 / pyopengv.  Every
 * entry point below names the reference call site (file:line under the reference tree)
 * whose arithmetic it replaces.  The ctypes binding a maintainer would add on the
 * reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns SOSVO_OK (0) or a negative sosvo_status; the message of the
 *     last failure on a context is sosvo_last_error(ctx);
 *   - all data pointers are DEVICE pointers owned by the caller (e.g. torch tensors'
 *     data_ptr()); the library never allocates result memory and keeps no pointer
 *     after a call returns, except the scratch workspace owned by the context;
 *   - every call enqueues work on the context's HIP stream and returns without
 *     synchronising (graph-capturable); counts live in device memory so no stage needs
 *     a host round trip;
 *   - one context per host thread; a context is bound to (device, stream);
 *   - "batched, fixed capacity" layout: problem p of a batch owns rows
 *     [p*stride, p*stride + count[p]) of an array, count[p] <= stride.
 *   - a descriptor is 32 bytes (256 bits); a match key is the u32
 *     (hamming_distance << 20) | train_index, so that unsigned order on keys is the
 *     reference's order: smaller distance first, then the FIRST (lowest) train index.
 */
#ifndef SOSVO_H
#define SOSVO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sosvo_ctx sosvo_ctx;

typedef enum {
  SOSVO_OK = 0,
  SOSVO_ERR_ARG = -1,      /* bad argument (null pointer, size out of range) */
  SOSVO_ERR_HIP = -2,      /* a HIP runtime call failed */
  SOSVO_ERR_NODEVICE = -3, /* no gfx950 device / device index out of range */
  SOSVO_ERR_CAPACITY = -4  /* a fixed capacity would be exceeded */
} sosvo_status;

#define SOSVO_DESC_BYTES 32
#define SOSVO_KEY_SHIFT 20
#define SOSVO_KEY_IDX_MASK 0xFFFFFu
#define SOSVO_KEY_NONE 0xFFFFFFFFu

/* ---- context ------------------------------------------------------------------------ */

/* Version of this ABI (bumped on any signature change). */
int32_t sosvo_abi_version(void);

/* Create a context on `device`; `stream` is a hipStream_t (0 = the null stream; pass
 * torch.cuda.current_stream().cuda_stream to order with torch work). */
int32_t sosvo_create(sosvo_ctx** out, int32_t device, void* stream);
int32_t sosvo_destroy(sosvo_ctx* ctx);
int32_t sosvo_set_stream(sosvo_ctx* ctx, void* stream);
int32_t sosvo_synchronize(sosvo_ctx* ctx);
const char* sosvo_last_error(const sosvo_ctx* ctx);

/* HIP-event timer on the context's stream (used by bench.py for per-kernel durations):
 * start/stop record events; elapsed synchronises on the stop event and returns ms. */
int32_t sosvo_timer_start(sosvo_ctx* ctx);
int32_t sosvo_timer_stop(sosvo_ctx* ctx);
int32_t sosvo_timer_elapsed_ms(sosvo_ctx* ctx, float* ms);

/* ---- K7: brute-force Hamming matching ------------------------------------------------
 * Replaces cv2.BFMatcher(NORM_HAMMING).match / .knnMatch as called from
 * omnistereo/camera_models.py:442 and :420 (FeatureMatcher.match, :404-446).
 *
 * For each problem p < nprob and each query row i < nq[p]:
 *   keys[(p*q_stride + i)*k + 0] = min over j < nt[p] of (hamming(q_i, t_j) << 20 | j)
 *   keys[(p*q_stride + i)*k + 1] = second smallest such key            (only if k == 2)
 * Missing neighbours (nt[p] < k) are SOSVO_KEY_NONE.  Rows i >= nq[p] are not written.
 * q_desc: [nprob*q_stride, 32] u8, t_desc: [nprob*t_stride, 32] u8; nq, nt: [nprob] i32.
 * k is 1 or 2.  q_stride, t_stride <= 2^20.                                             */
int32_t sosvo_match_hamming(sosvo_ctx* ctx, const uint8_t* q_desc, const uint8_t* t_desc,
                            const int32_t* nq, const int32_t* nt, int32_t nprob,
                            int32_t q_stride, int32_t t_stride, int32_t k, uint32_t* keys);

/* Stable sort of each problem's 1-NN matches by distance, the `sorted(matches,
 * key=distance)` of omnistereo/camera_models.py:444.  Input keys as written by
 * sosvo_match_hamming with k = 1.  Output, for rank r < nq[p]:
 *   order[p*q_stride + r] = query index of the r-th match (ties keep query order).
 * Queries whose key is SOSVO_KEY_NONE (empty train set) sort last.                       */
int32_t sosvo_sort_matches(sosvo_ctx* ctx, const uint32_t* keys, const int32_t* nq,
                           int32_t nprob, int32_t q_stride, int32_t* order);

/* ---- K8 / K10: 3D-2D absolute-pose RANSAC ------------------------------------------------
 * Replaces pyopengv.absolute_pose_noncentral_ransac (omnistereo/pose_est_tools.py:785) and
 * pyopengv.absolute_pose_ransac (:915).  Batched: problem b owns rows [b*stride, b*stride+n[b]).
 *   f   [nprob*stride, 3] f64  unit bearings of the CURRENT frame, in their camera's frame
 *   p   [nprob*stride, 3] f64  3-D points of the REFERENCE frame (frame [C], model units)
 *   cam [nprob*stride]    i32  camera index per correspondence, or NULL for the central case
 *   cam_off [ncam,3], cam_rot [ncam,3,3] f64 row-major: camera position / rotation wrt the
 *       viewpoint (TrackerStereoSE3.bootstrap_tracker, pose_est_tools.py:852-859); shared by the batch
 *   flags: SOSVO_FLAG_CAM_ROT_IDENTITY promises cam_rot are identities (skips a mat-vec; results
 *       are bit-identical for finite data)
 *   thr: inlier iff 1 - f . f_hat < thr (pose_est_tools.py:675-676: 1 - cos 5 deg)
 *   max_iter hypotheses are drawn by a counter-based sampler from (seed + b, iteration); with
 *   adaptive != 0 the sequential early stop of OpenGV's sac::Ransac (probability 0.99, sample
 *   size 4) is replayed over them, so the result equals a sequential run on the same samples.
 * Outputs per problem: T_out [nprob,3,4] = [R|t] (pose of the current viewpoint in the reference
 * frame: X_ref = R x + t), inlier_mask [nprob*stride] u8, inlier_idx [nprob*stride] i32 ascending
 * (first n_inliers[b] entries), n_inliers [nprob], info [nprob,4] = {best iteration (-1: none),
 * iterations drawn, status (0 ok / 1 no model -> T = identity), number of valid hypotheses}.
 * hyp_counts: optional [nprob,max_iter] i32, receives each hypothesis' inlier count (-1: the
 * minimal solve failed); NULL to keep them in the context workspace.                         */
#define SOSVO_FLAG_CAM_ROT_IDENTITY 1
int32_t sosvo_ransac_abs_pose(sosvo_ctx* ctx, const double* f, const double* p, const int32_t* cam,
                              const double* cam_off, const double* cam_rot, int32_t ncam, int32_t flags,
                              const int32_t* n, int32_t nprob, int32_t stride, double thr,
                              int32_t max_iter, int32_t adaptive, uint64_t seed, double* T_out,
                              uint8_t* inlier_mask, int32_t* inlier_idx, int32_t* n_inliers,
                              int32_t* info, int32_t* hyp_counts);

/* ---- K9: non-linear refinement --------------------------------------------------------------
 * Replaces pyopengv.absolute_pose_noncentral_optimize_nonlinear (pose_est_tools.py:830) and
 * absolute_pose_optimize_nonlinear (:937): Levenberg-Marquardt on (t, Cayley(R)), residual
 * 1 - f . f_hat per correspondence, forward-difference Jacobian.  Same layout as the RANSAC call;
 * idx/m (both or neither): the first m[b] entries of idx[b*stride ...] select the correspondences
 * (e.g. inlier_idx / n_inliers of sosvo_ransac_abs_pose); NULL = all n[b].
 * T_io [nprob,3,4] in: start pose, out: refined pose.  cost_out [nprob] f64 and iters_out
 * [nprob] i32 are optional.                                                                      */
int32_t sosvo_refine_abs_pose(sosvo_ctx* ctx, const double* f, const double* p, const int32_t* cam,
                              const double* cam_off, const double* cam_rot, int32_t ncam,
                              const int32_t* n, int32_t nprob, int32_t stride, const int32_t* idx,
                              const int32_t* m, int32_t max_lm_iter, double* T_io, double* cost_out,
                              int32_t* iters_out);

#ifdef __cplusplus
}
#endif
#endif /* SOSVO_H */
