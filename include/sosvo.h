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

/* The descriptor test pattern cv2.ORB_create(...).compute uses (omnistereo/camera_models.py:1682, :1765;
 * pose_est_tools.py:520, :553): OpenCV's learned table bit_pattern_31_, 256 tests x (x0, y0, x1, y1) = [512, 2] int8
 * points, |coordinate| <= 13, written to HOST memory (1024 bytes).  It is the `pattern` argument of sosvo_describe_orb*,
 * sosvo_detect_orb* and the batch entry points (which take it in device memory: copy it once).  Any other [512, 2] table
 * with |coordinate| <= 16 is accepted there too (the Python package keeps a seeded one as `orb_pattern.seeded_pattern()`).
 * No context needed. */
int32_t sosvo_orb_bit_pattern_31(int8_t* pattern_host);

/* Create a context on `device`; `stream` is a hipStream_t (0 = the null stream; pass
 * torch.cuda.current_stream().cuda_stream to order with torch work). */
int32_t sosvo_create(sosvo_ctx** out, int32_t device, void* stream);
int32_t sosvo_destroy(sosvo_ctx* ctx);
int32_t sosvo_set_stream(sosvo_ctx* ctx, void* stream);
/* Hints that change scheduling, never results.  SOSVO_HINT_SHARED_DEVICE (value != 0): kernels of OTHER contexts / streams
 * run on the device at the same time (a batch split over streams): the register-filling median kernel then keeps to three
 * workgroups per CU so that the others find wave slots (+1.3 % on the three-stream step, -2.5 % for that kernel alone).
 * sosvo_frame_pair_batch_streams sets it on its internal contexts itself. */
#define SOSVO_HINT_SHARED_DEVICE 1
/* SOSVO_HINT_SCORE_FP64_ONLY (value != 0): the RANSAC scoring kernel skips its single-precision first tier (which decides
 * an inlier test only where a proven rounding bound allows and leaves the rest to double precision) and evaluates every
 * test in double precision, as rounds 1 - 3 did.  Same counts, bit for bit (tests/test_gpu_ransac.py compares the two
 * forms); ~1.6 x the kernel's time.  For A/B measurements (scripts/score_tiers.py). */
#define SOSVO_HINT_SCORE_FP64_ONLY 2
int32_t sosvo_set_hint(sosvo_ctx* ctx, int32_t hint, int32_t value);
int32_t sosvo_synchronize(sosvo_ctx* ctx);
const char* sosvo_last_error(const sosvo_ctx* ctx);

/* HIP-event timer on the context's stream (used by bench.py for per-kernel durations):
 * start/stop record events; elapsed synchronises on the stop event and returns ms. */
int32_t sosvo_timer_start(sosvo_ctx* ctx);
int32_t sosvo_timer_stop(sosvo_ctx* ctx);
int32_t sosvo_timer_elapsed_ms(sosvo_ctx* ctx, float* ms);

/* Per-kernel HIP-event profile on the context's stream: while enabled, every kernel launch of
 * every entry point is bracketed by an event pair (up to 16384 launches since the last enable).
 * sosvo_profile_enable(on) also clears the record.  sosvo_profile_get synchronises on entry i and
 * returns the kernel's label and its duration in ms.                                            */
int32_t sosvo_profile_enable(sosvo_ctx* ctx, int32_t on);
int32_t sosvo_profile_count(sosvo_ctx* ctx);
int32_t sosvo_profile_get(sosvo_ctx* ctx, int32_t i, char* name_out, int32_t name_cap, float* ms);

/* Test hook: fills the context's internal scratch memory (response maps, sort keys, RANSAC hypotheses, blurred images;
 * the internal sub-contexts of sosvo_frame_pair_batch_streams included) with `byte`, on the context's stream(s).  No
 * entry point may read scratch it has not written in the same call, so results must not depend on this.            */
int32_t sosvo_debug_fill_scratch(sosvo_ctx* ctx, int32_t byte);

/* ---- K1: unwrap (a1 + a2) ---------------------------------------------------------------------
 * Replaces Panorama.get_panoramic_image's cv2.remap(omni, map_x, map_y, INTER_LINEAR,
 * BORDER_CONSTANT, 0) (omnistereo/panorama.py:293) for both mirrors of every frame, with
 * OmniStereoModel.get_fully_masked_images' bitwise_and (camera_models.py:2991-2996) folded into
 * the bilinear taps (a masked-out source pixel reads 0, exactly as masking the image first).
 *   omni  [nframes, H, W, 3] u8 (BGR)
 *   masks [2, H, W] u8, non-zero = pixel belongs to that mirror's annulus; NULL = no masking
 *   map_x, map_y [2, rows, cols] f32: Panorama.world2cam_LUT_map_x/y cast to float32 (:291-292);
 *       NaN entries give the border colour 0
 *   pano  [2, nframes, rows, cols, 3] u8 -- VIEW-MAJOR: image i = view * nframes + frame
 * Arithmetic: coordinates rounded to 1/32 px (ties to even), weights (32-fx)(32-fy),
 * (sum + 512) >> 10; taps outside the image read 0.                                            */
int32_t sosvo_unwrap(sosvo_ctx* ctx, const uint8_t* omni, const uint8_t* masks, const float* map_x,
                     const float* map_y, int32_t nframes, int32_t H, int32_t W, int32_t rows,
                     int32_t cols, uint8_t* pano);

/* Table-driven form of K1 for the batched path (identical results): sosvo_unwrap_prepare folds the
 * float maps, the 1/32-px rounding, the border test and the annulus masks into a packed table
 * [2, rows, cols, 2] u32 once per model (an opaque 8-byte entry per panorama pixel: the byte offset of the first
 * tap and the four blend weights; the table is specific to (H, W), and frames larger than 4 MB take a slower,
 * still exact entry format); sosvo_unwrap_table then unwraps nframes frames with one 8-byte table load and two
 * unaligned 8-byte tap loads per panorama pixel.  H * W * 3 >= 2 * 3 * W + 8.                              */
int32_t sosvo_unwrap_prepare(sosvo_ctx* ctx, const uint8_t* masks, const float* map_x, const float* map_y,
                             int32_t H, int32_t W, int32_t rows, int32_t cols, uint32_t* table);
int32_t sosvo_unwrap_table(sosvo_ctx* ctx, const uint8_t* omni, const uint32_t* table, int32_t nframes,
                           int32_t H, int32_t W, int32_t rows, int32_t cols, uint8_t* pano);

/* ---- K2 + K3: median blur + gray --------------------------------------------------------------
 * Replaces cv2.medianBlur(pano, ksize) (omnistereo/camera_models.py:1711, pose_est_tools.py:528;
 * ksize 11 for the SOS frames, 0 = none for RGB-D) followed by cv2.cvtColor(BGR2GRAY)
 * (camera_models.py:1714, or the conversion ORB does internally).  Exact k x k median per channel
 * with replicated border, then (1868 B + 9617 G + 4899 R + 8192) >> 14.
 *   img [nimg, rows, cols, 3] u8 -> gray [nimg, rows, cols] u8.  ksize in {0, 1, 3, 5, 11}.       */
int32_t sosvo_median_gray(sosvo_ctx* ctx, const uint8_t* img, int32_t nimg, int32_t rows,
                          int32_t cols, int32_t ksize, uint8_t* gray);
/* K1 + K2 + K3 in one kernel for the batched path (identical results to sosvo_unwrap_table followed by
 * sosvo_median_gray on its output): every lane unwraps its source pixel from the table on the fly, so the
 * colour panoramas (camera_models.py:2991-2996 -> :1711 -> :1714) are never written to HBM.
 *   omni [nframes, H, W, 3] u8, table from sosvo_unwrap_prepare -> gray [2 * nframes, rows, cols] u8
 *   (view-major).  ksize in {3, 5, 11}; 0 / 1 = no median: the pixel is unwrapped and converted to gray at once.   */
int32_t sosvo_unwrap_median_gray(sosvo_ctx* ctx, const uint8_t* omni, const uint32_t* table, int32_t nframes,
                                 int32_t H, int32_t W, int32_t rows, int32_t cols, int32_t ksize, uint8_t* gray);

/* The same with the output restricted to the rows a consumer can reach: row_range = int32 [2][2] in DEVICE memory
 * (view-major: first row, last row + 1 of the top and of the bottom panoramas), written by sosvo_gray_rows_needed;
 * NULL = all rows (sosvo_unwrap_median_gray).  Rows outside a view's range are neither computed nor written (the
 * caller's buffer keeps what it held); inside, the result is sosvo_unwrap_median_gray's bit for bit.  (ksize 0 / 1 writes
 * every row and does not read row_range.) */
int32_t sosvo_unwrap_median_gray_rows(sosvo_ctx* ctx, const uint8_t* omni, const uint32_t* table, int32_t nframes,
                                      int32_t H, int32_t W, int32_t rows, int32_t cols, int32_t ksize,
                                      const int32_t* row_range, uint8_t* gray);

/* Rows of the gray panoramas that goodFeaturesToTrack on the azimuthal masks (camera_models.py:1739: the response at
 * the masked pixels and its 3x3 dilation, i.e. gray within 3 rows of a masked row) and ORB.compute on its keypoints
 * (camera_models.py:1765: keypoints edge <= y < rows - edge inside the masks, rotated pattern on the 7x7-blurred
 * image) can reach, per mask set: mask_bits [nsets, rows, cols] u32 -> row_range [nsets][2] int32 (device): first row,
 * last row + 1 (0, 0 for an empty set).  Model constant: compute once per (masks, edge, pattern, angle). */
int32_t sosvo_gray_rows_needed(sosvo_ctx* ctx, const uint32_t* mask_bits, int32_t nsets, int32_t rows, int32_t cols,
                               int32_t nmask, int32_t edge, const int8_t* pattern, float cos_a, float sin_a,
                               int32_t* row_range);

/* ---- K4: goodFeaturesToTrack per azimuthal mask (a4, the reference's default detector) ---------
 * Replaces cv2.goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, mask, useHarris=False)
 * as called per mask from OmniCamModel.detect_sparse_features_on_panorama
 * (omnistereo/camera_models.py:1739, :1778) and RGBDFrame.detect_sparse_features (pose_est_tools.py:544).
 *   gray      [nimg, rows, cols] u8 (the median-blurred gray panorama, camera_models.py:1711-1714)
 *   mask_bits [nsets, rows, cols] u32: bit m set = the pixel belongs to azimuthal mask m (masks may
 *             overlap); image i uses set i / images_per_maskset (one set per mirror)
 * Problem p = image * nmask + mask.  Per problem: min-eigenvalue map (Sobel 3, block 3, reflect-101,
 * float32), maxVal over the mask, threshold quality * maxVal, 3x3 dilation NMS excluding the 1-px
 * image border, descending sort (ties: higher address first), greedy min-distance on a cell grid,
 * stop at max_corners (<= 0: no limit) or at the output capacity `cap`.  quality > 0; quality >= 1 is accepted as by
 * OpenCV and yields no corners (threshold >= maximum).  Only positive responses can become corners (with 0 < quality
 * the threshold is positive wherever the mask's maximum is; a mask whose maximum is <= 0 has no corners).
 *   kp [nimg*nmask, cap, 2] f32 (x, y) in acceptance order, n [nimg*nmask] i32,
 *   status [nimg*nmask] i32 (optional): bit 0 = more candidates than the selection keeps (4096; 16384 when
 *   cap > 1024: the large-mask variant for whole-image detection, e.g. the RGB-D frames) -- the result then
 *   depends on which were kept; bit 1 = mask too large for the on-chip grid (3584 / 15872 cells of
 *   minDistance^2 pixels) and more than 1024 corners wanted, or more than 1024 corners were the THIRD of their grid
 *   cell (possible from minDistance ~30 px on; up to 1024 of them are kept exactly in an overflow list). */
int32_t sosvo_detect_gft(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_bits, int32_t nimg,
                         int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask,
                         double quality, double min_distance, int32_t max_corners, int32_t cap,
                         float* kp, int32_t* n, int32_t* status);

/* ---- K6: ORB descriptors on provided keypoints (a4) --------------------------------------------
 * Replaces ORB_create(nfeatures).compute(image, keypoints) (omnistereo/camera_models.py:1765, :1785;
 * pose_est_tools.py:553): keypoints nearer than `edge` (31) px to the border are REMOVED (kp and n
 * are compacted in place, order kept), the image is blurred 7x7 sigma 2 (8.8 fixed point), and the
 * 256 pair tests of `pattern` [512,2] i8 (device) are evaluated at offsets rotated by the keypoint
 * angle (cos_a, sin_a as float32: the GFT path gives every keypoint angle -1 degree) and rounded.
 * Problem p = image * nmask + mask as above.  desc [nimg*nmask, cap, 32] u8, 8 tests per byte, LSB first. */
int32_t sosvo_describe_orb(sosvo_ctx* ctx, const uint8_t* gray, int32_t nimg, int32_t rows, int32_t cols,
                           int32_t nmask, int32_t cap, float* kp, int32_t* n, float cos_a, float sin_a,
                           const int8_t* pattern, int32_t edge, uint8_t* desc);

/* The same with the blur restricted to the rows a descriptor can read: row_range = int32 [2][2] in DEVICE memory as written
 * by sosvo_gray_rows_needed for the same masks / edge / pattern (view-major: first row, last row + 1; image i belongs to
 * view i / (nimg / 2)); NULL = all rows (sosvo_describe_orb).  Descriptors are sosvo_describe_orb's bit for bit.
 * Precondition made safe: a keypoint whose patch rows y - R .. y + R (R = the rotated pattern's radius) leave the rows
 * [first + 3, last + 1 - 3) of its view's range (inner ends only; an end at the image border does not count) would read
 * rows that were never blurred -- it is REMOVED like a keypoint within `edge` of the border.  Keypoints inside the masks
 * the range was derived from are never affected. */
int32_t sosvo_describe_orb_rows(sosvo_ctx* ctx, const uint8_t* gray, int32_t nimg, int32_t rows, int32_t cols,
                                int32_t nmask, int32_t cap, float* kp, int32_t* n, float cos_a, float sin_a,
                                const int8_t* pattern, int32_t edge, const int32_t* row_range, uint8_t* desc);

/* ---- K5: ORB keypoint detection per azimuthal mask (a4, feature_detection_method "ORB") ----------
 * Replaces cv2.ORB_create(nfeatures=N).detect(image, mask) (omnistereo/camera_models.py:1640, :1755;
 * pose_est_tools.py:478, :547) for all azimuthal masks of all images at once: 8 levels x 1.2, per-level
 * quotas, FAST-9/16 (threshold 20) with 3x3 NMS, 31-px border, (resized) mask, best 2 n_l by FAST score,
 * Harris response, best n_l, intensity-centroid orientation; see vo_single_camera_sos_amd/csrc/orb.hip.
 *   sosvo_orb_pyramid_pixels(rows, cols): pixels of one image's 8-level pyramid (host helper)
 *   sosvo_orb_mask_pyramid: mask_bits [nsets, rows, cols] u32 -> mask_pyr [nsets, pyramid_pixels] u32,
 *       every mask resized level by level (kept where > 254); call once per model
 *   sosvo_detect_orb: gray [nimg, rows, cols] u8 -> kp4 [nimg*nmask, cap, 4] f32 = (x, y in level-0
 *       coordinates, angle in degrees, level), resp [nimg*nmask, cap] f32 (Harris), n [nimg*nmask] i32;
 *       keypoints ordered by level, then response descending, then (y, x).  Per (problem, level) any number of FAST
 *       maxima is handled; only a retainBest set (best 2 n_l by FAST score, ties kept) of more than 2048 keypoints --
 *       thousands tied at the threshold score -- is cut to 2048 (which of the tied ones go on is then unspecified).
 * ---- K6': ORB descriptors for oriented multi-level keypoints
 * Replaces .compute(image, keypoints) on ORB's own keypoints (camera_models.py:1765): keypoints within
 * 31 px of the level-0 border are removed (kp4 / n compacted in place), each level is blurred 7x7 sigma 2,
 * the pattern is rotated by the keypoint's angle.  desc [nimg*nmask, cap, 32] u8; kp_xy (optional)
 * [nimg*nmask, cap, 2] f32 receives the compacted (x, y) for the matching stages.
 * sosvo_describe_orb_levels builds its own pyramid (all 8 levels: the keypoints may come from anywhere).
 *   sosvo_detect_describe_orb: detect, then compute, as the reference does (camera_models.py:1755, :1765), in ONE call
 *       that builds ONE pyramid -- only as far as a keypoint can come from (levels with a quota and more than the 31-px
 *       border) -- and shares it between the two halves.  Same outputs as sosvo_detect_orb followed by
 *       sosvo_describe_orb_levels, bit for bit; cap <= 2048.                                                        */
int64_t sosvo_orb_pyramid_pixels(int32_t rows, int32_t cols);
int32_t sosvo_orb_mask_pyramid(sosvo_ctx* ctx, const uint32_t* mask_bits, int32_t nsets, int32_t rows,
                               int32_t cols, int32_t nmask, uint32_t* mask_pyr);
int32_t sosvo_detect_orb(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_pyr, int32_t nimg,
                         int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask,
                         int32_t nfeatures, int32_t cap, float* kp4, float* resp, int32_t* n);
int32_t sosvo_describe_orb_levels(sosvo_ctx* ctx, const uint8_t* gray, int32_t nimg, int32_t rows,
                                  int32_t cols, int32_t nmask, int32_t cap, float* kp4, int32_t* n,
                                  const int8_t* pattern, uint8_t* desc, float* kp_xy);
int32_t sosvo_detect_describe_orb(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_pyr, int32_t nimg,
                                  int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask,
                                  int32_t nfeatures, int32_t cap, float* kp4, float* resp, int32_t* n,
                                  const int8_t* pattern, uint8_t* desc, float* kp_xy);

/* ---- FAST as a detector of its own (a4, feature_detection_method "FAST") -------------------------------
 * Replaces cv2.FastFeatureDetector_create() + setNonmaxSuppression(True) + .detect(image, mask)
 * (omnistereo/camera_models.py:1664-1666, :1755; pose_est_tools.py:506-508) for all azimuthal masks of all
 * images: FAST-9/16 with `threshold` (OpenCV default 10) on the whole image, 3x3 non-maximum suppression on the
 * corner score, keypoints where the mask bit is set, in raster order (rows, then x).  The descriptors then come
 * from sosvo_describe_orb with the fixed angle -1 degree, as for GFT keypoints (camera_models.py:1765).
 *   kp [nimg*nmask, cap, 2] f32 (x, y), n [nimg*nmask] i32, status (optional): 1 = more than cap corners (the
 *   first cap in raster order are kept).                                                                      */
int32_t sosvo_detect_fast(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_bits, int32_t nimg,
                          int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask,
                          int32_t threshold, int32_t cap, float* kp, int32_t* n, int32_t* status);

/* ---- AGAST as a detector (a4, feature_detection_method "AGAST") -------------------------------------------
 * Replaces cv2.AgastFeatureDetector_create() + setNonmaxSuppression(True) + .detect(image, mask)
 * (omnistereo/camera_models.py:1670-1671, :1755; pose_est_tools.py:508-509): OAST 9/16 -- the corner set and response of
 * FAST-9/16 at `threshold` (OpenCV default 10) -- with AGAST's own non-maximum suppression (one maximum per block of
 * vertically / horizontally touching corners), keypoints where the mask bit is set, raster order.  Same arguments and
 * outputs as sosvo_detect_fast; status bit 1: more than 16384 corners in an image (the rest is dropped); rows <= 1024. */
int32_t sosvo_detect_agast(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_bits, int32_t nimg,
                           int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask,
                           int32_t threshold, int32_t cap, float* kp, int32_t* n, int32_t* status);

/* ---- K7: brute-force Hamming matching ------------------------------------------------
 * Replaces cv2.BFMatcher(NORM_HAMMING).match / .knnMatch as called from
 * omnistereo/camera_models.py:442 and :420 (FeatureMatcher.match, :404-446).
 *
 * Problem p < nprob matches the query rows of block qs = q_slot ? q_slot[p] : p against the
 * train rows of block ts = t_slot ? t_slot[p] : p (the slots let many problems share one
 * descriptor store, e.g. frame-to-frame problems index frames).  For each query row i < nq[qs]:
 *   keys[(p*q_stride + i)*k + 0] = min over j < nt[ts] of (hamming(q_i, t_j) << 20 | j)
 *   keys[(p*q_stride + i)*k + 1] = second smallest such key            (only if k == 2)
 * Missing neighbours (nt[ts] < k) are SOSVO_KEY_NONE.  Rows i >= nq[qs] are unspecified.
 * q_desc: [*, q_stride, 32] u8, t_desc: [*, t_stride, 32] u8 (16-byte aligned); nq, nt: i32
 * counts per block; q_slot, t_slot: [nprob] i32 or NULL.  k is 1 or 2.  strides <= 2^20.   */
int32_t sosvo_match_hamming(sosvo_ctx* ctx, const uint8_t* q_desc, const uint8_t* t_desc,
                            const int32_t* nq, const int32_t* nt, const int32_t* q_slot,
                            const int32_t* t_slot, int32_t nprob, int32_t q_stride,
                            int32_t t_stride, int32_t k, uint32_t* keys);

/* Float descriptors (SIFT / SURF / KAZE ...; feature_detection_method "SIFT" / "SURF" makes FeatureMatcher build
 * cv2.BFMatcher() with its default NORM_L2, omnistereo/camera_models.py:394-396): brute-force L2 matching.
 * q_desc [nprob, q_stride, dim] f32, t_desc [nprob, t_stride, dim] f32, dim <= 128, finite values.  distance = sqrt of the
 * float32 sum of squared differences, four terms per step in index order (OpenCV's scalar normL2Sqr_).
 *   keys[(p*q_stride + i)*k + r] = (bits of the float32 distance << 32) | train index of the r-th neighbour (r < k,
 *   k = 1 or 2; smallest distance, first train index on ties), all ones if absent.
 * The reference's FLANN matcher (camera_models.py:384-393: KD-trees for float, LSH for binary descriptors, randomised,
 * approximate) has no counterpart: matcher_type "FLANN" is served by the exact searches (sosvo_match_hamming / this). */
int32_t sosvo_match_l2(sosvo_ctx* ctx, const float* q_desc, const float* t_desc, const int32_t* nq, const int32_t* nt,
                       int32_t nprob, int32_t q_stride, int32_t t_stride, int32_t dim, int32_t k, uint64_t* keys);

/* Radius match: cv2.BFMatcher.radiusMatch(query, train, maxDistance) as called from FeatureMatcher.match when
 * use_radius_match is set (omnistereo/camera_models.py:412-415).  For each query row i < nq[qs] ALL train rows j
 * with hamming(q_i, t_j) <= max_distance, as packed keys in ascending order (distance, then train index):
 *   keys[(p*q_stride + i)*cap + r], r < min(counts, cap); the rest SOSVO_KEY_NONE
 *   counts[p*q_stride + i] = number of train rows within the radius (when it exceeds cap, the cap smallest
 *   keys are the ones kept).  cap <= 512.  Blocks / slots as in sosvo_match_hamming.                      */
int32_t sosvo_match_radius(sosvo_ctx* ctx, const uint8_t* q_desc, const uint8_t* t_desc, const int32_t* nq,
                           const int32_t* nt, const int32_t* q_slot, const int32_t* t_slot, int32_t nprob,
                           int32_t q_stride, int32_t t_stride, int32_t max_distance, int32_t cap,
                           uint32_t* keys, int32_t* counts);

/* Stable sort of each problem's 1-NN matches by distance, the `sorted(matches,
 * key=distance)` of omnistereo/camera_models.py:444.  Input keys as written by
 * sosvo_match_hamming with k = 1.  Output, for rank r < nq[qs]:
 *   order[p*q_stride + r] = query index of the r-th match (ties keep query order).
 * Queries whose key is SOSVO_KEY_NONE (empty train set) sort last.  q_stride <= 16384.    */
int32_t sosvo_sort_matches(sosvo_ctx* ctx, const uint32_t* keys, const int32_t* nq,
                           const int32_t* q_slot, int32_t nprob, int32_t q_stride, int32_t* order);

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
/* central problems (cam == NULL): hypotheses from EPnP on 6-point samples, as OpenGV's absolute_pose_ransac does for
 * algorithm "EPNP" (the RGB-D tracker's choice, pose_est_tools.py:697, :915), instead of Kneip P3P + a 4th point;
 * the adaptive stop then uses w^6. */
#define SOSVO_FLAG_EPNP 2
/* hypotheses from the GENERALISED P3P: four distinct correspondences out of ALL cameras, the generalised three-point
 * problem on the first three (up to 8 poses), the fourth picks one -- what OpenGV's non-central problem does ("will
 * ALWAYS use GP3P", pose_est_tools.py:696, :785).  Without the flag the three solve points of a sample come from ONE
 * camera (central Kneip P3P moved to the body frame; BASELINE config 2's "P3P RANSAC").  Also valid for central
 * problems (then a P3P that only returns configurations in front of the camera). */
#define SOSVO_FLAG_GP3P 4
/* central problems: "TWOPT" (pose_est_tools.py:95-107) -- 2-point samples, the translation of a camera whose rotation is
 * known (the identity, as pyopengv's binding leaves OpenGV's rotation prior); the adaptive stop then uses w^2. */
#define SOSVO_FLAG_TWOPT 8
int32_t sosvo_ransac_abs_pose(sosvo_ctx* ctx, const double* f, const double* p, const int32_t* cam,
                              const double* cam_off, const double* cam_rot, int32_t ncam, int32_t flags,
                              const int32_t* n, int32_t nprob, int32_t stride, double thr,
                              int32_t max_iter, int32_t adaptive, uint64_t seed, double* T_out,
                              uint8_t* inlier_mask, int32_t* inlier_idx, int32_t* n_inliers,
                              int32_t* info, int32_t* hyp_counts);

/* ---- K9: non-linear refinement --------------------------------------------------------------
 * Replaces pyopengv.absolute_pose_noncentral_optimize_nonlinear (pose_est_tools.py:830) and
 * absolute_pose_optimize_nonlinear (:937): Levenberg-Marquardt on (t, Cayley(R)), residual
 * 1 - f . f_hat per correspondence, analytic Jacobian (OpenGV differentiates numerically).  Same layout as the
 * RANSAC call;
 * idx/m (both or neither): the first m[b] entries of idx[b*stride ...] select the correspondences
 * (e.g. inlier_idx / n_inliers of sosvo_ransac_abs_pose); NULL = all n[b].
 * T_io [nprob,3,4] in: start pose, out: refined pose.  cost_out [nprob] f64 and iters_out
 * [nprob] i32 are optional.                                                                      */
int32_t sosvo_refine_abs_pose(sosvo_ctx* ctx, const double* f, const double* p, const int32_t* cam,
                              const double* cam_off, const double* cam_rot, int32_t ncam,
                              const int32_t* n, int32_t nprob, int32_t stride, const int32_t* idx,
                              const int32_t* m, int32_t max_lm_iter, double* T_io, double* cost_out,
                              int32_t* iters_out);

/* ---- 2D-2D relative-pose RANSAC ------------------------------------------------
 * Replaces pyopengv.relative_pose_ransac(b1, b2, algorithm, threshold, max_iterations) as called by
 * pose_relative_ransac_2D_to_2D (omnistereo/pose_est_tools.py:54-90, call at :78).
 * f1, f2 [nprob*stride,3] f64: unit bearings of the SAME features seen from viewpoints 1 and 2; n [nprob].
 * Score of a correspondence under (R, t) -- the reference's own restatement, pose_est_tools.py:150-203: X = the midpoint
 * triangulation (OpenGV triangulate2) in frame 1, score = (1 - f1 . X/|X|) + (1 - f2 . x2/|x2|), x2 = R^T (X - t); inlier
 * iff score < thr.  algorithm selects the minimal solver (sample size incl. the disambiguation points in brackets):
 * SOSVO_REL_FIVEPT (5 + 3; Nister's five-point algorithm, the reference's default "STEWENIUS" and "NISTER"),
 * SOSVO_REL_SEVENPT (7 + 2), SOSVO_REL_EIGHTPT (8).  Among all decompositions of all essential matrices of a sample the
 * (R, t) with the smallest summed score over the sample becomes the hypothesis.  Outputs as sosvo_ransac_abs_pose: T_out [nprob,3,4] = [R | t], the pose of viewpoint 2 in
 * frame 1 (X1 = R X2 + t), |t| = 1 (the scale of a 2D-2D motion is unobservable). */
#define SOSVO_REL_FIVEPT 5
#define SOSVO_REL_SEVENPT 7
#define SOSVO_REL_EIGHTPT 8
int32_t sosvo_ransac_rel_pose(sosvo_ctx* ctx, const double* f1, const double* f2, const int32_t* n, int32_t nprob,
                              int32_t stride, int32_t algorithm, double thr, int32_t max_iter, int32_t adaptive,
                              uint64_t seed, double* T_out, uint8_t* inlier_mask, int32_t* inlier_idx,
                              int32_t* n_inliers, int32_t* info, int32_t* hyp_counts);

/* ---- geometry: a7-a10, a12 -------------------------------------------------------------------
 * Elementwise restatements of the reference's numpy geometry (FP64; values agree with the
 * reference to rel-tol 1e-12, the transcendental functions come from the device math library).
 *
 * sosvo_pano_to_bearing: uv [n,2] pano pixel -> az, el, bearing [n,3] (any output may be NULL).
 *   Panorama.get_direction_angles_from_pixel_pano (omnistereo/panorama.py:650, closed form :635,:616),
 *   GUM.get_3D_point_from_angles_wrt_focus (gum.py:2564) -> map_angles_to_unit_sphere
 *   (camera_models.py:1031).  NaN outside [0,cols) x [0,rows).
 * sosvo_triangulate_midpoint: OmniStereoModel.get_triangulated_point_from_direction_angles
 *   (camera_models.py:3323, use_midpoint_triangulation=True) -> get_triangulated_midpoint (:2420)
 *   -> triangulate_for_skew_rays (:2481).  F_top_host / F_bot_host: HOST pointers to 3 doubles.
 * sosvo_triangulate2: pyopengv.triangulation_triangulate2(b1, b2, t12, R12) (pose_est_tools.py:359, :163):
 *   OpenGV's closed-form midpoint, g = R12 b2, [[b1.b1, -b1.g], [b1.g, -g.g]] lambda = (t12.b1, t12.g),
 *   X = (lambda0 b1 + t12 + lambda1 g) / 2 in frame 1.  b1, b2, X [n,3] device; t12 (3), R12 (9, row-major) HOST.
 * sosvo_range_filter: filter_panoramic_points_due_to_range (camera_models.py:3299) as called with
 *   homogeneous rows at pose_est_tools.py:372 (norm includes the trailing 1); limit <= 0 disables.
 * sosvo_rgbd_backproject: RGBDCamModel.get_XYZ (camera_models.py:835) + get_depth_Z (:781) at
 *   integer pixels + get_normalized_points (:203).  depth [rows,cols] f32; zeros -> NaN.        */
int32_t sosvo_pano_to_bearing(sosvo_ctx* ctx, const double* uv, int32_t n, double cols, double rows,
                              double pixel_size, double cyl_height_max, double* az, double* el,
                              double* bearing);
int32_t sosvo_triangulate_midpoint(sosvo_ctx* ctx, const double* az_top, const double* el_top,
                                   const double* az_bot, const double* el_bot, int32_t n,
                                   const double* F_top_host, const double* F_bot_host, double* X);
int32_t sosvo_triangulate2(sosvo_ctx* ctx, const double* b1, const double* b2, int32_t n, const double* t12_host,
                           const double* R12_host, double* X);
int32_t sosvo_range_filter(sosvo_ctx* ctx, const double* X, int32_t n, double min_range,
                           double max_range, uint8_t* ok);
int32_t sosvo_rgbd_backproject(sosvo_ctx* ctx, const float* depth, int32_t rows, int32_t cols,
                               const int32_t* u, const int32_t* v, int32_t n, double fx, double fy,
                               double cx, double cy, double focal_length_m, int32_t depth_is_Z,
                               double* xyz, double* bearing);

/* Constants of one calibrated omnistereo rig as the hot path needs them (HOST struct, passed by
 * pointer and copied into kernel arguments).  pano_* = {cols, rows, pixel_size, cyl_height_max}
 * of Panorama (panorama.py:142-172); F_* = mirror foci wrt [C]; ranges in model units
 * (pose_est_tools.py:306-308); gates of pose_est_tools.py:298-304 and :866; pct_good_matches of
 * FeatureMatcher (camera_models.py:379).                                                         */
typedef struct sosvo_rig {
  double pano_top[4];
  double pano_bot[4];
  double F_top[3];
  double F_bot[3];
  double min_range;        /* <= 0 disables */
  double max_range;        /* <= 0 disables */
  double stereo_min_disp;  /* v_top - v_bot >= this, < 0 disables (common_cv.py:182) */
  double stereo_max_hdiff; /* |u_top - u_bot| <= this, <= 0 disables (common_cv.py:177) */
  double f2f_max_hdiff;    /* frame-to-frame |du| gate, < 0 disables (pose_est_tools.py:245) */
  double pct_good_matches; /* 1.0 */
} sosvo_rig;

/* ---- stereo assemble: a6 + a7..a11 fused ------------------------------------------------------
 * Replaces OmniStereoModel.match_features_panoramic_top_bottom (camera_models.py:3027-3101) after
 * the per-bucket matcher call, and StereoPanoramicFrame.establish_stereo_correspondences
 * (pose_est_tools.py:339-397).  Problem p = frame*nmask + mask owns rows [p*cap, p*cap + n_*[p]) of
 * kp_* [.,2] f32 / desc_* [.,32] u8; keys/order [nframes*nmask, cap] come from sosvo_match_hamming
 * (query = bottom, train = top, k = 1) and sosvo_sort_matches.  Per frame, candidates are visited
 * in bucket order then rank order (empty buckets skipped), gated (|du|, dv), lifted to angles and
 * bearings, triangulated, range-filtered, and the survivors are written in that order:
 * m_top/m_bot [nframes,out_cap,2] f32, d_top/d_bot [nframes,out_cap,32] u8, X (frame [C]),
 * b_top/b_bot [nframes,out_cap,3] f64, M [nframes] i32 (count), n_cand [nframes] (optional:
 * candidates before the gates).  This is the PanoramicCorrespondences contract
 * (camera_models.py:291-362) in structure-of-arrays form.  A frame with more survivors than out_cap keeps the first
 * out_cap in that order (M = out_cap): what the reference's arrays hold, cut after out_cap rows.  */
int32_t sosvo_stereo_assemble(sosvo_ctx* ctx, const sosvo_rig* rig_host, const float* kp_top,
                              const float* kp_bot, const uint8_t* desc_top, const uint8_t* desc_bot,
                              const int32_t* n_top, const int32_t* n_bot, const uint32_t* keys,
                              const int32_t* order, int32_t nframes, int32_t nmask, int32_t cap,
                              int32_t out_cap, float* m_top, float* m_bot, uint8_t* d_top,
                              uint8_t* d_bot, double* X, double* b_top, double* b_bot, int32_t* M,
                              int32_t* n_cand);

/* ---- frame-to-frame assemble: a13 + the stacking of a14 ---------------------------------------
 * Replaces match_features_frame_to_frame (pose_est_tools.py:211-269) after the matcher call and
 * the correspondence stacking of TrackerStereoSE3.track_frame (:752-778).  Pair pr tracks frame
 * cur_frame[pr] against (key)frame ref_frame[pr]; keys_top, order_top, keys_bot, order_bot
 * [npairs, frame_cap] come from
 * matching query = current, train = reference per view (use q_slot = cur_frame, t_slot =
 * ref_frame).  Output rows [pr*corr_cap ...): top-view correspondences first, then bottom view,
 * each in rank order after the |du| gate: f = bearing of the current frame, p = 3-D point of the
 * reference frame, cam = 0/1, corr_q / corr_t = indices into the two frames, n [npairs] = count,
 * n_topview [npairs] (optional) = how many of them are top-view.                                 */
int32_t sosvo_f2f_assemble(sosvo_ctx* ctx, const sosvo_rig* rig_host, const float* m_top,
                           const float* m_bot, const double* X, const double* b_top,
                           const double* b_bot, const int32_t* M, int32_t frame_cap,
                           const int32_t* ref_frame, const int32_t* cur_frame, const uint32_t* keys_top,
                           const int32_t* order_top, const uint32_t* keys_bot, const int32_t* order_bot,
                           int32_t npairs, int32_t corr_cap, double* f, double* p, int32_t* cam,
                           int32_t* corr_q, int32_t* corr_t, int32_t* n, int32_t* n_topview);

/* ---- RGB-D (perspective camera) variant: a12 + a15 glue ----------------------------------------
 * Constants of one RGB-D camera (HOST struct): RGBDCamModel (omnistereo/camera_models.py:756-779) and the
 * RGBDFrame ranges (pose_est_tools.py:428-430, in the depth map's units).                           */
typedef struct sosvo_rgbd_cam {
  double fx, fy, cx, cy;
  double focal_length_m;   /* only used when depth is radial (depth_is_Z == 0, camera_models.py:781-799) */
  int32_t depth_is_Z;      /* 1: the depth map holds Z values */
  int32_t reserved;
  double min_range;        /* keep min_range <= |Z| <= max_range; <= 0 disables (pose_est_tools.py:570-592) */
  double max_range;
} sosvo_rgbd_cam;

/* sosvo_rgbd_assemble: RGBDFrame.establish_keypoints after detection (pose_est_tools.py:609-623) for nframes
 * frames: depth at the keypoints' integer pixels -> XYZ (get_XYZ), keypoints with zero depth (NaN) or out of
 * range dropped, bearings = normalised XYZ, survivors written in keypoint order.
 *   kp [nframes, cap, 2] f32, desc [nframes, cap, 32] u8, n [nframes] i32, depth [nframes, rows, cols] f32
 *   -> m [nframes, out_cap, 2] f32, d [nframes, out_cap, 32] u8, X, b [nframes, out_cap, 3] f64, M [nframes] i32
 * sosvo_f2f_assemble_central: TrackerRGBDSE3.track_frame steps 1-2 (pose_est_tools.py:896-913) for npairs pairs:
 *   keys / order [npairs, frame_cap] from sosvo_match_hamming (query = current frame's d, train = reference
 *   frame's d, k = 1) + sosvo_sort_matches; the first pct_good_matches * nq matches (:225), |du| <= max_hdiff
 *   (<= 0 disables, :245-247) -> f (bearings of the current frame), p (points of the reference frame)
 *   [npairs, corr_cap, 3] f64, corr_q / corr_t [npairs, corr_cap] i32 (indices into the frames), n [npairs].
 *   Feed f, p, n to sosvo_ransac_abs_pose / sosvo_refine_abs_pose with cam = NULL (central, :915, :937).   */
int32_t sosvo_rgbd_assemble(sosvo_ctx* ctx, const sosvo_rgbd_cam* cam_host, const float* kp, const uint8_t* desc,
                            const int32_t* n, const float* depth, int32_t nframes, int32_t rows, int32_t cols,
                            int32_t cap, int32_t out_cap, float* m, uint8_t* d, double* X, double* b, int32_t* M);
int32_t sosvo_f2f_assemble_central(sosvo_ctx* ctx, double pct_good_matches, double max_hdiff, const float* m,
                                   const double* X, const double* b, const int32_t* M, int32_t frame_cap,
                                   const int32_t* ref_frame, const int32_t* cur_frame, const uint32_t* keys,
                                   const int32_t* order, int32_t npairs, int32_t corr_cap, double* f, double* p,
                                   int32_t* corr_q, int32_t* corr_t, int32_t* n);

/* ---- the whole hot path for a batch of frame pairs (the throughput entry point) ----------------
 * One call = for B independent pairs (frames 2i = reference / keyframe, 2i+1 = current):
 *   OmniStereoModel.set_current_omni_image + StereoPanoramicFrame.__init__ for both frames
 *   (omnistereo/camera_models.py:3107-3120, pose_est_tools.py:271-402: unwrap both mirrors, medianBlur, gray,
 *   goodFeaturesToTrack per azimuthal mask, ORB descriptors, per-bucket top/bottom matching, gates, bearings,
 *   midpoint triangulation, range filter), then TrackerStereoSE3.track_frame (pose_est_tools.py:736-847:
 *   frame-to-frame matching per view, |du| gate, stacking, non-central RANSAC, LM on the inliers).
 * It sequences the stage functions above on the context's stream; nothing is allocated and the host is never
 * synchronised.  Detector: GFT (the reference default, pose_est_tools.py:684).
 *   omni         [2B, H, W, 3] u8 BGR (device)
 *   unwrap_table from sosvo_unwrap_prepare (model constant), mask_bits [2, rows, cols] u32 (set 0 = top mirror,
 *                set 1 = bottom mirror), pattern [512, 2] i8 (descriptor test points)
 *   workspace    caller-owned device memory, 256-byte aligned, >= sosvo_frame_pair_batch_workspace(cfg) bytes;
 *                contents on entry are irrelevant, intermediates (panoramas, keypoints, correspondences, ...) are
 *                left in it
 *   results      [B, 16] f64 (device): refined pose T = [R | t] row-major 3x4 (model units), n_inliers,
 *                n_correspondences, status (0 ok, 1 too few correspondences), best RANSAC iteration        */
typedef struct sosvo_batch_cfg {
  int32_t n_pairs;
  int32_t H, W;            /* omni frame */
  int32_t rows, cols;      /* panorama (both mirrors) */
  int32_t nmask;           /* azimuthal masks per mirror */
  int32_t kp_cap;          /* keypoint capacity per (frame, mirror, mask) */
  int32_t frame_cap;       /* stereo correspondences kept per frame (<= 16384) */
  int32_t median_ksize;    /* 11 (pose_est_tools.py:296) */
  int32_t max_corners;     /* per mask: FeatureMatcher.num_of_features (pose_est_tools.py:862) */
  int32_t edge;            /* ORB border, 31 */
  int32_t ransac_max_iter;
  int32_t ransac_adaptive; /* 1 = OpenGV's adaptive stop, 0 = exactly ransac_max_iter hypotheses */
  int32_t lm_max_iter;
  double quality;          /* 0.01 */
  double min_distance;     /* 5 */
  double ransac_threshold; /* 1 - cos(5 deg) (pose_est_tools.py:675-676) */
  uint64_t seed;           /* pair i samples with seed + i */
  float cos_a, sin_a;      /* descriptor orientation given to the GFT keypoints */
  int32_t ransac_flags;    /* SOSVO_FLAG_GP3P: generalised-P3P hypotheses (samples across both mirrors); 0: one-mirror P3P */
  int32_t reserved;        /* 0 */
} sosvo_batch_cfg;

size_t sosvo_frame_pair_batch_workspace(const sosvo_batch_cfg* cfg);
int32_t sosvo_frame_pair_batch(sosvo_ctx* ctx, const sosvo_rig* rig_host, const sosvo_batch_cfg* cfg_host,
                               const uint8_t* omni, const uint32_t* unwrap_table, const uint32_t* mask_bits,
                               const int8_t* pattern, void* workspace, size_t workspace_bytes, double* results);

/* The same batch split over n_streams (1..4) INTERNAL HIP streams: the VALU-bound median of one part overlaps the
 * latency-bound stages of the others (the medians take turns), which is worth ~15 % over one stream at a few hundred
 * pairs.  Part s = pairs [lo_s, hi_s) (contiguous, sizes differ by at most one) samples with seed + pair index, so the
 * results are those of sosvo_frame_pair_batch, bit for bit.  The parts start after the work already queued on the context's
 * stream and the context's stream continues after all of them (events; no host synchronisation); the internal streams
 * and their scratch memory are created on first use and live as long as the context.
 * workspace: sosvo_frame_pair_batch_streams_workspace(cfg, n_streams) bytes, 256-aligned; n_pairs >= n_streams. */
size_t sosvo_frame_pair_batch_streams_workspace(const sosvo_batch_cfg* cfg, int32_t n_streams);
int32_t sosvo_frame_pair_batch_streams(sosvo_ctx* ctx, const sosvo_rig* rig_host, const sosvo_batch_cfg* cfg_host,
                                       int32_t n_streams, const uint8_t* omni, const uint32_t* unwrap_table,
                                       const uint32_t* mask_bits, const int8_t* pattern, void* workspace,
                                       size_t workspace_bytes, double* results);

/* Back-to-back batches without the join: ..._enqueue does everything sosvo_frame_pair_batch_streams does EXCEPT making the
 * context's stream wait for the parts, so the parts of the next call start while this call's later stages still run (part s
 * of call k + 1 follows part s of call k on the same internal stream; the medians' turn-taking goes on across calls) -- the
 * overlap the Python engine (pipeline.OverlappedFramePairs) has, for a host that calls the C ABI.  The caller then
 *   - hands every call inputs that are ready on the context's stream at the time of the call (as before),
 *   - alternates (at least) two `results` buffers if something still reads the previous call's records, and
 *   - calls ..._join(ctx) before reading records on the context's stream: it makes that stream wait for every part of
 *     the LATEST call (hence, in stream order, of all earlier ones), and
 *   - treats the INPUTS like the records: until the join nothing orders the context's stream after the parts, so `omni`
 *     (and the model tables) of an un-joined call must not be overwritten on the context's stream -- double-buffer the
 *     frames or join first.
 * The workspace may be the same for every call: part s of every call uses the same slice on the same internal stream.
 * That holds only for the same split, so a call whose (n_pairs, n_streams) differ from the un-joined previous call's joins
 * internally first (correct, but without the overlap).  A call made while the context's stream is being captured into a
 * HIP graph fails with SOSVO_ERR_ARG while un-joined work is pending (join before capturing); after a join -- or after the
 * joined form sosvo_frame_pair_batch_streams -- every call is self-contained and capturable. */
int32_t sosvo_frame_pair_batch_streams_enqueue(sosvo_ctx* ctx, const sosvo_rig* rig_host, const sosvo_batch_cfg* cfg_host,
                                               int32_t n_streams, const uint8_t* omni, const uint32_t* unwrap_table,
                                               const uint32_t* mask_bits, const int8_t* pattern, void* workspace,
                                               size_t workspace_bytes, double* results);
int32_t sosvo_frame_pair_batch_streams_join(sosvo_ctx* ctx);

/* ---- Sequence mode: every frame's front end ONCE, tracking against keyframes -----------------------------
 * The reference's VO loop (run_VO, omnistereo/pose_est_tools.py:1416-1628) builds ONE StereoPanoramicFrame per image
 * (:1447-1463: unwrap, median, detection, static stereo, triangulation) and tracks it against the current KEYFRAME
 * (:1478, TrackerStereoSE3.track_frame :736-847); the keyframe policy (:1509-1565) depends on each tracked pose, so the
 * tracking is a serial chain while the front ends are independent.  These entry points split the batch path the same
 * way around a FRAME STORE in the caller's workspace: `slots` frame records (the frame's PanoramicCorrespondences as
 * SoA: pixels, descriptors, 3-D points, bearings, count), filled `window` frames at a time, tracked by slot numbers.
 *   cfg: sosvo_batch_cfg as for sosvo_frame_pair_batch; n_pairs = slot pairs one sosvo_sequence_track call may hold.
 *   workspace: sosvo_sequence_workspace(cfg, window, slots) bytes, 256-aligned, caller-owned; the SAME (cfg, window,
 *   slots) on every call.
 * sosvo_sequence_front_end: the front end of n_frames <= window omni frames [n_frames, H, W, 3] u8 (device) -> store
 *   slots first_slot .. first_slot + n_frames - 1.  Asynchronous.
 * sosvo_sequence_track: TrackerStereoSE3.track_frame for n_pairs (reference slot, current slot) pairs given as HOST
 *   arrays; pair i samples with seed + i -> results [n_pairs, 16] f64 (device) as sosvo_frame_pair_batch.  A pair's
 *   record depends only on its two slots' frames and its seed, so speculative tracking (every frame against its
 *   predecessor, in one call, before the keyframe decisions are known) gives exactly the records of the serial loop
 *   wherever the guess was right.  Asynchronous (up to 16 pairs travel as kernel arguments).
 * sosvo_sequence_copy_slot: store slot src -> dst (a frame promoted to keyframe outlives its window).  Asynchronous.
 * sosvo_sequence_frame_counts: the valid-correspondence counts (StereoPanoramicFrame.num_valid_keypoints, :372) of n
 *   slots -> HOST int32 array; SYNCHRONISES the context's stream (the one blocking call of the group).             */
size_t sosvo_sequence_workspace(const sosvo_batch_cfg* cfg, int32_t window, int32_t slots);
int32_t sosvo_sequence_front_end(sosvo_ctx* ctx, const sosvo_rig* rig_host, const sosvo_batch_cfg* cfg_host, int32_t window,
                                 int32_t slots, const uint8_t* omni, int32_t n_frames, int32_t first_slot,
                                 const uint32_t* unwrap_table, const uint32_t* mask_bits, const int8_t* pattern,
                                 void* workspace, size_t workspace_bytes);
int32_t sosvo_sequence_track(sosvo_ctx* ctx, const sosvo_rig* rig_host, const sosvo_batch_cfg* cfg_host, int32_t window,
                             int32_t slots, const int32_t* ref_slot_host, const int32_t* cur_slot_host, int32_t n_pairs,
                             uint64_t seed, void* workspace, size_t workspace_bytes, double* results);
int32_t sosvo_sequence_copy_slot(sosvo_ctx* ctx, const sosvo_batch_cfg* cfg_host, int32_t window, int32_t slots,
                                 int32_t src_slot, int32_t dst_slot, void* workspace, size_t workspace_bytes);
int32_t sosvo_sequence_frame_counts(sosvo_ctx* ctx, const sosvo_batch_cfg* cfg_host, int32_t window, int32_t slots,
                                    int32_t first_slot, int32_t n, void* workspace, size_t workspace_bytes,
                                    int32_t* counts_host);

/* ---- The RGB-D path for B independent frame pairs behind ONE call (BASELINE config 5) ------------------
 * RGBDFrame.establish_keypoints (omnistereo/pose_est_tools.py:600-623: [median,] gray, goodFeaturesToTrack on the
 * whole image or on RGBDFrame.mask, ORB descriptors, depth back-projection, NaN / range filter, bearings) for the
 * 2 B frames, then TrackerRGBDSE3.track_frame (:896-954: frame-to-frame matching, |du| gate, central RANSAC, LM)
 * for every pair (pair i tracks frame 2i+1 against frame 2i).  Sequences the stage entry points above on the
 * context's stream; same results as calling them one by one (vo_single_camera_sos_amd/pipeline.py:RGBDPairPipeline).
 *   bgr       [2B, rows, cols, 3] u8, depth [2B, rows, cols] f32 (camera depth units, 0 = no reading)
 *   mask_bits [1, rows, cols] u32 (bit 0 = pixel may hold a keypoint; all ones = RGBDFrame.mask None)
 *   pattern   [512, 2] i8 (the 256 ORB test pairs), workspace: sosvo_rgbd_pair_batch_workspace(cfg) bytes, 256-aligned
 *   results   [B, 16] f64 as sosvo_frame_pair_batch                                                             */
typedef struct sosvo_rgbd_batch_cfg {
  int32_t n_pairs;
  int32_t rows, cols;
  int32_t kp_cap;          /* keypoint capacity per frame; > 1024 selects the detector's whole-image variant */
  int32_t frame_cap;       /* keypoints with valid depth kept per frame */
  int32_t median_ksize;    /* 0: none (pose_est_tools.py:609 default) */
  int32_t max_corners;     /* FeatureMatcher.num_of_features, 1000 (pose_est_tools.py:691) */
  int32_t edge;            /* ORB border, 31 */
  int32_t ransac_max_iter;
  int32_t ransac_adaptive;
  int32_t lm_max_iter;
  int32_t flags;           /* SOSVO_FLAG_EPNP: "EPNP" (6-point samples); 0: "KNEIP" (P3P + 4th point); SOSVO_FLAG_GP3P: "GAO" / "GP3P";
                              SOSVO_FLAG_TWOPT: "TWOPT" -- passed through to sosvo_ransac_abs_pose */
  double quality, min_distance;        /* 0.01, 5 */
  double ransac_threshold;             /* 1 - cos(5 deg) */
  double pct_good_matches;             /* 1.0 (pose_est_tools.py:225) */
  double f2f_max_hdiff;                /* |du| gate in pixels: 0.5 * 2 * center_x (:956-958); <= 0: none */
  uint64_t seed;
  float cos_a, sin_a;
} sosvo_rgbd_batch_cfg;

size_t sosvo_rgbd_pair_batch_workspace(const sosvo_rgbd_batch_cfg* cfg);
int32_t sosvo_rgbd_pair_batch(sosvo_ctx* ctx, const sosvo_rgbd_cam* cam_host, const sosvo_rgbd_batch_cfg* cfg_host,
                              const uint8_t* bgr, const float* depth, const uint32_t* mask_bits, const int8_t* pattern,
                              void* workspace, size_t workspace_bytes, double* results);

/* ---- RGB-D sequence mode (the same split for the perspective path: demo_vo_rgbd.py's loop) ------------------
 * run_VO builds ONE RGBDFrame per image (pose_est_tools.py:1440-1446, :600-623) and tracks it against the current keyframe
 * (TrackerRGBDSE3.track_frame, :896-954).  Frame store, window, slots, seeds and the speculation argument exactly as in the
 * omnistereo sequence mode above; cfg: sosvo_rgbd_batch_cfg as for sosvo_rgbd_pair_batch (n_pairs = slot pairs one
 * ..._track call may hold).
 *   bgr [n_frames, rows, cols, 3] u8, depth [n_frames, rows, cols] f32 (device)                                      */
size_t sosvo_rgbd_sequence_workspace(const sosvo_rgbd_batch_cfg* cfg, int32_t window, int32_t slots);
int32_t sosvo_rgbd_sequence_front_end(sosvo_ctx* ctx, const sosvo_rgbd_cam* cam_host, const sosvo_rgbd_batch_cfg* cfg_host,
                                      int32_t window, int32_t slots, const uint8_t* bgr, const float* depth, int32_t n_frames,
                                      int32_t first_slot, const uint32_t* mask_bits, const int8_t* pattern, void* workspace,
                                      size_t workspace_bytes);
int32_t sosvo_rgbd_sequence_track(sosvo_ctx* ctx, const sosvo_rgbd_batch_cfg* cfg_host, int32_t window, int32_t slots,
                                  const int32_t* ref_slot_host, const int32_t* cur_slot_host, int32_t n_pairs, uint64_t seed,
                                  void* workspace, size_t workspace_bytes, double* results);
int32_t sosvo_rgbd_sequence_copy_slot(sosvo_ctx* ctx, const sosvo_rgbd_batch_cfg* cfg_host, int32_t window, int32_t slots,
                                      int32_t src_slot, int32_t dst_slot, void* workspace, size_t workspace_bytes);
int32_t sosvo_rgbd_sequence_frame_counts(sosvo_ctx* ctx, const sosvo_rgbd_batch_cfg* cfg_host, int32_t window, int32_t slots,
                                         int32_t first_slot, int32_t n, void* workspace, size_t workspace_bytes,
                                         int32_t* counts_host);

#ifdef __cplusplus
}
#endif
#endif /* SOSVO_H */
