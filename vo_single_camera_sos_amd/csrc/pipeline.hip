// sosvo_frame_pair_batch -- the whole hot path for B independent frame pairs behind ONE C-ABI call:
// the order of operations of OmniStereoModel.set_current_omni_image + StereoPanoramicFrame.__init__
// (omnistereo/camera_models.py:3107-3120, pose_est_tools.py:271-402) for both frames of every pair, then
// TrackerStereoSE3.track_frame (pose_est_tools.py:736-847) per pair.  It only sequences the stage entry points
// of this library on the context's stream (no host synchronisation, no allocation: every intermediate lives in
// the caller's workspace), so its results are those of calling the stages one by one.
#include "common.h"

namespace {

struct Carver {
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t count) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (count * sizeof(T) + 255) & ~(size_t)255;
    return p;
  }
};

struct Buffers {
  uint8_t *pano, *gray, *desc, *d_top, *d_bot, *mask;
  float *kp, *m_top, *m_bot;
  int32_t *n, *status, *s_order, *M, *n_cand, *ref_frame, *cur_frame, *o_top, *o_bot, *cam, *cq, *ct, *cn, *cn_top, *idx,
      *n_inl, *info, *lm_iters, *gray_rows;
  uint32_t *s_keys, *k_top, *k_bot;
  double *X, *b_top, *b_bot, *f, *p, *T_ransac, *T, *cam_off, *cam_rot, *lm_cost;
  size_t bytes;
};

Buffers carve(const sosvo_batch_cfg& c, void* ws) {
  Carver cv{reinterpret_cast<char*>(ws)};
  Buffers b;
  const size_t B = c.n_pairs, F = 2 * B, NI = 2 * F, P = NI * c.nmask, cap = c.kp_cap, Fc = c.frame_cap, Cc = 2 * Fc;
  const size_t npx = (size_t)c.rows * c.cols;
  const bool fused = c.median_ksize <= 1 || c.median_ksize == 3 || c.median_ksize == 5 || c.median_ksize == 11;  // K1 inside the median / gray kernel
  b.pano = fused ? nullptr : cv.take<uint8_t>(NI * npx * 3);
  b.gray = cv.take<uint8_t>(NI * npx);
  b.kp = cv.take<float>(P * cap * 2);
  b.n = cv.take<int32_t>(P);
  b.status = cv.take<int32_t>(P);
  b.desc = cv.take<uint8_t>(P * cap * 32);
  b.s_keys = cv.take<uint32_t>(P / 2 * cap);
  b.s_order = cv.take<int32_t>(P / 2 * cap);
  b.m_top = cv.take<float>(F * Fc * 2);
  b.m_bot = cv.take<float>(F * Fc * 2);
  b.d_top = cv.take<uint8_t>(F * Fc * 32);
  b.d_bot = cv.take<uint8_t>(F * Fc * 32);
  b.X = cv.take<double>(F * Fc * 3);
  b.b_top = cv.take<double>(F * Fc * 3);
  b.b_bot = cv.take<double>(F * Fc * 3);
  b.M = cv.take<int32_t>(F);
  b.n_cand = cv.take<int32_t>(F);
  b.ref_frame = cv.take<int32_t>(B);
  b.cur_frame = cv.take<int32_t>(B);
  b.k_top = cv.take<uint32_t>(B * Fc);
  b.k_bot = cv.take<uint32_t>(B * Fc);
  b.o_top = cv.take<int32_t>(B * Fc);
  b.o_bot = cv.take<int32_t>(B * Fc);
  b.f = cv.take<double>(B * Cc * 3);
  b.p = cv.take<double>(B * Cc * 3);
  b.cam = cv.take<int32_t>(B * Cc);
  b.cq = cv.take<int32_t>(B * Cc);
  b.ct = cv.take<int32_t>(B * Cc);
  b.cn = cv.take<int32_t>(B);
  b.cn_top = cv.take<int32_t>(B);
  b.T_ransac = cv.take<double>(B * 12);
  b.mask = cv.take<uint8_t>(B * Cc);
  b.idx = cv.take<int32_t>(B * Cc);
  b.n_inl = cv.take<int32_t>(B);
  b.info = cv.take<int32_t>(B * 4);
  b.T = cv.take<double>(B * 12);
  b.cam_off = cv.take<double>(2 * 3);
  b.cam_rot = cv.take<double>(2 * 9);
  b.lm_cost = cv.take<double>(B);
  b.lm_iters = cv.take<int32_t>(B);
  b.gray_rows = cv.take<int32_t>(4);
  b.bytes = cv.off;
  return b;
}

struct Foci {
  double top[3], bot[3];
};

// pair i tracks frame 2i + 1 against frame 2i; the two mirrors' foci are the non-central rig (identity rotations)
__global__ void batch_setup_kernel(int npairs, Foci foci, int32_t* __restrict__ ref_frame, int32_t* __restrict__ cur_frame,
                                   double* __restrict__ cam_off, double* __restrict__ cam_rot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npairs) {
    ref_frame[i] = 2 * i;
    cur_frame[i] = 2 * i + 1;
  }
  if (i < 3) {
    cam_off[i] = foci.top[i];
    cam_off[3 + i] = foci.bot[i];
  }
  if (i < 18) cam_rot[i] = (i % 9) % 4 == 0 ? 1.0 : 0.0;
}

// [B,16]: refined 3x4 pose, n_inliers, n_correspondences, status, RANSAC best iteration
__global__ void batch_results_kernel(int npairs, const double* __restrict__ T, const int32_t* __restrict__ n_inl,
                                     const int32_t* __restrict__ n_corr, const int32_t* __restrict__ info,
                                     double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npairs) return;
  for (int k = 0; k < 12; ++k) out[16 * i + k] = T[12 * i + k];
  out[16 * i + 12] = (double)n_inl[i];
  out[16 * i + 13] = (double)n_corr[i];
  out[16 * i + 14] = (double)info[4 * i + 2];
  out[16 * i + 15] = (double)info[4 * i + 0];
}

// ---- sequence mode: a frame STORE of `slots` frame records, a front end over a WINDOW of frames, tracking of slot pairs ----
struct SeqBuffers {
  // front-end temporaries of one window
  uint8_t *pano, *gray, *desc;
  float* kp;
  int32_t *n, *status, *s_order, *n_cand, *gray_rows;
  uint32_t* s_keys;
  // the frame store (PanoramicCorrespondences as SoA, slot-major)
  uint8_t *d_top, *d_bot;
  float *m_top, *m_bot;
  double *X, *b_top, *b_bot;
  int32_t* M;
  // tracking of up to cfg.n_pairs slot pairs per call
  uint8_t* mask;
  int32_t *ref_frame, *cur_frame, *o_top, *o_bot, *cam, *cq, *ct, *cn, *cn_top, *idx, *n_inl, *info, *lm_iters;
  uint32_t *k_top, *k_bot;
  double *f, *p, *T_ransac, *T, *cam_off, *cam_rot, *lm_cost;
  size_t bytes;
};

SeqBuffers carve_seq(const sosvo_batch_cfg& c, int window, int slots, void* ws) {
  Carver cv{reinterpret_cast<char*>(ws)};
  SeqBuffers b;
  const size_t B = c.n_pairs, F = window, NI = 2 * F, P = NI * c.nmask, cap = c.kp_cap, Fc = c.frame_cap, Cc = 2 * Fc, S = slots;
  const size_t npx = (size_t)c.rows * c.cols;
  const bool fused = c.median_ksize <= 1 || c.median_ksize == 3 || c.median_ksize == 5 || c.median_ksize == 11;
  b.pano = fused ? nullptr : cv.take<uint8_t>(NI * npx * 3);
  b.gray = cv.take<uint8_t>(NI * npx);
  b.kp = cv.take<float>(P * cap * 2);
  b.n = cv.take<int32_t>(P);
  b.status = cv.take<int32_t>(P);
  b.desc = cv.take<uint8_t>(P * cap * 32);
  b.s_keys = cv.take<uint32_t>(P / 2 * cap);
  b.s_order = cv.take<int32_t>(P / 2 * cap);
  b.n_cand = cv.take<int32_t>(F);
  b.gray_rows = cv.take<int32_t>(4);
  b.m_top = cv.take<float>(S * Fc * 2);
  b.m_bot = cv.take<float>(S * Fc * 2);
  b.d_top = cv.take<uint8_t>(S * Fc * 32);
  b.d_bot = cv.take<uint8_t>(S * Fc * 32);
  b.X = cv.take<double>(S * Fc * 3);
  b.b_top = cv.take<double>(S * Fc * 3);
  b.b_bot = cv.take<double>(S * Fc * 3);
  b.M = cv.take<int32_t>(S);
  b.ref_frame = cv.take<int32_t>(B);
  b.cur_frame = cv.take<int32_t>(B);
  b.k_top = cv.take<uint32_t>(B * Fc);
  b.k_bot = cv.take<uint32_t>(B * Fc);
  b.o_top = cv.take<int32_t>(B * Fc);
  b.o_bot = cv.take<int32_t>(B * Fc);
  b.f = cv.take<double>(B * Cc * 3);
  b.p = cv.take<double>(B * Cc * 3);
  b.cam = cv.take<int32_t>(B * Cc);
  b.cq = cv.take<int32_t>(B * Cc);
  b.ct = cv.take<int32_t>(B * Cc);
  b.cn = cv.take<int32_t>(B);
  b.cn_top = cv.take<int32_t>(B);
  b.T_ransac = cv.take<double>(B * 12);
  b.mask = cv.take<uint8_t>(B * Cc);
  b.idx = cv.take<int32_t>(B * Cc);
  b.n_inl = cv.take<int32_t>(B);
  b.info = cv.take<int32_t>(B * 4);
  b.T = cv.take<double>(B * 12);
  b.cam_off = cv.take<double>(2 * 3);
  b.cam_rot = cv.take<double>(2 * 9);
  b.lm_cost = cv.take<double>(B);
  b.lm_iters = cv.take<int32_t>(B);
  b.bytes = cv.off;
  return b;
}

constexpr int kSeqArgPairs = 16;  // slot pairs that travel as kernel arguments (no host-to-device copy per call)
struct SlotPairs {
  int32_t ref[kSeqArgPairs], cur[kSeqArgPairs];
};
__global__ void seq_setup_kernel(int npairs, SlotPairs sp, Foci foci, int32_t* __restrict__ ref_frame,
                                 int32_t* __restrict__ cur_frame, double* __restrict__ cam_off, double* __restrict__ cam_rot) {
  const int i = threadIdx.x;
  if (i < npairs && i < kSeqArgPairs) {
    ref_frame[i] = sp.ref[i];
    cur_frame[i] = sp.cur[i];
  }
  if (i < 3) {
    cam_off[i] = foci.top[i];
    cam_off[3 + i] = foci.bot[i];
  }
  if (i < 18) cam_rot[i] = (i % 9) % 4 == 0 ? 1.0 : 0.0;
}

// one frame record of the store -> another slot (a frame promoted to keyframe outlives its window)
__global__ void seq_copy_slot_kernel(int src, int dst, int Fc, float* m_top, float* m_bot, uint4* d_top, uint4* d_bot,
                                     double* X, double* b_top, double* b_bot, int32_t* M) {
  const int n = M[src];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const size_t s = (size_t)src * Fc + i, d = (size_t)dst * Fc + i;
    m_top[2 * d] = m_top[2 * s];
    m_top[2 * d + 1] = m_top[2 * s + 1];
    m_bot[2 * d] = m_bot[2 * s];
    m_bot[2 * d + 1] = m_bot[2 * s + 1];
    d_top[2 * d] = d_top[2 * s];
    d_top[2 * d + 1] = d_top[2 * s + 1];
    d_bot[2 * d] = d_bot[2 * s];
    d_bot[2 * d + 1] = d_bot[2 * s + 1];
    for (int k = 0; k < 3; ++k) {
      X[3 * d + k] = X[3 * s + k];
      b_top[3 * d + k] = b_top[3 * s + k];
      b_bot[3 * d + k] = b_bot[3 * s + k];
    }
  }
  __syncthreads();  // (every thread has read M[src] before it may be overwritten when src == dst is excluded by the caller)
  if (blockIdx.x == 0 && threadIdx.x == 0) M[dst] = n;
}

// ---- RGB-D variant (BASELINE config 5) ------------------------------------------------------------------
struct RgbdBuffers {
  uint8_t *gray, *desc, *d, *mask;
  float *kp, *m;
  int32_t *n, *status, *M, *ref_frame, *cur_frame, *order, *cq, *ct, *cn, *idx, *n_inl, *info, *lm_iters;
  uint32_t* keys;
  double *X, *b, *f, *p, *T_ransac, *T, *lm_cost;
  size_t bytes;
};

RgbdBuffers carve_rgbd(const sosvo_rgbd_batch_cfg& c, void* ws) {
  Carver cv{reinterpret_cast<char*>(ws)};
  RgbdBuffers b;
  const size_t B = c.n_pairs, F = 2 * B, cap = c.kp_cap, Fc = c.frame_cap, npx = (size_t)c.rows * c.cols;
  b.gray = cv.take<uint8_t>(F * npx);
  b.kp = cv.take<float>(F * cap * 2);
  b.n = cv.take<int32_t>(F);
  b.status = cv.take<int32_t>(F);
  b.desc = cv.take<uint8_t>(F * cap * 32);
  b.m = cv.take<float>(F * Fc * 2);
  b.d = cv.take<uint8_t>(F * Fc * 32);
  b.X = cv.take<double>(F * Fc * 3);
  b.b = cv.take<double>(F * Fc * 3);
  b.M = cv.take<int32_t>(F);
  b.ref_frame = cv.take<int32_t>(B);
  b.cur_frame = cv.take<int32_t>(B);
  b.keys = cv.take<uint32_t>(B * Fc);
  b.order = cv.take<int32_t>(B * Fc);
  b.f = cv.take<double>(B * Fc * 3);
  b.p = cv.take<double>(B * Fc * 3);
  b.cq = cv.take<int32_t>(B * Fc);
  b.ct = cv.take<int32_t>(B * Fc);
  b.cn = cv.take<int32_t>(B);
  b.T_ransac = cv.take<double>(B * 12);
  b.mask = cv.take<uint8_t>(B * Fc);
  b.idx = cv.take<int32_t>(B * Fc);
  b.n_inl = cv.take<int32_t>(B);
  b.info = cv.take<int32_t>(B * 4);
  b.T = cv.take<double>(B * 12);
  b.lm_cost = cv.take<double>(B);
  b.lm_iters = cv.take<int32_t>(B);
  b.bytes = cv.off;
  return b;
}

// RGB-D sequence mode: a store of `slots` RGBDFrame records (keypoints with valid depth, descriptors, points, bearings)
struct RgbdSeqBuffers {
  uint8_t *gray, *desc;                        // front-end temporaries of one window
  float* kp;
  int32_t *n, *status;
  uint8_t* d;                                  // the frame store (slot-major)
  float* m;
  double *X, *b;
  int32_t* M;
  uint8_t* mask;                               // tracking of up to cfg.n_pairs slot pairs per call
  int32_t *ref_frame, *cur_frame, *order, *cq, *ct, *cn, *idx, *n_inl, *info, *lm_iters;
  uint32_t* keys;
  double *f, *p, *T_ransac, *T, *lm_cost;
  size_t bytes;
};

RgbdSeqBuffers carve_rgbd_seq(const sosvo_rgbd_batch_cfg& c, int window, int slots, void* ws) {
  Carver cv{reinterpret_cast<char*>(ws)};
  RgbdSeqBuffers b;
  const size_t B = c.n_pairs, F = window, S = slots, cap = c.kp_cap, Fc = c.frame_cap, npx = (size_t)c.rows * c.cols;
  b.gray = cv.take<uint8_t>(F * npx);
  b.kp = cv.take<float>(F * cap * 2);
  b.n = cv.take<int32_t>(F);
  b.status = cv.take<int32_t>(F);
  b.desc = cv.take<uint8_t>(F * cap * 32);
  b.m = cv.take<float>(S * Fc * 2);
  b.d = cv.take<uint8_t>(S * Fc * 32);
  b.X = cv.take<double>(S * Fc * 3);
  b.b = cv.take<double>(S * Fc * 3);
  b.M = cv.take<int32_t>(S);
  b.ref_frame = cv.take<int32_t>(B);
  b.cur_frame = cv.take<int32_t>(B);
  b.keys = cv.take<uint32_t>(B * Fc);
  b.order = cv.take<int32_t>(B * Fc);
  b.f = cv.take<double>(B * Fc * 3);
  b.p = cv.take<double>(B * Fc * 3);
  b.cq = cv.take<int32_t>(B * Fc);
  b.ct = cv.take<int32_t>(B * Fc);
  b.cn = cv.take<int32_t>(B);
  b.T_ransac = cv.take<double>(B * 12);
  b.mask = cv.take<uint8_t>(B * Fc);
  b.idx = cv.take<int32_t>(B * Fc);
  b.n_inl = cv.take<int32_t>(B);
  b.info = cv.take<int32_t>(B * 4);
  b.T = cv.take<double>(B * 12);
  b.lm_cost = cv.take<double>(B);
  b.lm_iters = cv.take<int32_t>(B);
  b.bytes = cv.off;
  return b;
}

__global__ void rgbd_seq_setup_kernel(int npairs, SlotPairs sp, int32_t* __restrict__ ref_frame, int32_t* __restrict__ cur_frame) {
  const int i = threadIdx.x;
  if (i < npairs && i < kSeqArgPairs) {
    ref_frame[i] = sp.ref[i];
    cur_frame[i] = sp.cur[i];
  }
}

__global__ void rgbd_seq_copy_slot_kernel(int src, int dst, int Fc, float* m, uint4* d, double* X, double* b, int32_t* M) {
  const int n = M[src];
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const size_t s = (size_t)src * Fc + i, t = (size_t)dst * Fc + i;
    m[2 * t] = m[2 * s];
    m[2 * t + 1] = m[2 * s + 1];
    d[2 * t] = d[2 * s];
    d[2 * t + 1] = d[2 * s + 1];
    for (int k = 0; k < 3; ++k) {
      X[3 * t + k] = X[3 * s + k];
      b[3 * t + k] = b[3 * s + k];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) M[dst] = n;
}

__global__ void pair_index_kernel(int npairs, int32_t* __restrict__ ref_frame, int32_t* __restrict__ cur_frame) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npairs) {
    ref_frame[i] = 2 * i;
    cur_frame[i] = 2 * i + 1;
  }
}

}  // namespace

extern "C" {

size_t sosvo_frame_pair_batch_workspace(const sosvo_batch_cfg* cfg) {
  if (!cfg || cfg->n_pairs <= 0 || cfg->rows <= 0 || cfg->cols <= 0 || cfg->nmask <= 0 || cfg->kp_cap <= 0 ||
      cfg->frame_cap <= 0)
    return 0;
  return carve(*cfg, nullptr).bytes;
}

#define STAGE(call)              \
  do {                           \
    rc = (call);                 \
    if (rc != SOSVO_OK) return rc; \
  } while (0)

// The image front end + static stereo of F frames (OmniStereoModel.set_current_omni_image + StereoPanoramicFrame.__init__,
// camera_models.py:3107-3120, pose_est_tools.py:271-402): K1, K2 + K3, K4, K6 over all 2 F panoramas, per-bucket top/bottom
// matching, gates, bearings, midpoint triangulation, range filter -> F frame records written at out_* (frame-major).
struct FrontEndTmp {
  uint8_t *pano, *gray, *desc;
  float* kp;
  int32_t *n, *status, *s_order, *n_cand, *gray_rows;
  uint32_t* s_keys;
};
struct FrameStore {
  float *m_top, *m_bot;
  uint8_t *d_top, *d_bot;
  double *X, *b_top, *b_bot;
  int32_t* M;
};
static int32_t run_front_end(sosvo_ctx* ctx, const sosvo_rig* rig, const sosvo_batch_cfg* cfg, int F, const uint8_t* omni,
                             const uint32_t* unwrap_table, const uint32_t* mask_bits, const int8_t* pattern,
                             const FrontEndTmp& t, const FrameStore& o, hipEvent_t median_wait, hipEvent_t median_done) {
  const int NI = 2 * F, NM = cfg->nmask, cap = cfg->kp_cap, Fc = cfg->frame_cap;
  const int h = F * NM;  // problems of one view: (frame, mask); the top view's come first (view-major images)
  int32_t rc;
  if (median_wait) SOSVO_HIP(ctx, hipStreamWaitEvent(ctx->stream, median_wait, 0));
  const bool rows_known = cfg->median_ksize == 3 || cfg->median_ksize == 5 || cfg->median_ksize == 11;
  if (rows_known) {
    // rows beyond the reach of GFT on the masks and of ORB.compute on its keypoints are not computed (the range is a
    // model constant; a 2-workgroup kernel per call keeps this entry point free of caller-side state)
    STAGE(sosvo_gray_rows_needed(ctx, mask_bits, 2, cfg->rows, cfg->cols, NM, cfg->edge, pattern, cfg->cos_a, cfg->sin_a,
                                 t.gray_rows));
    STAGE(sosvo_unwrap_median_gray_rows(ctx, omni, unwrap_table, F, cfg->H, cfg->W, cfg->rows, cfg->cols, cfg->median_ksize,
                                        t.gray_rows, t.gray));
  } else if (cfg->median_ksize <= 1) {  // no median: unwrap straight to gray, every row
    STAGE(sosvo_unwrap_median_gray_rows(ctx, omni, unwrap_table, F, cfg->H, cfg->W, cfg->rows, cfg->cols, cfg->median_ksize,
                                        nullptr, t.gray));
  } else {
    STAGE(sosvo_unwrap_table(ctx, omni, unwrap_table, F, cfg->H, cfg->W, cfg->rows, cfg->cols, t.pano));
    STAGE(sosvo_median_gray(ctx, t.pano, NI, cfg->rows, cfg->cols, cfg->median_ksize, t.gray));
  }
  if (median_done) SOSVO_HIP(ctx, hipEventRecord(median_done, ctx->stream));
  STAGE(sosvo_detect_gft(ctx, t.gray, mask_bits, NI, F, cfg->rows, cfg->cols, NM, cfg->quality, cfg->min_distance,
                         cfg->max_corners, cap, t.kp, t.n, t.status));
  STAGE(sosvo_describe_orb_rows(ctx, t.gray, NI, cfg->rows, cfg->cols, NM, cap, t.kp, t.n, cfg->cos_a, cfg->sin_a, pattern,
                                cfg->edge, rows_known ? t.gray_rows : nullptr, t.desc));
  // static stereo per frame: query = bottom view, train = top view, bucket by bucket
  const float *kp_top = t.kp, *kp_bot = t.kp + (size_t)h * cap * 2;
  const uint8_t *desc_top = t.desc, *desc_bot = t.desc + (size_t)h * cap * 32;
  const int32_t *n_top = t.n, *n_bot = t.n + h;
  STAGE(sosvo_match_hamming(ctx, desc_bot, desc_top, n_bot, n_top, nullptr, nullptr, h, cap, cap, 1, t.s_keys));
  STAGE(sosvo_sort_matches(ctx, t.s_keys, n_bot, nullptr, h, cap, t.s_order));
  STAGE(sosvo_stereo_assemble(ctx, rig, kp_top, kp_bot, desc_top, desc_bot, n_top, n_bot, t.s_keys, t.s_order, F, NM, cap,
                              Fc, o.m_top, o.m_bot, o.d_top, o.d_bot, o.X, o.b_top, o.b_bot, o.M, t.n_cand));
  return SOSVO_OK;
}

// TrackerStereoSE3.track_frame (pose_est_tools.py:736-847) for B (reference slot, current slot) pairs of a frame store:
// frame-to-frame matching per view (query = current frame, train = reference frame), |du| gate, stacking, non-central
// RANSAC (problem i samples with seed + i), LM on the inliers, [B,16] records.
struct TrackTmp {
  int32_t *ref_frame, *cur_frame, *o_top, *o_bot, *cam, *cq, *ct, *cn, *cn_top, *idx, *n_inl, *info, *lm_iters;
  uint32_t *k_top, *k_bot;
  uint8_t* mask;
  double *f, *p, *T_ransac, *T, *cam_off, *cam_rot, *lm_cost;
};
static int32_t run_tracking(sosvo_ctx* ctx, const sosvo_rig* rig, const sosvo_batch_cfg* cfg, int B, uint64_t seed,
                            const FrameStore& s, const TrackTmp& b, double* results) {
  const int Fc = cfg->frame_cap, Cc = 2 * Fc;
  int32_t rc;
  STAGE(sosvo_match_hamming(ctx, s.d_top, s.d_top, s.M, s.M, b.cur_frame, b.ref_frame, B, Fc, Fc, 1, b.k_top));
  STAGE(sosvo_sort_matches(ctx, b.k_top, s.M, b.cur_frame, B, Fc, b.o_top));
  STAGE(sosvo_match_hamming(ctx, s.d_bot, s.d_bot, s.M, s.M, b.cur_frame, b.ref_frame, B, Fc, Fc, 1, b.k_bot));
  STAGE(sosvo_sort_matches(ctx, b.k_bot, s.M, b.cur_frame, B, Fc, b.o_bot));
  STAGE(sosvo_f2f_assemble(ctx, rig, s.m_top, s.m_bot, s.X, s.b_top, s.b_bot, s.M, Fc, b.ref_frame, b.cur_frame, b.k_top,
                           b.o_top, b.k_bot, b.o_bot, B, Cc, b.f, b.p, b.cam, b.cq, b.ct, b.cn, b.cn_top));
  // 3D-2D absolute pose: RANSAC, then LM on the inliers
  STAGE(sosvo_ransac_abs_pose(ctx, b.f, b.p, b.cam, b.cam_off, b.cam_rot, 2,
                              SOSVO_FLAG_CAM_ROT_IDENTITY | (cfg->ransac_flags & SOSVO_FLAG_GP3P), b.cn, B, Cc,
                              cfg->ransac_threshold, cfg->ransac_max_iter, cfg->ransac_adaptive, seed, b.T_ransac,
                              b.mask, b.idx, b.n_inl, b.info, nullptr));
  SOSVO_HIP(ctx, hipMemcpyAsync(b.T, b.T_ransac, sizeof(double) * 12 * B, hipMemcpyDeviceToDevice, ctx->stream));
  STAGE(sosvo_refine_abs_pose(ctx, b.f, b.p, b.cam, b.cam_off, b.cam_rot, 2, b.cn, B, Cc, b.idx, b.n_inl, cfg->lm_max_iter,
                              b.T, b.lm_cost, b.lm_iters));
  SOSVO_LAUNCH(ctx, batch_results_kernel, dim3(cdiv(B, 256)), dim3(256), 0, ctx->stream, B, b.T, b.n_inl, b.cn, b.info,
               results);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

// (median_wait / median_done: the token that serialises the VALU-bound median launches of the parts of a multi-stream
// batch, see sosvo_frame_pair_batch_streams; nullptr for a plain call)
static int32_t run_frame_pairs(sosvo_ctx* ctx, const sosvo_rig* rig, const sosvo_batch_cfg* cfg, const uint8_t* omni,
                               const uint32_t* unwrap_table, const uint32_t* mask_bits, const int8_t* pattern,
                               void* workspace, size_t workspace_bytes, double* results, hipEvent_t median_wait,
                               hipEvent_t median_done) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, rig && cfg && omni && unwrap_table && mask_bits && pattern && workspace && results, "null pointer");
  SOSVO_REQUIRE(ctx, cfg->n_pairs > 0 && cfg->n_pairs <= 8192, "n_pairs out of range (1..8192)");
  SOSVO_REQUIRE(ctx, cfg->nmask >= 1 && cfg->nmask <= 32 && cfg->kp_cap > 0 && cfg->kp_cap <= 4096, "nmask / kp_cap out of range");
  SOSVO_REQUIRE(ctx, cfg->frame_cap > 0 && cfg->frame_cap <= 16384, "frame_cap out of range (max 16384)");
  SOSVO_REQUIRE(ctx, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  const Buffers b = carve(*cfg, workspace);
  SOSVO_REQUIRE(ctx, workspace_bytes >= b.bytes, "workspace too small (see sosvo_frame_pair_batch_workspace)");
  const int B = cfg->n_pairs, F = 2 * B;
  int32_t rc;
  Foci foci;
  for (int k = 0; k < 3; ++k) {
    foci.top[k] = rig->F_top[k];
    foci.bot[k] = rig->F_bot[k];
  }
  SOSVO_LAUNCH(ctx, batch_setup_kernel, dim3(cdiv(B > 18 ? B : 18, 256)), dim3(256), 0, ctx->stream, B, foci, b.ref_frame,
               b.cur_frame, b.cam_off, b.cam_rot);
  SOSVO_LAUNCH_CHECK(ctx);
  const FrontEndTmp fe{b.pano, b.gray, b.desc, b.kp, b.n, b.status, b.s_order, b.n_cand, b.gray_rows, b.s_keys};
  const FrameStore store{b.m_top, b.m_bot, b.d_top, b.d_bot, b.X, b.b_top, b.b_bot, b.M};
  STAGE(run_front_end(ctx, rig, cfg, F, omni, unwrap_table, mask_bits, pattern, fe, store, median_wait, median_done));
  const TrackTmp tt{b.ref_frame, b.cur_frame, b.o_top, b.o_bot, b.cam, b.cq, b.ct, b.cn, b.cn_top, b.idx, b.n_inl, b.info,
                    b.lm_iters, b.k_top, b.k_bot, b.mask, b.f, b.p, b.T_ransac, b.T, b.cam_off, b.cam_rot, b.lm_cost};
  return run_tracking(ctx, rig, cfg, B, cfg->seed, store, tt, results);
}

int32_t sosvo_frame_pair_batch(sosvo_ctx* ctx, const sosvo_rig* rig, const sosvo_batch_cfg* cfg, const uint8_t* omni,
                               const uint32_t* unwrap_table, const uint32_t* mask_bits, const int8_t* pattern,
                               void* workspace, size_t workspace_bytes, double* results) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, rig && cfg && omni && unwrap_table && mask_bits && pattern && workspace && results, "null pointer");
  return run_frame_pairs(ctx, rig, cfg, omni, unwrap_table, mask_bits, pattern, workspace, workspace_bytes, results, nullptr,
                         nullptr);
}

// ---- the same batch split over internal HIP streams ------------------------------------------------------
static void part_range(int n, int part, int parts, int* lo, int* hi) {  // contiguous blocks, sizes differ by at most one
  const int base = n / parts, extra = n % parts;
  *lo = part * base + (part < extra ? part : extra);
  *hi = *lo + base + (part < extra ? 1 : 0);
}

size_t sosvo_frame_pair_batch_streams_workspace(const sosvo_batch_cfg* cfg, int32_t n_streams) {
  if (!cfg || n_streams < 1 || n_streams > kSosvoMaxSubStreams || cfg->n_pairs < n_streams) return 0;
  size_t total = 0;
  for (int s = 0; s < n_streams; ++s) {
    int lo, hi;
    part_range(cfg->n_pairs, s, n_streams, &lo, &hi);
    sosvo_batch_cfg c = *cfg;
    c.n_pairs = hi - lo;
    const size_t bytes = sosvo_frame_pair_batch_workspace(&c);
    if (bytes == 0) return 0;
    total += (bytes + 255) & ~(size_t)255;
  }
  return total;
}

static int32_t sosvo_frame_pair_batch_streams_impl(sosvo_ctx* ctx, const sosvo_rig* rig, const sosvo_batch_cfg* cfg, int32_t n_streams,
                             const uint8_t* omni, const uint32_t* unwrap_table, const uint32_t* mask_bits,
                             const int8_t* pattern, void* workspace, size_t workspace_bytes, double* results, bool join) {
  SOSVO_REQUIRE(ctx, n_streams >= 1 && n_streams <= kSosvoMaxSubStreams, "n_streams out of range (1..4)");
  SOSVO_REQUIRE(ctx, cfg->n_pairs >= n_streams && cfg->n_pairs <= 8192, "n_pairs out of range (n_streams..8192)");
  SOSVO_REQUIRE(ctx, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  const size_t need = sosvo_frame_pair_batch_streams_workspace(cfg, n_streams);
  SOSVO_REQUIRE(ctx, need > 0 && workspace_bytes >= need, "workspace too small (see sosvo_frame_pair_batch_streams_workspace)");
  // internal sub-contexts: one stream and one scratch workspace each, created on first use, kept for the context's life
  while (ctx->n_sub < n_streams) {
    const int i = ctx->n_sub;
    if (i == 0) SOSVO_HIP(ctx, hipEventCreateWithFlags(&ctx->sub_begin, hipEventDisableTiming));
    hipStream_t st;
    SOSVO_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int32_t rc = sosvo_create(&ctx->sub[i], ctx->device, st);
    if (rc != SOSVO_OK) {
      (void)hipStreamDestroy(st);
      return sosvo_fail(ctx, rc, __func__, "cannot create an internal stream context");
    }
    ctx->sub[i]->hint_shared_device = 1;  // the parts run side by side
    ctx->sub[i]->hint_score_fp64_only = ctx->hint_score_fp64_only;
    SOSVO_HIP(ctx, hipEventCreateWithFlags(&ctx->sub_done[i], hipEventDisableTiming));
    SOSVO_HIP(ctx, hipEventCreateWithFlags(&ctx->sub_median[i], hipEventDisableTiming));
    ctx->n_sub = i + 1;
  }
  if (ctx->sub_pending) {
    // un-joined parts of an earlier ..._enqueue call may still run.  A stream capture cannot depend on them (their events
    // were recorded outside it), and a different split would lay this call's workspace slices over slices an earlier
    // part still uses on ANOTHER stream (a slice is ordered by its own part stream only): join first in both cases.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    SOSVO_HIP(ctx, hipStreamIsCapturing(ctx->stream, &cap));
    SOSVO_REQUIRE(ctx, cap == hipStreamCaptureStatusNone,
                  "un-joined sosvo_frame_pair_batch_streams_enqueue work is pending: call ..._join before capturing");
    if (ctx->sub_last != n_streams || ctx->sub_last_pairs != cfg->n_pairs) {
      for (int s = 0; s < ctx->sub_last; ++s) SOSVO_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->sub_done[s], 0));
      ctx->sub_pending = 0;
    }
  }
  // every part starts after the work already queued on the caller's stream ...
  SOSVO_HIP(ctx, hipEventRecord(ctx->sub_begin, ctx->stream));
  char* ws = reinterpret_cast<char*>(workspace);
  // ... and the medians take turns: part s waits for the median of part s - 1; the first part waits for the LAST part of
  // the previous call only while that call is un-joined (back-to-back ..._enqueue calls).  After a join the chain starts
  // afresh: everything earlier is already ordered before sub_begin, and an event recorded before a stream capture began
  // must not be waited on inside it.
  hipEvent_t token = (ctx->sub_pending && ctx->sub_last > 0) ? ctx->sub_median[ctx->sub_last - 1] : nullptr;
  for (int s = 0; s < n_streams; ++s) {
    int lo, hi;
    part_range(cfg->n_pairs, s, n_streams, &lo, &hi);
    sosvo_batch_cfg c = *cfg;
    c.n_pairs = hi - lo;
    c.seed = cfg->seed + (uint64_t)lo;  // pair i samples with seed + i, whatever the split
    const size_t bytes = (sosvo_frame_pair_batch_workspace(&c) + 255) & ~(size_t)255;
    sosvo_ctx* sc = ctx->sub[s];
    SOSVO_HIP(ctx, hipStreamWaitEvent(sc->stream, ctx->sub_begin, 0));
    const int32_t rc = run_frame_pairs(sc, rig, &c, omni + (size_t)2 * lo * cfg->H * cfg->W * 3, unwrap_table, mask_bits, pattern,
                                       ws, bytes, results + (size_t)16 * lo, token, ctx->sub_median[s]);
    if (rc != SOSVO_OK) {
      // the parts launched so far (and whatever this one enqueued before failing) still run: the caller's stream must
      // not overtake them
      (void)hipEventRecord(ctx->sub_done[s], sc->stream);
      for (int k = 0; k <= s; ++k) (void)hipStreamWaitEvent(ctx->stream, ctx->sub_done[k], 0);
      if (ctx->sub_pending)  // parts of the previous un-joined call beyond s: the caller's stream waits for them as well
        for (int k = s + 1; k < ctx->sub_last; ++k) (void)hipStreamWaitEvent(ctx->stream, ctx->sub_done[k], 0);
      ctx->sub_pending = 0;
      return sosvo_fail(ctx, rc, __func__, sc->err);
    }
    token = ctx->sub_median[s];
    SOSVO_HIP(ctx, hipEventRecord(ctx->sub_done[s], sc->stream));
    ws += bytes;
  }
  ctx->sub_last = n_streams;
  ctx->sub_last_pairs = cfg->n_pairs;
  ctx->sub_pending = join ? 0 : 1;
  // ... and the caller's stream continues after all of them
  if (join)
    for (int s = 0; s < n_streams; ++s) SOSVO_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->sub_done[s], 0));
  return SOSVO_OK;
}

int32_t sosvo_frame_pair_batch_streams(sosvo_ctx* ctx, const sosvo_rig* rig, const sosvo_batch_cfg* cfg, int32_t n_streams,
                                       const uint8_t* omni, const uint32_t* unwrap_table, const uint32_t* mask_bits,
                                       const int8_t* pattern, void* workspace, size_t workspace_bytes, double* results) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, rig && cfg && omni && unwrap_table && mask_bits && pattern && workspace && results, "null pointer");
  return sosvo_frame_pair_batch_streams_impl(ctx, rig, cfg, n_streams, omni, unwrap_table, mask_bits, pattern, workspace, workspace_bytes, results, true);
}

int32_t sosvo_frame_pair_batch_streams_enqueue(sosvo_ctx* ctx, const sosvo_rig* rig, const sosvo_batch_cfg* cfg,
                                               int32_t n_streams, const uint8_t* omni, const uint32_t* unwrap_table,
                                               const uint32_t* mask_bits, const int8_t* pattern, void* workspace,
                                               size_t workspace_bytes, double* results) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, rig && cfg && omni && unwrap_table && mask_bits && pattern && workspace && results, "null pointer");
  return sosvo_frame_pair_batch_streams_impl(ctx, rig, cfg, n_streams, omni, unwrap_table, mask_bits, pattern, workspace, workspace_bytes, results, false);
}

int32_t sosvo_frame_pair_batch_streams_join(sosvo_ctx* ctx) {
  SOSVO_ENTER(ctx);
  for (int s = 0; s < ctx->sub_last; ++s) SOSVO_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->sub_done[s], 0));
  ctx->sub_pending = 0;
  return SOSVO_OK;
}

// ---- sequence mode ---------------------------------------------------------------------------------------
static bool seq_shape_ok(const sosvo_batch_cfg* cfg, int32_t window, int32_t slots) {
  return cfg && cfg->n_pairs > 0 && cfg->n_pairs <= 8192 && cfg->rows > 0 && cfg->cols > 0 && cfg->nmask >= 1 && cfg->nmask <= 32 &&
         cfg->kp_cap > 0 && cfg->kp_cap <= 4096 && cfg->frame_cap > 0 && cfg->frame_cap <= 16384 && window >= 1 && window <= 16384 &&
         slots >= window && slots <= 65535;
}

size_t sosvo_sequence_workspace(const sosvo_batch_cfg* cfg, int32_t window, int32_t slots) {
  if (!seq_shape_ok(cfg, window, slots)) return 0;
  return carve_seq(*cfg, window, slots, nullptr).bytes;
}

int32_t sosvo_sequence_front_end(sosvo_ctx* ctx, const sosvo_rig* rig, const sosvo_batch_cfg* cfg, int32_t window, int32_t slots,
                                 const uint8_t* omni, int32_t n_frames, int32_t first_slot, const uint32_t* unwrap_table,
                                 const uint32_t* mask_bits, const int8_t* pattern, void* workspace, size_t workspace_bytes) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, rig && cfg && omni && unwrap_table && mask_bits && pattern && workspace, "null pointer");
  SOSVO_REQUIRE(ctx, seq_shape_ok(cfg, window, slots), "bad sequence configuration (window, slots, capacities)");
  SOSVO_REQUIRE(ctx, n_frames >= 0 && n_frames <= window && first_slot >= 0 && first_slot + n_frames <= slots,
                "frames do not fit the window / the store");
  SOSVO_REQUIRE(ctx, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  const SeqBuffers b = carve_seq(*cfg, window, slots, workspace);
  SOSVO_REQUIRE(ctx, workspace_bytes >= b.bytes, "workspace too small (see sosvo_sequence_workspace)");
  if (n_frames == 0) return SOSVO_OK;
  const size_t Fc = cfg->frame_cap, o = (size_t)first_slot * Fc;
  const FrontEndTmp fe{b.pano, b.gray, b.desc, b.kp, b.n, b.status, b.s_order, b.n_cand, b.gray_rows, b.s_keys};
  const FrameStore out{b.m_top + 2 * o, b.m_bot + 2 * o, b.d_top + 32 * o, b.d_bot + 32 * o,
                       b.X + 3 * o, b.b_top + 3 * o, b.b_bot + 3 * o, b.M + first_slot};
  return run_front_end(ctx, rig, cfg, n_frames, omni, unwrap_table, mask_bits, pattern, fe, out, nullptr, nullptr);
}

int32_t sosvo_sequence_track(sosvo_ctx* ctx, const sosvo_rig* rig, const sosvo_batch_cfg* cfg, int32_t window, int32_t slots,
                             const int32_t* ref_slot, const int32_t* cur_slot, int32_t n_pairs, uint64_t seed, void* workspace,
                             size_t workspace_bytes, double* results) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, rig && cfg && ref_slot && cur_slot && workspace && results, "null pointer");
  SOSVO_REQUIRE(ctx, seq_shape_ok(cfg, window, slots), "bad sequence configuration (window, slots, capacities)");
  SOSVO_REQUIRE(ctx, n_pairs >= 0 && n_pairs <= cfg->n_pairs, "more slot pairs than cfg->n_pairs");
  SOSVO_REQUIRE(ctx, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  const SeqBuffers b = carve_seq(*cfg, window, slots, workspace);
  SOSVO_REQUIRE(ctx, workspace_bytes >= b.bytes, "workspace too small (see sosvo_sequence_workspace)");
  if (n_pairs == 0) return SOSVO_OK;
  for (int i = 0; i < n_pairs; ++i)   // host arrays: a bad slot would index outside the store on the device
    SOSVO_REQUIRE(ctx, ref_slot[i] >= 0 && ref_slot[i] < slots && cur_slot[i] >= 0 && cur_slot[i] < slots, "slot out of range");
  Foci foci;
  for (int k = 0; k < 3; ++k) {
    foci.top[k] = rig->F_top[k];
    foci.bot[k] = rig->F_bot[k];
  }
  SlotPairs sp;
  for (int i = 0; i < kSeqArgPairs; ++i) {
    sp.ref[i] = i < n_pairs ? ref_slot[i] : 0;
    sp.cur[i] = i < n_pairs ? cur_slot[i] : 0;
  }
  if (n_pairs > kSeqArgPairs) {  // larger batches: the slot lists are copied (pageable host memory: the copy returns when staged)
    SOSVO_HIP(ctx, hipMemcpyAsync(b.ref_frame, ref_slot, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    SOSVO_HIP(ctx, hipMemcpyAsync(b.cur_frame, cur_slot, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
  }
  SOSVO_LAUNCH(ctx, seq_setup_kernel, dim3(1), dim3(64), 0, ctx->stream, n_pairs > kSeqArgPairs ? 0 : n_pairs, sp, foci, b.ref_frame,
               b.cur_frame, b.cam_off, b.cam_rot);
  SOSVO_LAUNCH_CHECK(ctx);
  const FrameStore store{b.m_top, b.m_bot, b.d_top, b.d_bot, b.X, b.b_top, b.b_bot, b.M};
  const TrackTmp tt{b.ref_frame, b.cur_frame, b.o_top, b.o_bot, b.cam, b.cq, b.ct, b.cn, b.cn_top, b.idx, b.n_inl, b.info,
                    b.lm_iters, b.k_top, b.k_bot, b.mask, b.f, b.p, b.T_ransac, b.T, b.cam_off, b.cam_rot, b.lm_cost};
  return run_tracking(ctx, rig, cfg, n_pairs, seed, store, tt, results);
}

int32_t sosvo_sequence_copy_slot(sosvo_ctx* ctx, const sosvo_batch_cfg* cfg, int32_t window, int32_t slots, int32_t src_slot,
                                 int32_t dst_slot, void* workspace, size_t workspace_bytes) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, cfg && workspace, "null pointer");
  SOSVO_REQUIRE(ctx, seq_shape_ok(cfg, window, slots), "bad sequence configuration (window, slots, capacities)");
  SOSVO_REQUIRE(ctx, src_slot >= 0 && src_slot < slots && dst_slot >= 0 && dst_slot < slots, "slot out of range");
  SOSVO_REQUIRE(ctx, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  const SeqBuffers b = carve_seq(*cfg, window, slots, workspace);
  SOSVO_REQUIRE(ctx, workspace_bytes >= b.bytes, "workspace too small (see sosvo_sequence_workspace)");
  if (src_slot == dst_slot) return SOSVO_OK;
  SOSVO_LAUNCH(ctx, seq_copy_slot_kernel, dim3(1), dim3(1024), 0, ctx->stream, src_slot, dst_slot, cfg->frame_cap, b.m_top,
               b.m_bot, reinterpret_cast<uint4*>(b.d_top), reinterpret_cast<uint4*>(b.d_bot), b.X, b.b_top, b.b_bot, b.M);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_sequence_frame_counts(sosvo_ctx* ctx, const sosvo_batch_cfg* cfg, int32_t window, int32_t slots, int32_t first_slot,
                                    int32_t n, void* workspace, size_t workspace_bytes, int32_t* counts_host) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, cfg && workspace && counts_host, "null pointer");
  SOSVO_REQUIRE(ctx, seq_shape_ok(cfg, window, slots), "bad sequence configuration (window, slots, capacities)");
  SOSVO_REQUIRE(ctx, first_slot >= 0 && n >= 0 && first_slot + n <= slots, "slots out of range");
  const SeqBuffers b = carve_seq(*cfg, window, slots, workspace);
  SOSVO_REQUIRE(ctx, workspace_bytes >= b.bytes, "workspace too small (see sosvo_sequence_workspace)");
  if (n == 0) return SOSVO_OK;
  SOSVO_HIP(ctx, hipMemcpyAsync(counts_host, b.M + first_slot, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
  SOSVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SOSVO_OK;
}

size_t sosvo_rgbd_pair_batch_workspace(const sosvo_rgbd_batch_cfg* cfg) {
  if (!cfg || cfg->n_pairs <= 0 || cfg->rows <= 0 || cfg->cols <= 0 || cfg->kp_cap <= 0 || cfg->frame_cap <= 0) return 0;
  return carve_rgbd(*cfg, nullptr).bytes;
}

// RGBDFrame.establish_keypoints for F frames (pose_est_tools.py:600-623): [median,] gray, whole-image GFT (one mask), ORB
// descriptors, depth back-projection + range filter + bearings -> F frame records at the store pointers.
struct RgbdFrontTmp {
  uint8_t *gray, *desc;
  float* kp;
  int32_t *n, *status;
};
struct RgbdStore {
  float* m;
  uint8_t* d;
  double *X, *b;
  int32_t* M;
};
static int32_t run_rgbd_front_end(sosvo_ctx* ctx, const sosvo_rgbd_cam* cam, const sosvo_rgbd_batch_cfg* cfg, int F,
                                  const uint8_t* bgr, const float* depth, const uint32_t* mask_bits, const int8_t* pattern,
                                  const RgbdFrontTmp& t, const RgbdStore& o) {
  const int cap = cfg->kp_cap, Fc = cfg->frame_cap;
  int32_t rc;
  STAGE(sosvo_median_gray(ctx, bgr, F, cfg->rows, cfg->cols, cfg->median_ksize, t.gray));
  STAGE(sosvo_detect_gft(ctx, t.gray, mask_bits, F, F, cfg->rows, cfg->cols, 1, cfg->quality, cfg->min_distance,
                         cfg->max_corners, cap, t.kp, t.n, t.status));
  STAGE(sosvo_describe_orb(ctx, t.gray, F, cfg->rows, cfg->cols, 1, cap, t.kp, t.n, cfg->cos_a, cfg->sin_a, pattern, cfg->edge,
                           t.desc));
  STAGE(sosvo_rgbd_assemble(ctx, cam, t.kp, t.desc, t.n, depth, F, cfg->rows, cfg->cols, cap, Fc, o.m, o.d, o.X, o.b, o.M));
  return SOSVO_OK;
}

// TrackerRGBDSE3.track_frame (pose_est_tools.py:896-954) for B (reference slot, current slot) pairs of a frame store:
// query = current frame, train = reference frame; central RANSAC (problem i samples with seed + i) + LM.
struct RgbdTrackTmp {
  int32_t *ref_frame, *cur_frame, *order, *cq, *ct, *cn, *idx, *n_inl, *info, *lm_iters;
  uint32_t* keys;
  uint8_t* mask;
  double *f, *p, *T_ransac, *T, *lm_cost;
};
static int32_t run_rgbd_tracking(sosvo_ctx* ctx, const sosvo_rgbd_batch_cfg* cfg, int B, uint64_t seed, const RgbdStore& s,
                                 const RgbdTrackTmp& b, double* results) {
  const int Fc = cfg->frame_cap;
  int32_t rc;
  STAGE(sosvo_match_hamming(ctx, s.d, s.d, s.M, s.M, b.cur_frame, b.ref_frame, B, Fc, Fc, 1, b.keys));
  STAGE(sosvo_sort_matches(ctx, b.keys, s.M, b.cur_frame, B, Fc, b.order));
  STAGE(sosvo_f2f_assemble_central(ctx, cfg->pct_good_matches, cfg->f2f_max_hdiff, s.m, s.X, s.b, s.M, Fc, b.ref_frame,
                                   b.cur_frame, b.keys, b.order, B, Fc, b.f, b.p, b.cq, b.ct, b.cn));
  STAGE(sosvo_ransac_abs_pose(ctx, b.f, b.p, nullptr, nullptr, nullptr, 1,
                              cfg->flags & (SOSVO_FLAG_EPNP | SOSVO_FLAG_GP3P | SOSVO_FLAG_TWOPT), b.cn, B, Fc,
                              cfg->ransac_threshold, cfg->ransac_max_iter, cfg->ransac_adaptive, seed, b.T_ransac, b.mask, b.idx,
                              b.n_inl, b.info, nullptr));
  SOSVO_HIP(ctx, hipMemcpyAsync(b.T, b.T_ransac, sizeof(double) * 12 * B, hipMemcpyDeviceToDevice, ctx->stream));
  STAGE(sosvo_refine_abs_pose(ctx, b.f, b.p, nullptr, nullptr, nullptr, 1, b.cn, B, Fc, b.idx, b.n_inl, cfg->lm_max_iter, b.T,
                              b.lm_cost, b.lm_iters));
  SOSVO_LAUNCH(ctx, batch_results_kernel, dim3(cdiv(B, 256)), dim3(256), 0, ctx->stream, B, b.T, b.n_inl, b.cn, b.info,
               results);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_rgbd_pair_batch(sosvo_ctx* ctx, const sosvo_rgbd_cam* cam, const sosvo_rgbd_batch_cfg* cfg,
                              const uint8_t* bgr, const float* depth, const uint32_t* mask_bits, const int8_t* pattern,
                              void* workspace, size_t workspace_bytes, double* results) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, cam && cfg && bgr && depth && mask_bits && pattern && workspace && results, "null pointer");
  SOSVO_REQUIRE(ctx, cfg->n_pairs > 0 && cfg->n_pairs <= 8192, "n_pairs out of range (1..8192)");
  SOSVO_REQUIRE(ctx, cfg->kp_cap > 0 && cfg->kp_cap <= 4096 && cfg->frame_cap > 0 && cfg->frame_cap <= 16384,
                "kp_cap / frame_cap out of range");
  SOSVO_REQUIRE(ctx, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  const RgbdBuffers b = carve_rgbd(*cfg, workspace);
  SOSVO_REQUIRE(ctx, workspace_bytes >= b.bytes, "workspace too small (see sosvo_rgbd_pair_batch_workspace)");
  const int B = cfg->n_pairs, F = 2 * B;
  int32_t rc;
  SOSVO_LAUNCH(ctx, pair_index_kernel, dim3(cdiv(B, 256)), dim3(256), 0, ctx->stream, B, b.ref_frame, b.cur_frame);
  SOSVO_LAUNCH_CHECK(ctx);
  const RgbdFrontTmp fe{b.gray, b.desc, b.kp, b.n, b.status};
  const RgbdStore store{b.m, b.d, b.X, b.b, b.M};
  STAGE(run_rgbd_front_end(ctx, cam, cfg, F, bgr, depth, mask_bits, pattern, fe, store));
  const RgbdTrackTmp tt{b.ref_frame, b.cur_frame, b.order, b.cq, b.ct, b.cn, b.idx, b.n_inl, b.info, b.lm_iters, b.keys, b.mask,
                        b.f, b.p, b.T_ransac, b.T, b.lm_cost};
  return run_rgbd_tracking(ctx, cfg, B, cfg->seed, store, tt, results);
}

// ---- RGB-D sequence mode -------------------------------------------------------------------------------------
static bool rgbd_seq_shape_ok(const sosvo_rgbd_batch_cfg* cfg, int32_t window, int32_t slots) {
  return cfg && cfg->n_pairs > 0 && cfg->n_pairs <= 8192 && cfg->rows > 0 && cfg->cols > 0 && cfg->kp_cap > 0 && cfg->kp_cap <= 4096 &&
         cfg->frame_cap > 0 && cfg->frame_cap <= 16384 && window >= 1 && window <= 16384 && slots >= window && slots <= 65535;
}

size_t sosvo_rgbd_sequence_workspace(const sosvo_rgbd_batch_cfg* cfg, int32_t window, int32_t slots) {
  if (!rgbd_seq_shape_ok(cfg, window, slots)) return 0;
  return carve_rgbd_seq(*cfg, window, slots, nullptr).bytes;
}

int32_t sosvo_rgbd_sequence_front_end(sosvo_ctx* ctx, const sosvo_rgbd_cam* cam, const sosvo_rgbd_batch_cfg* cfg, int32_t window,
                                      int32_t slots, const uint8_t* bgr, const float* depth, int32_t n_frames, int32_t first_slot,
                                      const uint32_t* mask_bits, const int8_t* pattern, void* workspace, size_t workspace_bytes) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, cam && cfg && bgr && depth && mask_bits && pattern && workspace, "null pointer");
  SOSVO_REQUIRE(ctx, rgbd_seq_shape_ok(cfg, window, slots), "bad sequence configuration (window, slots, capacities)");
  SOSVO_REQUIRE(ctx, n_frames >= 0 && n_frames <= window && first_slot >= 0 && first_slot + n_frames <= slots,
                "frames do not fit the window / the store");
  SOSVO_REQUIRE(ctx, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  const RgbdSeqBuffers b = carve_rgbd_seq(*cfg, window, slots, workspace);
  SOSVO_REQUIRE(ctx, workspace_bytes >= b.bytes, "workspace too small (see sosvo_rgbd_sequence_workspace)");
  if (n_frames == 0) return SOSVO_OK;
  const size_t Fc = cfg->frame_cap, o = (size_t)first_slot * Fc;
  const RgbdFrontTmp fe{b.gray, b.desc, b.kp, b.n, b.status};
  const RgbdStore out{b.m + 2 * o, b.d + 32 * o, b.X + 3 * o, b.b + 3 * o, b.M + first_slot};
  return run_rgbd_front_end(ctx, cam, cfg, n_frames, bgr, depth, mask_bits, pattern, fe, out);
}

int32_t sosvo_rgbd_sequence_track(sosvo_ctx* ctx, const sosvo_rgbd_batch_cfg* cfg, int32_t window, int32_t slots,
                                  const int32_t* ref_slot, const int32_t* cur_slot, int32_t n_pairs, uint64_t seed, void* workspace,
                                  size_t workspace_bytes, double* results) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, cfg && ref_slot && cur_slot && workspace && results, "null pointer");
  SOSVO_REQUIRE(ctx, rgbd_seq_shape_ok(cfg, window, slots), "bad sequence configuration (window, slots, capacities)");
  SOSVO_REQUIRE(ctx, n_pairs >= 0 && n_pairs <= cfg->n_pairs, "more slot pairs than cfg->n_pairs");
  SOSVO_REQUIRE(ctx, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  const RgbdSeqBuffers b = carve_rgbd_seq(*cfg, window, slots, workspace);
  SOSVO_REQUIRE(ctx, workspace_bytes >= b.bytes, "workspace too small (see sosvo_rgbd_sequence_workspace)");
  if (n_pairs == 0) return SOSVO_OK;
  for (int i = 0; i < n_pairs; ++i)
    SOSVO_REQUIRE(ctx, ref_slot[i] >= 0 && ref_slot[i] < slots && cur_slot[i] >= 0 && cur_slot[i] < slots, "slot out of range");
  SlotPairs sp;
  for (int i = 0; i < kSeqArgPairs; ++i) {
    sp.ref[i] = i < n_pairs ? ref_slot[i] : 0;
    sp.cur[i] = i < n_pairs ? cur_slot[i] : 0;
  }
  if (n_pairs > kSeqArgPairs) {
    SOSVO_HIP(ctx, hipMemcpyAsync(b.ref_frame, ref_slot, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    SOSVO_HIP(ctx, hipMemcpyAsync(b.cur_frame, cur_slot, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
  } else {
    SOSVO_LAUNCH(ctx, rgbd_seq_setup_kernel, dim3(1), dim3(64), 0, ctx->stream, n_pairs, sp, b.ref_frame, b.cur_frame);
    SOSVO_LAUNCH_CHECK(ctx);
  }
  const RgbdStore store{b.m, b.d, b.X, b.b, b.M};
  const RgbdTrackTmp tt{b.ref_frame, b.cur_frame, b.order, b.cq, b.ct, b.cn, b.idx, b.n_inl, b.info, b.lm_iters, b.keys, b.mask,
                        b.f, b.p, b.T_ransac, b.T, b.lm_cost};
  return run_rgbd_tracking(ctx, cfg, n_pairs, seed, store, tt, results);
}

int32_t sosvo_rgbd_sequence_copy_slot(sosvo_ctx* ctx, const sosvo_rgbd_batch_cfg* cfg, int32_t window, int32_t slots,
                                      int32_t src_slot, int32_t dst_slot, void* workspace, size_t workspace_bytes) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, cfg && workspace, "null pointer");
  SOSVO_REQUIRE(ctx, rgbd_seq_shape_ok(cfg, window, slots), "bad sequence configuration (window, slots, capacities)");
  SOSVO_REQUIRE(ctx, src_slot >= 0 && src_slot < slots && dst_slot >= 0 && dst_slot < slots, "slot out of range");
  SOSVO_REQUIRE(ctx, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  const RgbdSeqBuffers b = carve_rgbd_seq(*cfg, window, slots, workspace);
  SOSVO_REQUIRE(ctx, workspace_bytes >= b.bytes, "workspace too small (see sosvo_rgbd_sequence_workspace)");
  if (src_slot == dst_slot) return SOSVO_OK;
  SOSVO_LAUNCH(ctx, rgbd_seq_copy_slot_kernel, dim3(1), dim3(1024), 0, ctx->stream, src_slot, dst_slot, cfg->frame_cap, b.m,
               reinterpret_cast<uint4*>(b.d), b.X, b.b, b.M);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_rgbd_sequence_frame_counts(sosvo_ctx* ctx, const sosvo_rgbd_batch_cfg* cfg, int32_t window, int32_t slots,
                                         int32_t first_slot, int32_t n, void* workspace, size_t workspace_bytes,
                                         int32_t* counts_host) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, cfg && workspace && counts_host, "null pointer");
  SOSVO_REQUIRE(ctx, rgbd_seq_shape_ok(cfg, window, slots), "bad sequence configuration (window, slots, capacities)");
  SOSVO_REQUIRE(ctx, first_slot >= 0 && n >= 0 && first_slot + n <= slots, "slots out of range");
  const RgbdSeqBuffers b = carve_rgbd_seq(*cfg, window, slots, workspace);
  SOSVO_REQUIRE(ctx, workspace_bytes >= b.bytes, "workspace too small (see sosvo_rgbd_sequence_workspace)");
  if (n == 0) return SOSVO_OK;
  SOSVO_HIP(ctx, hipMemcpyAsync(counts_host, b.M + first_slot, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
  SOSVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SOSVO_OK;
}

}  // extern "C"
