// K5 -- ORB keypoint detection per azimuthal mask, and K6' -- ORB descriptors for oriented multi-level keypoints.
//
//   cv2.ORB_create(nfeatures=N).detect(pano, mask)   omnistereo/camera_models.py:1640, :1755; pose_est_tools.py:478, :547
//   .compute(pano, keypoints)                        camera_models.py:1765; pose_est_tools.py:553
//
// Batched over images (view-major) x azimuthal masks; problem p = image * nmask + mask.
//   pyramid   8 levels x 1.2, each level bilinear (11-bit fixed point) from the previous one; all levels of one
//             image live back to back in one workspace row (level offsets in a by-value struct);
//   fast      FAST-9/16 score for every pyramid pixel in one launch (16 ring loads per lane from L1/L2);
//   select    one workgroup per problem walks the levels: 3x3 NMS + 31-px border + mask -> candidates in LDS,
//             keep the best 2 n_l by FAST score through a 256-bin LDS histogram (ties kept), Harris response
//             per candidate, rank sort (response desc, then y, x), keep n_l (+ ties), intensity-centroid angle
//             with one wave per keypoint (lanes = patch rows, shuffle reduction), fastAtan2 polynomial in
//             float32 with a pinned operation order;
//   describe  border rule on level-0 coordinates + stable compaction, then one wave per keypoint: the rotated
//             offsets are evaluated per lane in float32, four 64-bit ballots are the 32 descriptor bytes.
// Integer work except the Harris/angle float32 arithmetic, whose operation order is pinned -> bit-exact against
// the oracle (cos/sin of the angle are double-precision library calls rounded to float32 on both sides).
#include "common.h"
#include "trig_core.h"

namespace {

constexpr int kThreads = 256;
constexpr int kLevels = 8;
constexpr int kEdge = 31;
constexpr int kHalfPatch = 15;
constexpr int kFastThr = 20;
constexpr int kCandMax = 2048;  // candidates per (problem, level) held in LDS

struct Pyr {
  int h[kLevels], w[kLevels];
  long long off[kLevels];
  float scale[kLevels];
  int quota[kLevels];
  int nlev;          // levels that exist (h, w >= 1)
  long long total;   // pixels of one image's pyramid
};

__device__ __forceinline__ int refl101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

// ---- pyramid -------------------------------------------------------------------------------------------
struct ResizeTap {
  int i0, i1, w0, w1;
};

__device__ __forceinline__ ResizeTap resize_tap(int d, double scale, int n0) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int i = (int)floorf(f);
  f -= (float)i;
  if (i < 0) {
    i = 0;
    f = 0.f;
  }
  if (i >= n0 - 1) {
    i = n0 - 1;
    f = 0.f;
  }
  ResizeTap t;
  t.i0 = i;
  t.i1 = i + 1 < n0 ? i + 1 : n0 - 1;
  t.w1 = __float2int_rn(f * 2048.f);
  t.w0 = 2048 - t.w1;
  return t;
}

__global__ __launch_bounds__(kThreads) void copy_level0_kernel(const uint8_t* __restrict__ gray, int npix,
                                                               long long total, uint8_t* __restrict__ pyr) {
  SOSVO_STREAMING_PRIO();
  const int i = blockIdx.x * kThreads + threadIdx.x, img = blockIdx.y;
  if (i < npix) pyr[(size_t)img * total + i] = gray[(size_t)img * npix + i];
}

__global__ __launch_bounds__(kThreads) void resize_level_kernel(uint8_t* __restrict__ pyr, long long total,
                                                                long long off0, int h0, int w0, long long off1, int h1,
                                                                int w1) {
  SOSVO_STREAMING_PRIO();
  const int i = blockIdx.x * kThreads + threadIdx.x, img = blockIdx.y;
  if (i >= h1 * w1) return;
  const int dy = i / w1, dx = i - dy * w1;
  const ResizeTap ty = resize_tap(dy, (double)h0 / h1, h0), tx = resize_tap(dx, (double)w0 / w1, w0);
  const uint8_t* s = pyr + (size_t)img * total + off0;
  const long long top = (long long)tx.w0 * s[(size_t)ty.i0 * w0 + tx.i0] + (long long)tx.w1 * s[(size_t)ty.i0 * w0 + tx.i1];
  const long long bot = (long long)tx.w0 * s[(size_t)ty.i1 * w0 + tx.i0] + (long long)tx.w1 * s[(size_t)ty.i1 * w0 + tx.i1];
  pyr[(size_t)img * total + off1 + i] = (uint8_t)((ty.w0 * top + ty.w1 * bot + (1 << 21)) >> 22);
}

// mask level l from level l-1: every mask bit is resized as a 0/255 image and kept where the result is > 254
__global__ __launch_bounds__(kThreads) void mask_level_kernel(uint32_t* __restrict__ mp, long long total, long long off0,
                                                              int h0, int w0, long long off1, int h1, int w1, int nmask) {
  const int i = blockIdx.x * kThreads + threadIdx.x, set = blockIdx.y;
  if (i >= h1 * w1) return;
  const int dy = i / w1, dx = i - dy * w1;
  const ResizeTap ty = resize_tap(dy, (double)h0 / h1, h0), tx = resize_tap(dx, (double)w0 / w1, w0);
  const uint32_t* s = mp + (size_t)set * total + off0;
  const uint32_t b00 = s[(size_t)ty.i0 * w0 + tx.i0], b01 = s[(size_t)ty.i0 * w0 + tx.i1];
  const uint32_t b10 = s[(size_t)ty.i1 * w0 + tx.i0], b11 = s[(size_t)ty.i1 * w0 + tx.i1];
  uint32_t out = 0;
  for (int m = 0; m < nmask; ++m) {
    const long long top = (long long)tx.w0 * (((b00 >> m) & 1u) * 255) + (long long)tx.w1 * (((b01 >> m) & 1u) * 255);
    const long long bot = (long long)tx.w0 * (((b10 >> m) & 1u) * 255) + (long long)tx.w1 * (((b11 >> m) & 1u) * 255);
    const int v = (int)((ty.w0 * top + ty.w1 * bot + (1 << 21)) >> 22);
    if (v > 254) out |= 1u << m;
  }
  mp[(size_t)set * total + off1 + i] = out;
}

// ---- FAST-9/16 score map over the whole pyramid ------------------------------------------------------------
// d[k] = circle pixel k minus the centre pixel -> corner score: the largest threshold for which the pixel is still a FAST-9
// corner, minus 1 (0: not a corner at `thr`).  Max over the 16 arcs of 9 consecutive circle pixels of min(d) (brighter) and
// of min(-d) = -max(d) (darker), by doubling: windows of 2, 4, 8, then 9 -- 4 min + 4 max per start instead of 8 + 8.
__device__ __forceinline__ int fast_score_from_diffs(const int (&d)[16], int thr) {
  int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo2[k] = min(d[k], d[(k + 1) & 15]);
    hi2[k] = max(d[k], d[(k + 1) & 15]);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
    hi4[k] = max(hi2[k], hi2[(k + 2) & 15]);
  }
  int best = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int mn_b = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);   // min of d over k .. k + 8
    const int mx = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);     // max of d over k .. k + 8
    best = max(best, max(mn_b, -mx));
  }
  return best > thr ? best - 1 : 0;
}

// FAST-9/16 score map of nimg images that lie img_stride bytes apart (a dense batch, or one pyramid level): a wave owns a
// 64-column strip (58 output columns, 3 halo columns each side) and walks down the rows with the last seven rows of its
// column in a register ring; the 16 circle pixels of a row come from the ring slots of neighbouring lanes (one
// cross-lane read each) instead of 16 scattered byte loads per pixel.  3-pixel image border (and everything when
// `enabled` is 0: a pyramid level that cannot hold a keypoint) scores 0.
constexpr int kFsHalo = 3, kFsStripW = 64 - 2 * kFsHalo;
__global__ __launch_bounds__(kThreads) void fast_score_rolling_kernel(const uint8_t* __restrict__ in, long long img_stride,
                                                                      int nimg, int rows, int cols, int strips, int thr,
                                                                      int enabled, uint8_t* __restrict__ out) {
  SOSVO_STREAMING_PRIO();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * kThreads + threadIdx.x) >> 6));
  if (wave >= nimg * strips) return;  // wave-uniform
  const int img = wave / strips, strip = wave - img * strips;
  const int xb = strip * kFsStripW - kFsHalo;
  const int xc = xb + lane;
  const int xs = min(max(xc, 0), cols - 1);
  const bool out_lane = lane >= kFsHalo && lane < 64 - kFsHalo && xc < cols;
  const bool interior_x = xc >= 3 && xc < cols - 3;
  const uint8_t* g = in + (size_t)img * img_stride;
  uint8_t* o = out + (size_t)img * img_stride;
  int v[7] = {0, 0, 0, 0, 0, 0, 0};  // rows t-6 .. t of this lane's column, slot = row mod 7 (compile-time after unrolling)
  const int t_last = rows - 1 + 3;
  int c_next = (int)g[(uint32_t)xs];
  auto step = [&](auto phase_tag, const int t) __attribute__((always_inline)) {
    constexpr int P = decltype(phase_tag)::value;
    if (t > t_last) return;  // uniform
    v[P] = c_next;           // row t (rows past the image repeat the last one: they are never used as circle pixels)
    c_next = (int)g[(uint32_t)(min(t + 1, rows - 1) * cols) + (uint32_t)xs];
    const int y = t - 3;
    if (y < 0) return;  // uniform
    int s = 0;
    if (enabled && y >= 3 && y < rows - 3) {  // uniform
      // row y + dy lives in slot (P + 4 + dy) mod 7; circle pixel k = (rx[k], ry[k]) as in OpenCV's table
      const int c = v[(P + 4) % 7];
      int d[16];
      d[0] = v[(P + 7) % 7] - c;                                   // ( 0,  3)
      d[1] = __shfl(v[(P + 7) % 7], lane + 1) - c;                 // ( 1,  3)
      d[2] = __shfl(v[(P + 6) % 7], lane + 2) - c;                 // ( 2,  2)
      d[3] = __shfl(v[(P + 5) % 7], lane + 3) - c;                 // ( 3,  1)
      d[4] = __shfl(v[(P + 4) % 7], lane + 3) - c;                 // ( 3,  0)
      d[5] = __shfl(v[(P + 3) % 7], lane + 3) - c;                 // ( 3, -1)
      d[6] = __shfl(v[(P + 2) % 7], lane + 2) - c;                 // ( 2, -2)
      d[7] = __shfl(v[(P + 1) % 7], lane + 1) - c;                 // ( 1, -3)
      d[8] = v[(P + 1) % 7] - c;                                   // ( 0, -3)
      d[9] = __shfl(v[(P + 1) % 7], lane - 1) - c;                 // (-1, -3)
      d[10] = __shfl(v[(P + 2) % 7], lane - 2) - c;                // (-2, -2)
      d[11] = __shfl(v[(P + 3) % 7], lane - 3) - c;                // (-3, -1)
      d[12] = __shfl(v[(P + 4) % 7], lane - 3) - c;                // (-3,  0)
      d[13] = __shfl(v[(P + 5) % 7], lane - 3) - c;                // (-3,  1)
      d[14] = __shfl(v[(P + 6) % 7], lane - 2) - c;                // (-2,  2)
      d[15] = __shfl(v[(P + 7) % 7], lane - 1) - c;                // (-1,  3)
      s = interior_x ? fast_score_from_diffs(d, thr) : 0;
    }
    if (out_lane) o[(uint32_t)(y * cols) + (uint32_t)xc] = (uint8_t)s;
  };
  for (int t = 0; t <= t_last; t += 7) {
    step(std::integral_constant<int, 0>{}, t);
    step(std::integral_constant<int, 1>{}, t + 1);
    step(std::integral_constant<int, 2>{}, t + 2);
    step(std::integral_constant<int, 3>{}, t + 3);
    step(std::integral_constant<int, 4>{}, t + 4);
    step(std::integral_constant<int, 5>{}, t + 5);
    step(std::integral_constant<int, 6>{}, t + 6);
  }
}

static int32_t launch_fast_score(sosvo_ctx* ctx, const uint8_t* in, long long img_stride, int nimg, int rows, int cols, int thr,
                                 int enabled, uint8_t* out) {
  const int strips = cdiv(cols, kFsStripW);
  SOSVO_LAUNCH(ctx, fast_score_rolling_kernel, dim3(cdiv(nimg * strips, kThreads / 64)), dim3(kThreads), 0, ctx->stream, in,
               img_stride, nimg, rows, cols, strips, thr, enabled, out);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

// ---- FAST as a detector of its own (feature_detection_method "FAST") ------------------------------------------
// cv2.FastFeatureDetector_create() + setNonmaxSuppression(True) + detect(image, mask) (omnistereo/camera_models.py:
// 1664-1666, :1755; pose_est_tools.py:506-508): FAST-9/16 with threshold 10 on the whole image (3-px border), 3x3
// non-maximum suppression on the corner score (strictly greater than the 8 neighbours), keypoints kept where the
// mask is set, in raster order.  Three launches: score map, NMS flags (one u64 per 64 pixels of a row), and one
// WAVE per (image, mask) that walks the rows and compacts the flagged, masked pixels in order.
__global__ __launch_bounds__(kThreads) void fast_nms_flags_kernel(const uint8_t* __restrict__ score, int rows, int cols,
                                                                  int words, unsigned long long* __restrict__ flags) {
  SOSVO_STREAMING_PRIO();
  // one wave per (row, 64-pixel word): grid.x covers rows * words waves, grid.y = image
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * kThreads + threadIdx.x) >> 6));
  if (wv >= rows * words) return;
  const int y = wv / words, w = wv - y * words, x = 64 * w + lane, img = blockIdx.y;
  const uint8_t* sc = score + (size_t)img * rows * cols;
  bool keep = false;
  if (x < cols) {
    const int s = sc[(size_t)y * cols + x];
    if (s > 0) {  // score > 0 implies the 3-px border, so all 8 neighbours exist
      keep = true;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
          if ((dy || dx) && sc[(size_t)(y + dy) * cols + x + dx] >= s) keep = false;
    }
  }
  const unsigned long long bal = __ballot(keep);
  if (lane == 0) flags[((size_t)img * rows + y) * words + w] = bal;
}

__global__ __launch_bounds__(64) void fast_collect_kernel(const unsigned long long* __restrict__ flags,
                                                          const uint32_t* __restrict__ mask_bits, int images_per_maskset,
                                                          int nmask, int rows, int cols, int words, int cap,
                                                          float* __restrict__ kp, int32_t* __restrict__ n_out,
                                                          int32_t* __restrict__ status) {
  SOSVO_STREAMING_PRIO();
  const int lane = threadIdx.x, p = blockIdx.x;
  const int img = p / nmask, m = p - img * nmask;
  const uint32_t* mb = mask_bits + (size_t)(img / images_per_maskset) * rows * cols;
  const unsigned long long* fl = flags + (size_t)img * rows * words;
  int total = 0;  // wave-uniform
  for (int y = 3; y < rows - 3; ++y) {
    for (int w0 = 0; w0 < words; w0 += 64) {
      const int w = w0 + lane;
      unsigned long long f = w < words ? fl[(size_t)y * words + w] : 0ULL;
      // keep the flagged pixels whose mask bit is set (flags are sparse: a handful of bits per word)
      unsigned long long kept = 0ULL;
      while (f) {
        const int b = __ffsll((long long)f) - 1;
        f &= f - 1ULL;
        if ((mb[(size_t)y * cols + 64 * w + b] >> m) & 1u) kept |= 1ULL << b;
      }
      int c = __popcll(kept), pre = c;  // inclusive prefix over the lanes (= words, in raster order)
#pragma unroll
      for (int s = 1; s < 64; s <<= 1) {
        const int o = __shfl_up(pre, s);
        if (lane >= s) pre += o;
      }
      int pos = total + pre - c;
      while (kept) {
        const int b = __ffsll((long long)kept) - 1;
        kept &= kept - 1ULL;
        if (pos < cap) {
          kp[((size_t)p * cap + pos) * 2] = (float)(64 * w + b);
          kp[((size_t)p * cap + pos) * 2 + 1] = (float)y;
        }
        pos++;
      }
      total += __shfl(pre, 63);
    }
  }
  if (lane == 0) {
    n_out[p] = min(total, cap);
    if (status) status[p] = total > cap ? 1 : 0;
  }
}

// ---- selection ------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float scale = (float)(180.0 / 3.14159265358979323846);
  const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
  const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + 2.220446049250313e-16f);
    c2 = c * c;
    a = ((((p7 * c2) + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + 2.220446049250313e-16f);
    c2 = c * c;
    a = 90.f - ((((p7 * c2) + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

__device__ __forceinline__ float harris_response(const uint8_t* __restrict__ im, int h, int w, int cx, int cy) {
  int a = 0, b = 0, c = 0;
  for (int dy = -3; dy <= 3; ++dy)
    for (int dx = -3; dx <= 3; ++dx) {
      const int y = cy + dy, x = cx + dx;
#define PX(yy, xx) ((int)im[(size_t)refl101((yy), h) * w + refl101((xx), w)])
      const int Ix = (PX(y, x + 1) - PX(y, x - 1)) * 2 + (PX(y - 1, x + 1) - PX(y - 1, x - 1)) + (PX(y + 1, x + 1) - PX(y + 1, x - 1));
      const int Iy = (PX(y + 1, x) - PX(y - 1, x)) * 2 + (PX(y + 1, x - 1) - PX(y - 1, x - 1)) + (PX(y + 1, x + 1) - PX(y - 1, x + 1));
#undef PX
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  const float scale = 1.f / (4 * 7 * 255.f);
  const float s4 = (scale * scale) * (scale * scale);
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  return (((fa * fb) - (fc * fc)) - ((0.04f * (fa + fb)) * (fa + fb))) * s4;
}

// Bounding box of every mask on every pyramid level of every mask set (depends on the masks only; cheap enough to
// redo per call): bbox[((set * kLevels + l) * 32 + m) * 4 + {0,1,2,3}] = {~xmin, ~ymin, xmax + 1, ymax + 1} grown by
// atomicMax from zero (field 2 == 0: the mask is empty on that level).  One workgroup per (pyramid row, set).
__global__ __launch_bounds__(kThreads) void orb_mask_bbox_kernel(const uint32_t* __restrict__ mask_pyr, Pyr P, int nmask,
                                                                 uint32_t* __restrict__ bbox) {
  __shared__ uint32_t sb[2][32];
  const int tid = threadIdx.x, set = blockIdx.y;
  int row = blockIdx.x, l = 0;
  while (l < P.nlev && row >= P.h[l]) {
    row -= P.h[l];
    ++l;
  }
  if (l >= P.nlev) return;  // uniform
  if (tid < 64) (&sb[0][0])[tid] = 0u;
  __syncthreads();
  const int w = P.w[l];
  const uint32_t* mk = mask_pyr + (size_t)set * P.total + P.off[l] + (size_t)row * w;
  const uint32_t mask_all = nmask < 32 ? (1u << nmask) - 1u : 0xFFFFFFFFu;
  for (int x = tid; x < w; x += kThreads) {
    uint32_t b = mk[x] & mask_all;
    while (b) {
      const int m = __ffs(b) - 1;
      b &= b - 1;
      atomicMax(&sb[0][m], 0xFFFFFFFFu - (uint32_t)x);
      atomicMax(&sb[1][m], (uint32_t)x + 1u);
    }
  }
  __syncthreads();
  if (tid < nmask && sb[1][tid]) {
    uint32_t* o = bbox + (((size_t)set * kLevels + l) * 32 + tid) * 4;
    atomicMax(&o[0], sb[0][tid]);
    atomicMax(&o[1], 0xFFFFFFFFu - (uint32_t)row);
    atomicMax(&o[2], sb[1][tid]);
    atomicMax(&o[3], (uint32_t)row + 1u);
  }
}

__global__ __launch_bounds__(kThreads) void orb_select_kernel(const uint8_t* __restrict__ pyr,
                                                              const uint8_t* __restrict__ score,
                                                              const uint32_t* __restrict__ mask_pyr,
                                                              const uint32_t* __restrict__ bbox, Pyr P,
                                                              int images_per_maskset, int nmask, int cap,
                                                              float* __restrict__ kp4, float* __restrict__ resp_out,
                                                              int32_t* __restrict__ n_out) {
  SOSVO_LATENCY_BOUND_PRIO();
  __shared__ uint32_t cxy[kCandMax];            // (y << 16) | x
  __shared__ uint8_t cfast[kCandMax];
  __shared__ unsigned long long ckey[kCandMax];  // ordered(harris) << 32 | (0xFFFFFFFF - linear index)
  uint32_t* sxy = cxy;  // the sorted positions reuse the candidates' array (dead once the sort keys exist): 35 KB of LDS per
                        // workgroup instead of 43, i.e. four workgroups per CU instead of three for this latency-bound kernel
  __shared__ float sresp[kCandMax];
  __shared__ int hist[256];
  __shared__ int s_nc, s_thr, s_keep, s_nout;
  __shared__ int wave_off[5];
  __shared__ int s_running;
  const int tid = threadIdx.x, p = blockIdx.x, lane = tid & 63, wid = tid >> 6;
  const int img = p / nmask, m = p - img * nmask;
  const int kumax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  if (tid == 0) s_nout = 0;
  __syncthreads();
  for (int l = 0; l < P.nlev; ++l) {
    const int h = P.h[l], w = P.w[l], quota = P.quota[l];
    if (quota <= 0 || h <= 2 * kEdge || w <= 2 * kEdge) continue;  // uniform
    const uint8_t* im = pyr + (size_t)img * P.total + P.off[l];
    const uint8_t* sc = score + (size_t)img * P.total + P.off[l];
    const uint32_t* mk = mask_pyr + (size_t)(img / images_per_maskset) * P.total + P.off[l];
    if (tid == 0) s_nc = 0;
    for (int k = tid; k < 256; k += kThreads) hist[k] = 0;
    __syncthreads();
    // 1. NMS + border + mask -> candidates (unordered); only the mask's bounding box on this level is scanned (the
    // azimuthal masks are column bands: a twelfth of the level each)
    const uint32_t* bb = bbox + (((size_t)(img / images_per_maskset) * kLevels + l) * 32 + m) * 4;
    const int bx0 = max(kEdge, (int)(0xFFFFFFFFu - bb[0])), by0 = max(kEdge, (int)(0xFFFFFFFFu - bb[1]));
    const int bx1 = min(w - kEdge, (int)bb[2]), by1 = min(h - kEdge, (int)bb[3]);  // exclusive
    const int rw = bb[2] ? max(0, bx1 - bx0) : 0, rh = bb[2] ? max(0, by1 - by0) : 0;
    for (int i = tid; i < rw * rh; i += kThreads) {
      const int y = by0 + i / rw, x = bx0 + i % rw;
      const int s = sc[(size_t)y * w + x];
      if (!s || !((mk[(size_t)y * w + x] >> m) & 1u)) continue;
      bool is_max = true;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
          if ((dy || dx) && sc[(size_t)(y + dy) * w + x + dx] >= s) is_max = false;
      if (is_max) {
        const int slot = atomicAdd(&s_nc, 1);
        if (slot < kCandMax) {
          cxy[slot] = ((uint32_t)y << 16) | (uint32_t)x;
          cfast[slot] = (uint8_t)s;
          atomicAdd(&hist[s], 1);
        }
      }
    }
    __syncthreads();
    const int nc0 = min(s_nc, kCandMax);
    // 2. retainBest(2 * quota) by FAST score, ties kept: threshold from the histogram
    if (tid == 0) {
      int thr = 0;
      if (nc0 > 2 * quota) {
        int acc = 0;
        for (thr = 255; thr >= 0; --thr) {
          acc += hist[thr];
          if (acc >= 2 * quota) break;
        }
      }
      s_thr = thr;
      s_running = 0;
    }
    __syncthreads();
    const int thr = s_thr;
    // 3. Harris response of the survivors -> sort keys (compacted, order irrelevant: keys are unique)
    for (int i0 = 0; i0 < nc0; i0 += kThreads) {
      const int i = i0 + tid;
      const bool keep = i < nc0 && cfast[i] >= thr;
      uint32_t xy = 0;
      float r = 0.f;
      if (keep) {
        xy = cxy[i];
        r = harris_response(im, h, w, (int)(xy & 0xFFFFu), (int)(xy >> 16));
      }
      const int pos = sosvo_block_compact_pos(keep, wave_off, &s_running, tid);
      if (keep) {
        const uint32_t lin = (xy >> 16) * (uint32_t)w + (xy & 0xFFFFu);
        ckey[pos] = ((unsigned long long)sosvo_float_ordered(r) << 32) | (0xFFFFFFFFu - lin);
      }
    }
    __syncthreads();
    const int nc = s_running;
    // 4. rank sort: response descending, then (y, x) ascending
    for (int i = tid; i < nc; i += kThreads) {
      const unsigned long long mine = ckey[i];
      int rank = 0;
      for (int j = 0; j < nc; ++j) rank += ckey[j] > mine;
      const uint32_t lin = 0xFFFFFFFFu - (uint32_t)mine;
      sxy[rank] = ((lin / (uint32_t)w) << 16) | (lin % (uint32_t)w);
      sresp[rank] = sosvo_ordered_float((uint32_t)(mine >> 32));
    }
    __syncthreads();
    // 5. retainBest(quota) by Harris response, ties kept
    if (tid == 0) {
      int keep = nc;
      if (nc > quota) {
        const float amb = sresp[quota - 1];
        keep = quota;
        while (keep < nc && sresp[keep] >= amb) keep++;
      }
      s_keep = keep;
    }
    __syncthreads();
    const int keep = s_keep, base_out = s_nout;
    // 6. orientation: one wave per keypoint, lanes 0..30 take the patch rows v = -15..15
    for (int j = wid; j < keep && base_out + j < cap; j += kThreads / 64) {
      const int cx = (int)(sxy[j] & 0xFFFFu), cy = (int)(sxy[j] >> 16);
      int m10 = 0, m01 = 0;
      if (lane < 2 * kHalfPatch + 1) {
        const int v = lane - kHalfPatch, d = kumax[v < 0 ? -v : v];
        const uint8_t* row = im + (size_t)refl101(cy + v, h) * w;
        int rs = 0;
        for (int u = -d; u <= d; ++u) {
          const int val = row[refl101(cx + u, w)];
          m10 += u * val;
          rs += val;
        }
        m01 = v * rs;
      }
      for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_down(m10, o);
        m01 += __shfl_down(m01, o);
      }
      if (lane == 0) {
        const size_t o = (size_t)p * cap + base_out + j;
        kp4[4 * o + 0] = (float)cx * P.scale[l];
        kp4[4 * o + 1] = (float)cy * P.scale[l];
        kp4[4 * o + 2] = fast_atan2_deg((float)m01, (float)m10);
        kp4[4 * o + 3] = (float)l;
        resp_out[o] = sresp[j];
      }
    }
    __syncthreads();
    if (tid == 0) s_nout = min(cap, base_out + keep);
    __syncthreads();
  }
  if (tid == 0) n_out[p] = s_nout;
}

// ---- descriptors of oriented multi-level keypoints ------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void orb_describe_levels_kernel(const uint8_t* __restrict__ blur, Pyr P, int rows,
                                                                       int cols, int nmask, int cap,
                                                                       float* __restrict__ kp4, int32_t* __restrict__ n_io,
                                                                       const int8_t* __restrict__ pattern,
                                                                       uint8_t* __restrict__ desc,
                                                                       float* __restrict__ kp_xy) {
  SOSVO_LATENCY_BOUND_PRIO();
  extern __shared__ float lds_kp[];  // [cap][4]
  __shared__ int8_t spat[1024];
  __shared__ int wave_off[5];
  __shared__ int s_running;
  const int tid = threadIdx.x, p = blockIdx.x, lane = tid & 63, wid = tid >> 6;
  const int img = p / nmask;
  const int n = min(n_io[p], cap);
  if (tid == 0) s_running = 0;
  for (int i = tid; i < 1024; i += kThreads) spat[i] = pattern[i];
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += kThreads) {
    const int i = i0 + tid;
    float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;
    bool keep = false;
    if (i < n) {
      const float* s = kp4 + ((size_t)p * cap + i) * 4;
      k0 = s[0];
      k1 = s[1];
      k2 = s[2];
      k3 = s[3];
      const int l = (int)k3;
      keep = k0 >= (float)kEdge && k0 < (float)(cols - kEdge) && k1 >= (float)kEdge && k1 < (float)(rows - kEdge) &&
             l >= 0 && l < P.nlev;
    }
    const int pos = sosvo_block_compact_pos(keep, wave_off, &s_running, tid);
    if (keep) {
      lds_kp[4 * pos + 0] = k0;
      lds_kp[4 * pos + 1] = k1;
      lds_kp[4 * pos + 2] = k2;
      lds_kp[4 * pos + 3] = k3;
    }
  }
  __syncthreads();
  const int mkept = s_running;
  for (int i = tid; i < 4 * mkept; i += kThreads) kp4[(size_t)p * cap * 4 + i] = lds_kp[i];
  if (kp_xy)
    for (int i = tid; i < mkept; i += kThreads) {
      kp_xy[((size_t)p * cap + i) * 2] = lds_kp[4 * i];
      kp_xy[((size_t)p * cap + i) * 2 + 1] = lds_kp[4 * i + 1];
    }
  if (tid == 0) n_io[p] = mkept;
  for (int j = wid; j < mkept; j += kThreads / 64) {
    const int l = (int)lds_kp[4 * j + 3];
    const int hh = P.h[l], ww = P.w[l];
    const float inv = 1.f / P.scale[l];
    float angle = lds_kp[4 * j + 2];
    angle *= (float)(3.14159265358979323846 / 180.0);
    double sd, cd;  // trig_core.h's sincos: the same bits as the oracle's, then one rounding to float on both sides
    sv_sincos((double)angle, &sd, &cd);
    const float ca = (float)cd, sa = (float)sd;
    const int cx = __float2int_rn(lds_kp[4 * j] * inv), cy = __float2int_rn(lds_kp[4 * j + 1] * inv);
    const uint8_t* im = blur + (size_t)img * P.total + P.off[l];
    unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((size_t)p * cap + j) * 32);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = 64 * r + lane;
      int val[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float px = (float)spat[2 * (2 * t + e)], py = (float)spat[2 * (2 * t + e) + 1];
        const float xr = (px * ca) - (py * sa), yr = (px * sa) + (py * ca);
        const int xx = refl101(cx + __float2int_rn(xr), ww), yy = refl101(cy + __float2int_rn(yr), hh);
        val[e] = im[(size_t)yy * ww + xx];
      }
      const unsigned long long bal = __ballot(val[0] < val[1]);
      if (lane == 0) d[r] = bal;
    }
  }
}

Pyr make_pyr(int rows, int cols, int nfeatures) {
  Pyr P;
  memset(&P, 0, sizeof(P));
  long long off = 0;
  for (int l = 0; l < kLevels; ++l) {
    const double s = pow(1.2, (double)l);
    const int w = (int)lrint((double)cols / s), h = (int)lrint((double)rows / s);
    if (h < 1 || w < 1) break;
    P.h[l] = h;
    P.w[l] = w;
    P.off[l] = off;
    P.scale[l] = (float)s;
    off += (long long)h * w;
    P.nlev = l + 1;
  }
  P.total = off;
  const double f = 1.0 / 1.2;
  double nd = nfeatures * (1.0 - f) / (1.0 - pow(f, (double)kLevels));
  int sum = 0;
  for (int l = 0; l < kLevels - 1; ++l) {
    P.quota[l] = (int)lrint(nd);
    sum += P.quota[l];
    nd *= f;
  }
  P.quota[kLevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
  return P;
}

int32_t build_pyramid(sosvo_ctx* ctx, const uint8_t* gray, int nimg, int rows, int cols, const Pyr& P, uint8_t* pyr) {
  SOSVO_LAUNCH(ctx, copy_level0_kernel, dim3(cdiv(rows * cols, kThreads), nimg), dim3(kThreads), 0, ctx->stream, gray,
               rows * cols, P.total, pyr);
  SOSVO_LAUNCH_CHECK(ctx);
  for (int l = 1; l < P.nlev; ++l) {
    SOSVO_LAUNCH(ctx, resize_level_kernel, dim3(cdiv(P.h[l] * P.w[l], kThreads), nimg), dim3(kThreads), 0, ctx->stream, pyr,
                 P.total, P.off[l - 1], P.h[l - 1], P.w[l - 1], P.off[l], P.h[l], P.w[l]);
    SOSVO_LAUNCH_CHECK(ctx);
  }
  return SOSVO_OK;
}

}  // namespace

extern "C" {

int64_t sosvo_orb_pyramid_pixels(int32_t rows, int32_t cols) {
  if (rows < 1 || cols < 1) return 0;
  return make_pyr(rows, cols, 0).total;
}

int32_t sosvo_orb_mask_pyramid(sosvo_ctx* ctx, const uint32_t* mask_bits, int32_t nsets, int32_t rows, int32_t cols,
                               int32_t nmask, uint32_t* mask_pyr) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, mask_bits && mask_pyr, "null pointer");
  SOSVO_REQUIRE(ctx, nsets >= 1 && nsets <= 65535 && rows >= 1 && cols >= 1 && nmask >= 1 && nmask <= 32, "bad sizes");
  const Pyr P = make_pyr(rows, cols, 0);
  SOSVO_HIP(ctx, hipMemcpy2DAsync(mask_pyr, (size_t)P.total * 4, mask_bits, (size_t)rows * cols * 4,
                                  (size_t)rows * cols * 4, nsets, hipMemcpyDeviceToDevice, ctx->stream));
  for (int l = 1; l < P.nlev; ++l) {
    SOSVO_LAUNCH(ctx, mask_level_kernel, dim3(cdiv(P.h[l] * P.w[l], kThreads), nsets), dim3(kThreads), 0, ctx->stream,
                 mask_pyr, P.total, P.off[l - 1], P.h[l - 1], P.w[l - 1], P.off[l], P.h[l], P.w[l], nmask);
    SOSVO_LAUNCH_CHECK(ctx);
  }
  return SOSVO_OK;
}

int32_t sosvo_detect_orb(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_pyr, int32_t nimg,
                         int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask, int32_t nfeatures,
                         int32_t cap, float* kp4, float* resp, int32_t* n) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, gray && mask_pyr && kp4 && resp && n, "null pointer");
  SOSVO_REQUIRE(ctx, nimg >= 0 && nimg <= 65535 && images_per_maskset > 0, "nimg out of range");
  SOSVO_REQUIRE(ctx, rows >= 1 && cols >= 1 && cols < 65536 && rows < 65536 && rows * (int64_t)cols < (1 << 28),
                "image sizes out of range");
  SOSVO_REQUIRE(ctx, nmask >= 1 && nmask <= 32 && nfeatures >= 1 && cap >= 1 && cap <= 4096, "bad detector parameters");
  if (nimg == 0) return SOSVO_OK;
  const Pyr P = make_pyr(rows, cols, nfeatures);
  const size_t bytes = (size_t)nimg * P.total;
  const int nsets = cdiv(nimg, images_per_maskset);
  const size_t bbox_bytes = ((size_t)nsets * kLevels * 32 * 4 * sizeof(uint32_t) + 255) & ~(size_t)255;
  int32_t rc = sosvo_ws_reserve(ctx, 2 * ((bytes + 255) & ~(size_t)255) + bbox_bytes);
  if (rc != SOSVO_OK) return rc;
  uint8_t* pyr = (uint8_t*)ctx->ws;
  uint8_t* score = pyr + ((bytes + 255) & ~(size_t)255);
  uint32_t* bbox = (uint32_t*)(score + ((bytes + 255) & ~(size_t)255));
  rc = build_pyramid(ctx, gray, nimg, rows, cols, P, pyr);
  if (rc != SOSVO_OK) return rc;
  SOSVO_HIP(ctx, hipMemsetAsync(bbox, 0, bbox_bytes, ctx->stream));
  int total_rows = 0;
  for (int l = 0; l < P.nlev; ++l) total_rows += P.h[l];
  SOSVO_LAUNCH(ctx, orb_mask_bbox_kernel, dim3(total_rows, nsets), dim3(kThreads), 0, ctx->stream, mask_pyr, P, nmask, bbox);
  SOSVO_LAUNCH_CHECK(ctx);
  for (int l = 0; l < P.nlev; ++l) {  // only levels that can hold a keypoint (31-px border) and have a quota are scored
    const int enabled = P.quota[l] > 0 && P.h[l] > 2 * kEdge && P.w[l] > 2 * kEdge;
    rc = launch_fast_score(ctx, pyr + P.off[l], P.total, nimg, P.h[l], P.w[l], kFastThr, enabled, score + P.off[l]);
    if (rc != SOSVO_OK) return rc;
  }
  SOSVO_LAUNCH(ctx, orb_select_kernel, dim3((unsigned)((size_t)nimg * nmask)), dim3(kThreads), 0, ctx->stream, pyr, score,
               mask_pyr, bbox, P, images_per_maskset, nmask, cap, kp4, resp, n);
  SOSVO_LAUNCH_CHECK(ctx);
  ctx->pyr_gray = gray;  // the pyramid stays at the start of the scratch workspace for sosvo_describe_orb_levels
  ctx->pyr_nimg = nimg;
  ctx->pyr_rows = rows;
  ctx->pyr_cols = cols;
  return SOSVO_OK;
}

int32_t sosvo_describe_orb_levels(sosvo_ctx* ctx, const uint8_t* gray, int32_t nimg, int32_t rows, int32_t cols,
                                  int32_t nmask, int32_t cap, float* kp4, int32_t* n, const int8_t* pattern,
                                  uint8_t* desc, float* kp_xy) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, gray && kp4 && n && pattern && desc, "null pointer");
  SOSVO_REQUIRE(ctx, nimg >= 0 && nimg <= 65535 && nmask >= 1, "nimg / nmask out of range");
  SOSVO_REQUIRE(ctx, rows >= 1 && cols >= 1 && rows * (int64_t)cols < (1 << 28), "image sizes out of range");
  SOSVO_REQUIRE(ctx, cap > 0 && cap <= 2048, "cap out of range");
  SOSVO_REQUIRE(ctx, ((uintptr_t)desc & 7) == 0, "desc must be 8-byte aligned");
  if (nimg == 0) return SOSVO_OK;
  const Pyr P = make_pyr(rows, cols, 0);
  const size_t bytes = (size_t)nimg * P.total;
  // the pyramid sosvo_detect_orb built from the same images, if nothing has used the scratch workspace since
  const bool have_pyr = ctx->pyr_gray == (const void*)gray && ctx->pyr_nimg == nimg && ctx->pyr_rows == rows && ctx->pyr_cols == cols;
  const void* ws_before = ctx->ws;
  int32_t rc = sosvo_ws_reserve(ctx, 2 * ((bytes + 255) & ~(size_t)255));
  if (rc != SOSVO_OK) return rc;
  uint8_t* pyr = (uint8_t*)ctx->ws;
  uint8_t* blur = pyr + ((bytes + 255) & ~(size_t)255);
  if (!(have_pyr && ctx->ws == ws_before)) {
    rc = build_pyramid(ctx, gray, nimg, rows, cols, P, pyr);
    if (rc != SOSVO_OK) return rc;
  }
  for (int l = 0; l < P.nlev; ++l) {  // the rolling strip kernel of detect.hip, level by level (same integer arithmetic)
    rc = sosvo_launch_gauss7(ctx, pyr + P.off[l], P.total, nimg, P.h[l], P.w[l], blur + P.off[l]);
    if (rc != SOSVO_OK) return rc;
  }
  SOSVO_LAUNCH(ctx, orb_describe_levels_kernel, dim3((unsigned)((size_t)nimg * nmask)), dim3(kThreads),
               (size_t)cap * 4 * sizeof(float), ctx->stream, blur, P, rows, cols, nmask, cap, kp4, n, pattern, desc, kp_xy);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_detect_fast(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_bits, int32_t nimg,
                          int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask, int32_t threshold,
                          int32_t cap, float* kp, int32_t* n, int32_t* status) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, gray && mask_bits && kp && n, "null pointer");
  SOSVO_REQUIRE(ctx, nimg >= 0 && nimg <= 65535 && images_per_maskset > 0, "nimg out of range");
  SOSVO_REQUIRE(ctx, rows >= 7 && cols >= 7 && rows * (int64_t)cols < (1 << 28), "image sizes out of range");
  SOSVO_REQUIRE(ctx, nmask >= 1 && nmask <= 32, "nmask out of range (1..32)");
  SOSVO_REQUIRE(ctx, cap > 0 && cap <= 16384 && threshold >= 0 && threshold <= 254, "bad detector parameters");
  if (nimg == 0) return SOSVO_OK;
  const int words = cdiv(cols, 64);
  const size_t score_bytes = ((size_t)nimg * rows * cols + 255) & ~(size_t)255;
  const size_t flag_bytes = sizeof(unsigned long long) * (size_t)nimg * rows * words;
  int32_t rc = sosvo_ws_reserve(ctx, score_bytes + flag_bytes);
  if (rc != SOSVO_OK) return rc;
  uint8_t* score = (uint8_t*)ctx->ws;
  unsigned long long* flags = (unsigned long long*)((char*)ctx->ws + score_bytes);
  {
    const int32_t rc2 = launch_fast_score(ctx, gray, (long long)rows * cols, nimg, rows, cols, threshold, 1, score);
    if (rc2 != SOSVO_OK) return rc2;
  }
  SOSVO_LAUNCH(ctx, fast_nms_flags_kernel, dim3(cdiv(rows * words, kThreads / 64), nimg), dim3(kThreads), 0, ctx->stream, score,
               rows, cols, words, flags);
  SOSVO_LAUNCH_CHECK(ctx);
  SOSVO_LAUNCH(ctx, fast_collect_kernel, dim3((unsigned)((size_t)nimg * nmask)), dim3(64), 0, ctx->stream, flags, mask_bits,
               images_per_maskset, nmask, rows, cols, words, cap, kp, n, status);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

}  // extern "C"
