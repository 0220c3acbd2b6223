/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product
 * path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * CPU restatement of ORB keypoint detection and of ORB descriptors on multi-level oriented keypoints.
 * The arithmetic lives in OpenCV (cv2.ORB), a third-party dependency that is NOT under the reference
 * tree and is not pinned by it.  Reference call sites:
 *   omnistereo/camera_models.py:1640, :1755   ORB_create(nfeatures=N).detect(pano, mask)  (method "ORB")
 *   omnistereo/camera_models.py:1765          .compute(pano, keypoints)
 *   omnistereo/pose_est_tools.py:478, :547, :553   the same on the RGB-D image
 * Restated semantics (OpenCV's published ORB implementation, recalled, not
 * verifiable here): 8 levels x 1.2, per-level quota n_l = round(N (1-f)/(1-f^8) f^l) (last level takes the
 * rest), level images by bilinear resize from the previous level, FAST-9/16 threshold 20 with 3x3 NMS on
 * the FAST score, drop points within 31 px of the level border or outside the (resized, > 254) mask,
 * keep the best 2 n_l by FAST score (ties kept), Harris response (7x7 block, k = 0.04), keep the best n_l
 * (ties kept), orientation = fastAtan2(m01, m10) over the radius-15 disc (umax table), coordinates scaled
 * back to level 0; descriptors from the 7x7 sigma-2 blurred level image with the pattern rotated by the
 * keypoint angle and rounded.
 * OUR definitions where OpenCV's are version-dependent or unspecified (DESIGN.md "Image stages"): the
 * resize uses 11-bit fixed-point weights; retainBest's output order is (response descending, then y, then
 * x) -- OpenCV's comes out of nth_element and is unspecified; own descriptor pattern (oracle/image.c).
 * Parity status: UNPINNED against OpenCV binaries; pinned by KATs in tests/test_oracle_orb.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "trig_core.h"

#define ORB_LEVELS 8
#define ORB_EDGE 31
#define ORB_HALF_PATCH 15
#define ORB_FAST_THR 20
#define ORB_HARRIS_BLOCK 7

void orc_gauss7(const uint8_t* gray, int32_t rows, int32_t cols, uint8_t* out); /* oracle/image.c */

/* ---- pyramid ------------------------------------------------------------------------------------ */
void orc_orb_level_size(int32_t rows, int32_t cols, int32_t level, int32_t* lrows, int32_t* lcols) {
  const double s = pow(1.2, (double)level);
  *lcols = (int32_t)lrint((double)cols / s);
  *lrows = (int32_t)lrint((double)rows / s);
}

/* Bilinear resize with 11-bit fixed-point weights: sample position fx = (dx + 0.5) * (w0 / w1) - 0.5 in
 * float, clamped to the image; value = (sum of weight products + 2^21) >> 22. */
void orc_resize_linear(const uint8_t* src, int32_t h0, int32_t w0, uint8_t* dst, int32_t h1, int32_t w1) {
  const double sx = (double)w0 / w1, sy = (double)h0 / h1;
  for (int dy = 0; dy < h1; ++dy) {
    float fy = (float)(((double)dy + 0.5) * sy - 0.5);
    int iy = (int)floorf(fy);
    fy -= (float)iy;
    if (iy < 0) { iy = 0; fy = 0.f; }
    if (iy >= h0 - 1) { iy = h0 - 1; fy = 0.f; }
    const int iy1 = iy + 1 < h0 ? iy + 1 : h0 - 1;
    const int wy1 = (int)lrintf(fy * 2048.f), wy0 = 2048 - wy1;
    for (int dx = 0; dx < w1; ++dx) {
      float fx = (float)(((double)dx + 0.5) * sx - 0.5);
      int ix = (int)floorf(fx);
      fx -= (float)ix;
      if (ix < 0) { ix = 0; fx = 0.f; }
      if (ix >= w0 - 1) { ix = w0 - 1; fx = 0.f; }
      const int ix1 = ix + 1 < w0 ? ix + 1 : w0 - 1;
      const int wx1 = (int)lrintf(fx * 2048.f), wx0 = 2048 - wx1;
      const int64_t top = (int64_t)wx0 * src[(int64_t)iy * w0 + ix] + (int64_t)wx1 * src[(int64_t)iy * w0 + ix1];
      const int64_t bot = (int64_t)wx0 * src[(int64_t)iy1 * w0 + ix] + (int64_t)wx1 * src[(int64_t)iy1 * w0 + ix1];
      dst[(int64_t)dy * w1 + dx] = (uint8_t)((wy0 * top + wy1 * bot + (1 << 21)) >> 22);
    }
  }
}

/* ---- FAST-9/16 score map -------------------------------------------------------------------------- */
static const int kRingX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int kRingY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* score = (largest t such that 9 contiguous ring pixels are all > v + t or all < v - t) ... expressed as
 * max over the 16 arcs of the minimum |difference| along a same-sign arc, minus 1; 0 if not above thr. */
void orc_fast_score_map(const uint8_t* img, int32_t rows, int32_t cols, int32_t thr, uint8_t* score) {
  memset(score, 0, (size_t)rows * cols);
  for (int y = 3; y < rows - 3; ++y)
    for (int x = 3; x < cols - 3; ++x) {
      const int v = img[(int64_t)y * cols + x];
      int d[16];
      for (int k = 0; k < 16; ++k) d[k] = (int)img[(int64_t)(y + kRingY[k]) * cols + x + kRingX[k]] - v;
      int best = 0;
      for (int s = 0; s < 16; ++s) {
        int mn_b = 1 << 20, mn_d = 1 << 20;
        for (int j = 0; j < 9; ++j) {
          const int dd = d[(s + j) & 15];
          if (dd < mn_b) mn_b = dd;   /* brighter arc: min of (ring - v) */
          if (-dd < mn_d) mn_d = -dd; /* darker arc: min of (v - ring) */
        }
        if (mn_b > best) best = mn_b;
        if (mn_d > best) best = mn_d;
      }
      if (best > thr) score[(int64_t)y * cols + x] = (uint8_t)(best - 1);
    }
}

/* ---- AGAST (OAST 9/16) with non-maximum suppression -------------------------------------------------------
 * cv2.AgastFeatureDetector_create() (threshold 10, nonmaxSuppression, OAST_9_16) + .detect(image, mask)
 * (omnistereo/camera_models.py:1670-1671, :1755; pose_est_tools.py:508-509).  OpenCV's AGAST is not in the reference
 * tree; restated from the published method (Mair et al., ECCV 2010) and the recalled structure of its OpenCV port:
 *   - OAST 9/16 is the 9-of-16 accelerated segment test by an optimal decision tree: the SAME corner criterion as
 *     FAST-9/16, hence the same corner set (3-px border); the response is the largest threshold at which the pixel is
 *     still a corner (found by bisection there), i.e. the FAST score of orc_fast_score_map;
 *   - its non-maximum suppression is NOT a 3x3 test: corners that touch vertically or horizontally form blocks, and a
 *     raster scan with a union-find-like link table keeps one maximum per block (current corner against the block of
 *     the corner directly above, then against the block of the corner directly to the left; on equal responses the
 *     later corner takes over).
 * keep[rows * cols] receives 1 at the kept corners.  Parity status: UNPINNED against OpenCV binaries. */
void orc_agast_nms(const uint8_t* score, int32_t rows, int32_t cols, uint8_t* keep) {
  const int64_t npix = (int64_t)rows * cols;
  int32_t* idx_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)npix); /* pixel -> corner index, -1 */
  int64_t n = 0;
  for (int64_t i = 0; i < npix; ++i) {
    keep[i] = 0;
    idx_of[i] = score[i] ? (int32_t)n++ : -1;
  }
  int32_t* pix = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  int32_t* link = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < npix; ++i)
    if (idx_of[i] >= 0) {
      pix[idx_of[i]] = (int32_t)i;
      link[idx_of[i]] = -1;
    }
  for (int32_t cur = 0; cur < (int32_t)n; ++cur) {
    const int32_t y = pix[cur] / cols, x = pix[cur] - y * cols;
    const int resp = score[pix[cur]];
    if (y > 0 && idx_of[pix[cur] - cols] >= 0) { /* a corner directly above */
      int32_t w = idx_of[pix[cur] - cols];
      while (link[w] != -1) w = link[w];
      if (resp < (int)score[pix[w]]) link[cur] = w;
      else link[w] = cur;
    }
    if (x > 0 && idx_of[pix[cur] - 1] >= 0) { /* a corner directly to the left */
      int32_t t = idx_of[pix[cur] - 1];
      const int32_t above = link[cur];
      while (link[t] != -1) t = link[t];
      if (above == -1) {
        if (t != cur) {
          if (resp < (int)score[pix[t]]) link[cur] = t;
          else link[t] = cur;
        }
      } else if (t != above) {
        if ((int)score[pix[above]] < (int)score[pix[t]]) {
          link[above] = t;
          link[cur] = t;
        } else {
          link[t] = above;
          link[cur] = above;
        }
      }
    }
  }
  for (int32_t c = 0; c < (int32_t)n; ++c)
    if (link[c] == -1) keep[pix[c]] = 1;
  free(idx_of);
  free(pix);
  free(link);
}

/* ---- orientation and Harris ------------------------------------------------------------------------ */
static const int kUmax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

static float fast_atan2_deg(float y, float x) {
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

static int refl101o(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

static float ic_angle(const uint8_t* img, int rows, int cols, int cx, int cy) {
  int m01 = 0, m10 = 0;
  for (int u = -ORB_HALF_PATCH; u <= ORB_HALF_PATCH; ++u) m10 += u * img[(int64_t)cy * cols + refl101o(cx + u, cols)];
  for (int v = 1; v <= ORB_HALF_PATCH; ++v) {
    int vsum = 0;
    const int d = kUmax[v];
    const uint8_t* rp = img + (int64_t)refl101o(cy + v, rows) * cols;
    const uint8_t* rm = img + (int64_t)refl101o(cy - v, rows) * cols;
    for (int u = -d; u <= d; ++u) {
      const int xx = refl101o(cx + u, cols);
      const int vp = rp[xx], vm = rm[xx];
      vsum += vp - vm;
      m10 += u * (vp + vm);
    }
    m01 += v * vsum;
  }
  return fast_atan2_deg((float)m01, (float)m10);
}

static float harris_response(const uint8_t* img, int rows, int cols, int cx, int cy) {
  const int r = ORB_HARRIS_BLOCK / 2;
  int a = 0, b = 0, c = 0;
  for (int dy = -r; dy <= r; ++dy)
    for (int dx = -r; dx <= r; ++dx) {
      const int y = cy + dy, x = cx + dx;
#define PX(yy, xx) ((int)img[(int64_t)refl101o((yy), rows) * cols + refl101o((xx), cols)])
      const int Ix = (PX(y, x + 1) - PX(y, x - 1)) * 2 + (PX(y - 1, x + 1) - PX(y - 1, x - 1)) + (PX(y + 1, x + 1) - PX(y + 1, x - 1));
      const int Iy = (PX(y + 1, x) - PX(y - 1, x)) * 2 + (PX(y + 1, x - 1) - PX(y - 1, x - 1)) + (PX(y + 1, x + 1) - PX(y - 1, x + 1));
#undef PX
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  const float scale = 1.f / (4 * ORB_HARRIS_BLOCK * 255.f);
  const float s4 = (scale * scale) * (scale * scale);
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  return (((fa * fb) - (fc * fc)) - ((0.04f * (fa + fb)) * (fa + fb))) * s4;
}

typedef struct {
  float resp;
  int x, y, fast;
} okp_t;

static int okp_cmp(const void* pa, const void* pb) {
  const okp_t* a = (const okp_t*)pa;
  const okp_t* b = (const okp_t*)pb;
  if (a->resp > b->resp) return -1;
  if (a->resp < b->resp) return 1;
  if (a->y != b->y) return a->y < b->y ? -1 : 1;
  return a->x < b->x ? -1 : (a->x > b->x ? 1 : 0);
}

void orc_orb_quotas(int32_t nfeatures, int32_t* q /*[8]*/) {
  const double f = 1.0 / 1.2;
  double nd = nfeatures * (1.0 - f) / (1.0 - pow(f, (double)ORB_LEVELS));
  int sum = 0;
  for (int l = 0; l < ORB_LEVELS - 1; ++l) {
    q[l] = (int)lrint(nd);
    sum += q[l];
    nd *= f;
  }
  q[ORB_LEVELS - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

/* ORB.detect for every azimuthal mask of one image.  mask_bits as in orc_gft_select.  Outputs per mask m
 * (capacity cap each): kp [m][cap][4] = (x, y in level-0 coordinates, angle in degrees, level), resp
 * [m][cap] (Harris), n [m].  Keypoints are ordered by level, then response descending, then (y, x). */
void orc_orb_detect(const uint8_t* gray, int32_t rows, int32_t cols, const uint32_t* mask_bits, int32_t nmask,
                    int32_t nfeatures, int32_t cap, float* kp, float* resp, int32_t* n) {
  int quota[ORB_LEVELS];
  orc_orb_quotas(nfeatures, quota);
  for (int m = 0; m < nmask; ++m) n[m] = 0;
  uint8_t* img = (uint8_t*)malloc((size_t)rows * cols);
  memcpy(img, gray, (size_t)rows * cols);
  /* one 0/255 mask image per azimuthal mask, resized level by level */
  uint8_t** mk = (uint8_t**)malloc(sizeof(uint8_t*) * (size_t)nmask);
  for (int m = 0; m < nmask; ++m) {
    mk[m] = (uint8_t*)malloc((size_t)rows * cols);
    for (int64_t i = 0; i < (int64_t)rows * cols; ++i) mk[m][i] = ((mask_bits[i] >> m) & 1u) ? 255 : 0;
  }
  int h = rows, w = cols;
  for (int l = 0; l < ORB_LEVELS; ++l) {
    if (l > 0) {
      int h1, w1;
      orc_orb_level_size(rows, cols, l, &h1, &w1);
      if (h1 < 1 || w1 < 1) break;
      uint8_t* ni = (uint8_t*)malloc((size_t)h1 * w1);
      orc_resize_linear(img, h, w, ni, h1, w1);
      free(img);
      img = ni;
      for (int m = 0; m < nmask; ++m) {
        uint8_t* nm = (uint8_t*)malloc((size_t)h1 * w1);
        orc_resize_linear(mk[m], h, w, nm, h1, w1);
        for (int64_t i = 0; i < (int64_t)h1 * w1; ++i) nm[i] = nm[i] > 254 ? 255 : 0;
        free(mk[m]);
        mk[m] = nm;
      }
      h = h1;
      w = w1;
    }
    if (quota[l] <= 0 || h <= 2 * ORB_EDGE || w <= 2 * ORB_EDGE) continue;
    uint8_t* score = (uint8_t*)malloc((size_t)h * w);
    orc_fast_score_map(img, h, w, ORB_FAST_THR, score);
    const float scale = (float)pow(1.2, (double)l);
    for (int m = 0; m < nmask; ++m) {
      okp_t* c = (okp_t*)malloc(sizeof(okp_t) * (size_t)h * w);
      int nc = 0;
      for (int y = ORB_EDGE; y < h - ORB_EDGE; ++y)
        for (int x = ORB_EDGE; x < w - ORB_EDGE; ++x) {
          const int s = score[(int64_t)y * w + x];
          if (!s || !mk[m][(int64_t)y * w + x]) continue;
          int is_max = 1;
          for (int dy = -1; dy <= 1 && is_max; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
              if ((dy || dx) && score[(int64_t)(y + dy) * w + x + dx] >= s) {
                is_max = 0;
                break;
              }
          if (is_max) {
            c[nc].fast = s;
            c[nc].x = x;
            c[nc].y = y;
            nc++;
          }
        }
      /* retainBest(2 n_l) by FAST score, ties kept */
      const int want2 = 2 * quota[l];
      if (nc > want2) {
        int hist[256] = {0};
        for (int i = 0; i < nc; ++i) hist[c[i].fast]++;
        int acc = 0, thr = 255;
        for (thr = 255; thr >= 0; --thr) {
          acc += hist[thr];
          if (acc >= want2) break;
        }
        int k = 0;
        for (int i = 0; i < nc; ++i)
          if (c[i].fast >= thr) c[k++] = c[i];
        nc = k;
      }
      for (int i = 0; i < nc; ++i) c[i].resp = harris_response(img, h, w, c[i].x, c[i].y);
      qsort(c, (size_t)nc, sizeof(okp_t), okp_cmp);
      /* retainBest(n_l) by Harris response, ties kept */
      int keep = nc;
      if (nc > quota[l]) {
        const float amb = c[quota[l] - 1].resp;
        keep = quota[l];
        while (keep < nc && c[keep].resp >= amb) keep++;
      }
      for (int i = 0; i < keep && n[m] < cap; ++i) {
        float* o = kp + ((int64_t)m * cap + n[m]) * 4;
        o[0] = (float)c[i].x * scale;
        o[1] = (float)c[i].y * scale;
        o[2] = ic_angle(img, h, w, c[i].x, c[i].y);
        o[3] = (float)l;
        resp[(int64_t)m * cap + n[m]] = c[i].resp;
        n[m]++;
      }
      free(c);
    }
    free(score);
  }
  free(img);
  for (int m = 0; m < nmask; ++m) free(mk[m]);
  free(mk);
}

/* ORB.compute on oriented multi-level keypoints kp [n][4] = (x, y, angle_deg, level).  Keypoints within 31
 * px of the level-0 border are dropped.  Returns the number kept. */
int32_t orc_orb_describe_levels(const uint8_t* gray, int32_t rows, int32_t cols, const float* kp, int32_t n,
                                const int8_t* pattern, uint8_t* desc, int32_t* kept_idx) {
  uint8_t* blur[ORB_LEVELS];
  int hs[ORB_LEVELS], ws[ORB_LEVELS];
  uint8_t* img = (uint8_t*)malloc((size_t)rows * cols);
  memcpy(img, gray, (size_t)rows * cols);
  int h = rows, w = cols, nl = 0;
  for (int l = 0; l < ORB_LEVELS; ++l) {
    if (l > 0) {
      int h1, w1;
      orc_orb_level_size(rows, cols, l, &h1, &w1);
      if (h1 < 1 || w1 < 1) break;
      uint8_t* ni = (uint8_t*)malloc((size_t)h1 * w1);
      orc_resize_linear(img, h, w, ni, h1, w1);
      free(img);
      img = ni;
      h = h1;
      w = w1;
    }
    blur[l] = (uint8_t*)malloc((size_t)h * w);
    orc_gauss7(img, h, w, blur[l]);
    hs[l] = h;
    ws[l] = w;
    nl = l + 1;
  }
  free(img);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    const float x = kp[4 * i], y = kp[4 * i + 1];
    const int l = (int)kp[4 * i + 3];
    if (!(x >= (float)ORB_EDGE && x < (float)(cols - ORB_EDGE) && y >= (float)ORB_EDGE && y < (float)(rows - ORB_EDGE))) continue;
    if (l < 0 || l >= nl) continue;
    const float scale = 1.f / (float)pow(1.2, (double)l);
    float angle = kp[4 * i + 2];
    angle *= (float)(3.14159265358979323846 / 180.0);
    double sd, cd; /* trig_core.h: the same arithmetic as on the device */
    orc_sincos((double)angle, &sd, &cd);
    const float ca = (float)cd, sa = (float)sd;
    const int cx = (int)lrintf(x * scale), cy = (int)lrintf(y * scale);
    const int hh = hs[l], ww = ws[l];
    uint8_t* d = desc + 32 * (int64_t)kept;
    for (int byte = 0; byte < 32; ++byte) {
      int v = 0;
      for (int bit = 0; bit < 8; ++bit) {
        const int t = byte * 8 + bit;
        int val[2];
        for (int e = 0; e < 2; ++e) {
          const float px = (float)pattern[2 * (2 * t + e)], py = (float)pattern[2 * (2 * t + e) + 1];
          const float xr = (px * ca) - (py * sa), yr = (px * sa) + (py * ca);
          const int xx = refl101o(cx + (int)lrintf(xr), ww), yy = refl101o(cy + (int)lrintf(yr), hh);
          val[e] = blur[l][(int64_t)yy * ww + xx];
        }
        v |= (val[0] < val[1]) << bit;
      }
      d[byte] = (uint8_t)v;
    }
    kept_idx[kept++] = i;
  }
  for (int l = 0; l < nl; ++l) free(blur[l]);
  return kept;
}
