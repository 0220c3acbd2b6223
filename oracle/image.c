/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product
 * path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * CPU restatement of the image stages of the hot path.  The arithmetic lives in OpenCV (cv2), a
 * third-party dependency that is NOT under the reference tree and is not pinned by it ("OpenCV 3" +
 * contrib, reference README.md:119-190).  The restatement follows OpenCV's published/documented
 * algorithms and is anchored on the reference's call sites:
 *   omnistereo/panorama.py:293                cv2.remap(omni, map_x f32, map_y f32, INTER_LINEAR, BORDER_CONSTANT, 0)
 *   omnistereo/camera_models.py:2991-2996     bitwise_and with the cached annulus masks (folded into the taps)
 *   omnistereo/camera_models.py:1711          cv2.medianBlur(pano, 11)
 *   omnistereo/camera_models.py:1714          cv2.cvtColor(BGR2GRAY)
 *   omnistereo/camera_models.py:1739          cv2.goodFeaturesToTrack(maxCorners, 0.01, 5, mask, useHarris=False)
 *   omnistereo/camera_models.py:1765          ORB_create(nfeatures).compute(image, keypoints)  (angle -1 from KeyPoint_convert)
 * Restated semantics (recalled from OpenCV's public sources, not verifiable here):
 *   remap: map coordinates rounded to 1/32 px (ties to even), bilinear weights (32-fx)(32-fy)/1024,
 *          result (sum + 512) >> 10, taps outside the image (or NaN maps) read the border value 0;
 *   medianBlur: exact median of the k x k window per channel, replicated border;
 *   BGR2GRAY: (1868 B + 9617 G + 4899 R + 8192) >> 14;
 *   goodFeaturesToTrack: min-eigenvalue map (Sobel 3, block 3, reflect-101), threshold quality * max over
 *          the mask, 3x3 dilation NMS, descending sort (ties: higher address first), greedy minimum
 *          distance on a cell grid, stop at maxCorners;
 *   ORB.compute on provided keypoints: drop keypoints within 31 px of the border, 7x7 sigma=2 Gaussian,
 *          256 pair tests on offsets rotated by the keypoint angle and rounded, 8 tests per byte, LSB first.
 * Documented deviations (DESIGN.md "Image stages"): (1) closed in round 4: the default test pattern IS OpenCV's learned
 * 256-pair table (orc_orb_pattern_opencv, oracle/orb_bit_pattern_31.h, generated from the data file the product ships);
 * the seeded table of rounds 1-3 stays as orc_orb_pattern.  (2) The float32
 * evaluation order of the eigenvalue map and the 8.8 fixed-point Gaussian are OUR definitions (OpenCV's
 * differ in rounding details by version).  (3) The annulus masks are x^2 + y^2 <= r^2 discs built on the
 * host, not OpenCV's circle rasteriser.
 * Parity status: UNPINNED against OpenCV binaries (no golden vectors exist in the reference); pinned by
 * hand-checkable known-answer tests in tests/test_oracle_image.py and by third-party implementations on photographs
 * (tests/test_oracle_thirdparty.py: median, FAST-9 corner set, remap, ORB orientation / Harris response, the Gaussian, and --
 * every descriptor byte -- the rotated-BRIEF sampling rule against scikit-image's ORB descriptor loop).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- K1: unwrap = remap(INTER_LINEAR, BORDER_CONSTANT 0) of the masked omni image --------------- */
static int round_half_even_f(float v) { return (int)lrintf(v); /* default rounding mode: to nearest even */ }

void orc_unwrap(const uint8_t* omni, const uint8_t* mask, int32_t H, int32_t W, const float* map_x,
                const float* map_y, int32_t rows, int32_t cols, uint8_t* pano) {
  for (int64_t i = 0; i < (int64_t)rows * cols; ++i) {
    const float mx = map_x[i], my = map_y[i];
    uint8_t* out = pano + 3 * i;
    out[0] = out[1] = out[2] = 0;
    if (!(mx == mx) || !(my == my)) continue;                       /* NaN map entry -> border colour */
    if (!(mx > -4.0f && mx < (float)W + 4.0f && my > -4.0f && my < (float)H + 4.0f)) continue;
    const int sx = round_half_even_f(mx * 32.0f), sy = round_half_even_f(my * 32.0f);
    const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
    const int w[4] = {(32 - fx) * (32 - fy), fx * (32 - fy), (32 - fx) * fy, fx * fy};
    const int xs[4] = {ix, ix + 1, ix, ix + 1}, ys[4] = {iy, iy, iy + 1, iy + 1};
    int acc[3] = {0, 0, 0};
    for (int t = 0; t < 4; ++t) {
      if (xs[t] < 0 || xs[t] >= W || ys[t] < 0 || ys[t] >= H) continue;  /* border value 0 */
      const int64_t o = (int64_t)ys[t] * W + xs[t];
      if (mask && !mask[o]) continue;                                     /* masked-out pixel is 0 */
      for (int c = 0; c < 3; ++c) acc[c] += w[t] * omni[3 * o + c];
    }
    for (int c = 0; c < 3; ++c) out[c] = (uint8_t)((acc[c] + 512) >> 10);
  }
}

/* ---- K2 + K3: k x k median per channel (replicated border), then BGR -> gray --------------------- */
static void median_channel(const uint8_t* src, int rows, int cols, int nch, int ch, int k, uint8_t* dst) {
  const int r = k / 2, half = (k * k) / 2 + 1; /* the median is the half-th smallest (1-based) */
  for (int y = 0; y < rows; ++y) {
    int hist[256];
    memset(hist, 0, sizeof(hist));
    /* Huang's sliding histogram along the row */
    for (int dy = -r; dy <= r; ++dy) {
      int yy = y + dy;
      yy = yy < 0 ? 0 : (yy >= rows ? rows - 1 : yy);
      for (int dx = -r; dx <= r; ++dx) {
        int xx = dx < 0 ? 0 : (dx >= cols ? cols - 1 : dx);
        hist[src[((int64_t)yy * cols + xx) * nch + ch]]++;
      }
    }
    for (int x = 0; x < cols; ++x) {
      int acc = 0, m = 0;
      for (m = 0; m < 256; ++m) {
        acc += hist[m];
        if (acc >= half) break;
      }
      dst[(int64_t)y * cols + x] = (uint8_t)m;
      /* slide: remove column x - r, add column x + r + 1 */
      int xo = x - r, xi = x + r + 1;
      xo = xo < 0 ? 0 : (xo >= cols ? cols - 1 : xo);
      xi = xi < 0 ? 0 : (xi >= cols ? cols - 1 : xi);
      for (int dy = -r; dy <= r; ++dy) {
        int yy = y + dy;
        yy = yy < 0 ? 0 : (yy >= rows ? rows - 1 : yy);
        hist[src[((int64_t)yy * cols + xo) * nch + ch]]--;
        hist[src[((int64_t)yy * cols + xi) * nch + ch]]++;
      }
    }
  }
}

static uint8_t bgr2gray(int b, int g, int r) { return (uint8_t)((1868 * b + 9617 * g + 4899 * r + 8192) >> 14); }

/* ksize <= 1: no blur.  blurred_bgr may be NULL. */
void orc_median_gray(const uint8_t* pano, int32_t rows, int32_t cols, int32_t ksize, uint8_t* gray,
                     uint8_t* blurred_bgr) {
  const int64_t n = (int64_t)rows * cols;
  uint8_t* ch[3];
  for (int c = 0; c < 3; ++c) {
    ch[c] = (uint8_t*)malloc((size_t)n);
    if (ksize > 1)
      median_channel(pano, rows, cols, 3, c, ksize, ch[c]);
    else
      for (int64_t i = 0; i < n; ++i) ch[c][i] = pano[3 * i + c];
  }
  for (int64_t i = 0; i < n; ++i) {
    gray[i] = bgr2gray(ch[0][i], ch[1][i], ch[2][i]);
    if (blurred_bgr)
      for (int c = 0; c < 3; ++c) blurred_bgr[3 * i + c] = ch[c][i];
  }
  for (int c = 0; c < 3; ++c) free(ch[c]);
}

/* ---- K4a: minimum-eigenvalue map (cornerMinEigenVal, blockSize 3, ksize 3, reflect-101) ---------- */
static int refl101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

void orc_min_eigen(const uint8_t* gray, int32_t rows, int32_t cols, float* eig) {
  const float scale = (float)(1.0 / 3060.0); /* 1 / (2^(ksize-1) * blockSize * 255) */
  const int64_t n = (int64_t)rows * cols;
  float* xx = (float*)malloc(sizeof(float) * (size_t)n);
  float* xy = (float*)malloc(sizeof(float) * (size_t)n);
  float* yy = (float*)malloc(sizeof(float) * (size_t)n);
  for (int y = 0; y < rows; ++y) {
    const uint8_t* r0 = gray + (int64_t)refl101(y - 1, rows) * cols;
    const uint8_t* r1 = gray + (int64_t)y * cols;
    const uint8_t* r2 = gray + (int64_t)refl101(y + 1, rows) * cols;
    for (int x = 0; x < cols; ++x) {
      const int xl = refl101(x - 1, cols), xr = refl101(x + 1, cols);
      const int dxi = (r0[xr] + 2 * r1[xr] + r2[xr]) - (r0[xl] + 2 * r1[xl] + r2[xl]);
      const int dyi = (r2[xl] + 2 * r2[x] + r2[xr]) - (r0[xl] + 2 * r0[x] + r0[xr]);
      const float dx = (float)dxi * scale, dy = (float)dyi * scale;
      xx[(int64_t)y * cols + x] = dx * dx;
      xy[(int64_t)y * cols + x] = dx * dy;
      yy[(int64_t)y * cols + x] = dy * dy;
    }
  }
  /* 3x3 box sums: horizontal (left + centre) + right, then (above + centre) + below */
  for (int y = 0; y < rows; ++y) {
    const int ya = refl101(y - 1, rows), yb = refl101(y + 1, rows);
    for (int x = 0; x < cols; ++x) {
      const int xl = refl101(x - 1, cols), xr = refl101(x + 1, cols);
      float s[3];
      const float* src[3] = {xx, xy, yy};
      for (int k = 0; k < 3; ++k) {
        const float* v = src[k];
        const float ha = (v[(int64_t)ya * cols + xl] + v[(int64_t)ya * cols + x]) + v[(int64_t)ya * cols + xr];
        const float hc = (v[(int64_t)y * cols + xl] + v[(int64_t)y * cols + x]) + v[(int64_t)y * cols + xr];
        const float hb = (v[(int64_t)yb * cols + xl] + v[(int64_t)yb * cols + x]) + v[(int64_t)yb * cols + xr];
        s[k] = (ha + hc) + hb;
      }
      const float a = s[0] * 0.5f, b = s[1], c = s[2] * 0.5f;
      eig[(int64_t)y * cols + x] = (a + c) - sqrtf(((a - c) * (a - c)) + (b * b));
    }
  }
  free(xx);
  free(xy);
  free(yy);
}

/* ---- K4b: goodFeaturesToTrack selection for one mask ---------------------------------------------
 * Bit `which` of mask_bits[y*cols+x] set = the pixel belongs to this mask (masks may overlap).  Returns the number of
 * corners written to kp_xy (x, y as float, in acceptance order = descending quality). */
typedef struct {
  float v;
  int32_t idx;
} cand_t;

static int cand_cmp(const void* pa, const void* pb) {
  const cand_t* a = (const cand_t*)pa;
  const cand_t* b = (const cand_t*)pb;
  if (a->v > b->v) return -1;
  if (a->v < b->v) return 1;
  return (a->idx > b->idx) ? -1 : (a->idx < b->idx ? 1 : 0); /* ties: higher address first */
}

int32_t orc_gft_select(const float* eig, const uint32_t* mask_bits, int32_t which, int32_t rows, int32_t cols,
                       double quality, double min_distance, int32_t max_corners, float* kp_xy,
                       float* max_val_out) {
  float maxv = -INFINITY;
  int any = 0;
  for (int64_t i = 0; i < (int64_t)rows * cols; ++i)
    if ((mask_bits[i] >> which) & 1u) {
      if (!any || eig[i] > maxv) maxv = eig[i];
      any = 1;
    }
  if (max_val_out) *max_val_out = any ? maxv : 0.0f;
  if (!any) return 0;
  const float thr = (float)((double)maxv * quality); /* threshold(eig, maxVal * qualityLevel, THRESH_TOZERO) */
  cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * (size_t)rows * cols);
  int nc = 0;
  for (int y = 1; y < rows - 1; ++y)
    for (int x = 1; x < cols - 1; ++x) {
      const int64_t i = (int64_t)y * cols + x;
      if (!((mask_bits[i] >> which) & 1u)) continue;
      const float v = eig[i] > thr ? eig[i] : 0.0f;
      if (v == 0.0f) continue;
      float dil = v; /* 3x3 dilation of the thresholded map */
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const float e = eig[i + (int64_t)dy * cols + dx];
          const float t = e > thr ? e : 0.0f;
          if (t > dil) dil = t;
        }
      if (v == dil) {
        cand[nc].v = v;
        cand[nc].idx = (int32_t)i;
        nc++;
      }
    }
  qsort(cand, (size_t)nc, sizeof(cand_t), cand_cmp);
  int ncorners = 0;
  if (min_distance >= 1) {
    const int cell = (int)lrint(min_distance);
    const int gw = (cols + cell - 1) / cell, gh = (rows + cell - 1) / cell;
    /* up to 8 points per cell is ample for cell size == min distance */
    int32_t* gx = (int32_t*)malloc(sizeof(int32_t) * (size_t)gw * gh * 8); /* (32-bit: images may be wider than 32767) */
    int32_t* gy = (int32_t*)malloc(sizeof(int32_t) * (size_t)gw * gh * 8);
    uint8_t* gn = (uint8_t*)calloc((size_t)gw * gh, 1);
    const float md2 = (float)(min_distance * min_distance);
    for (int k = 0; k < nc; ++k) {
      const int y = cand[k].idx / cols, x = cand[k].idx % cols;
      const int xc = x / cell, yc = y / cell;
      int x1 = xc - 1, y1 = yc - 1, x2 = xc + 1, y2 = yc + 1;
      x1 = x1 < 0 ? 0 : x1;
      y1 = y1 < 0 ? 0 : y1;
      x2 = x2 > gw - 1 ? gw - 1 : x2;
      y2 = y2 > gh - 1 ? gh - 1 : y2;
      int good = 1;
      for (int yy = y1; yy <= y2 && good; ++yy)
        for (int xx2 = x1; xx2 <= x2 && good; ++xx2) {
          const int c = yy * gw + xx2;
          for (int j = 0; j < gn[c]; ++j) {
            const float dx = (float)(x - gx[c * 8 + j]), dy = (float)(y - gy[c * 8 + j]);
            if (dx * dx + dy * dy < md2) {
              good = 0;
              break;
            }
          }
        }
      if (good) {
        const int c = yc * gw + xc;
        if (gn[c] < 8) {
          gx[c * 8 + gn[c]] = x;
          gy[c * 8 + gn[c]] = y;
          gn[c]++;
        }
        kp_xy[2 * ncorners] = (float)x;
        kp_xy[2 * ncorners + 1] = (float)y;
        ncorners++;
        if (max_corners > 0 && ncorners == max_corners) break;
      }
    }
    free(gx);
    free(gy);
    free(gn);
  } else {
    for (int k = 0; k < nc; ++k) {
      kp_xy[2 * ncorners] = (float)(cand[k].idx % cols);
      kp_xy[2 * ncorners + 1] = (float)(cand[k].idx / cols);
      ncorners++;
      if (max_corners > 0 && ncorners == max_corners) break;
    }
  }
  free(cand);
  return ncorners;
}

/* ---- K6a: 7x7 sigma = 2 Gaussian, 8.8 fixed point, reflect-101 ------------------------------------ */
static const int kGauss7[7] = {18, 34, 49, 54, 49, 34, 18}; /* round(256 * exp(-d^2/8) / sum), sums to 256 */

void orc_gauss7(const uint8_t* gray, int32_t rows, int32_t cols, uint8_t* out) {
  uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)rows * cols);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int s = 0;
      for (int k = -3; k <= 3; ++k) s += kGauss7[k + 3] * gray[(int64_t)y * cols + refl101(x + k, cols)];
      tmp[(int64_t)y * cols + x] = (uint16_t)s; /* <= 255 * 256 */
    }
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      uint32_t s = 0;
      for (int k = -3; k <= 3; ++k) s += (uint32_t)kGauss7[k + 3] * tmp[(int64_t)refl101(y + k, rows) * cols + x];
      out[(int64_t)y * cols + x] = (uint8_t)((s + 32768u) >> 16);
    }
  free(tmp);
}

/* ---- K6b: the 256 test pairs ----------------------------------------------------------------------
 * default: OpenCV's bit_pattern_31_ */
#include "orb_bit_pattern_31.h"
void orc_orb_pattern_opencv(int8_t* pts /* [512][2] */) { memcpy(pts, orc_orb_bit_pattern_31, 1024); }

/* the seeded pattern of rounds 1-3 (kept as an option) ------------------------------------------------
 * 512 points (x, y) as int8, |x|, |y| <= 12: sum of three uniform draws in [-4, 4] (bell-shaped, as the
 * BRIEF paper recommends), drawn from a splitmix64 stream with a fixed seed. */
void orc_orb_pattern(int8_t* pts /* [512][2] */) {
  uint64_t state = 0x0B5EED5EED5EED01ULL;
  for (int i = 0; i < 1024; ++i) {
    int v = 0;
    for (int k = 0; k < 3; ++k) {
      state += 0x9E3779B97F4A7C15ULL;
      uint64_t z = state;
      z ^= z >> 30;
      z *= 0xBF58476D1CE4E5B9ULL;
      z ^= z >> 27;
      z *= 0x94D049BB133111EBULL;
      z ^= z >> 31;
      v += (int)((z >> 32) % 9u) - 4;
    }
    pts[i] = (int8_t)v;
  }
  /* a test whose two points coincide carries no information: nudge the second point */
  for (int t = 0; t < 256; ++t)
    if (pts[4 * t] == pts[4 * t + 2] && pts[4 * t + 1] == pts[4 * t + 3]) pts[4 * t + 2] = (int8_t)(pts[4 * t + 2] < 12 ? pts[4 * t + 2] + 1 : pts[4 * t + 2] - 1);
}

/* ---- K6c: ORB.compute on provided keypoints --------------------------------------------------------
 * cos_a / sin_a: float32 cosine / sine of the keypoint angle (the GFT path passes angle -1 degree for
 * every keypoint: KeyPoint_convert default).  Keypoints within `edge` (31) px of the border are dropped.
 * Returns the number kept; kept_idx[j] = index of the j-th kept input keypoint. */
int32_t orc_orb_describe(const uint8_t* blurred, int32_t rows, int32_t cols, const float* kp_xy, int32_t n,
                         float cos_a, float sin_a, const int8_t* pattern, int32_t edge, uint8_t* desc,
                         int32_t* kept_idx) {
  int off[512];
  for (int i = 0; i < 512; ++i) {
    const float px = (float)pattern[2 * i], py = (float)pattern[2 * i + 1];
    const float xr = (px * cos_a) - (py * sin_a), yr = (px * sin_a) + (py * cos_a);
    off[i] = (int)lrintf(yr) * cols + (int)lrintf(xr);
  }
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    const float x = kp_xy[2 * i], y = kp_xy[2 * i + 1];
    if (!(x >= (float)edge && x < (float)(cols - edge) && y >= (float)edge && y < (float)(rows - edge))) continue;
    const uint8_t* c = blurred + (int64_t)lrintf(y) * cols + lrintf(x);
    uint8_t* d = desc + 32 * (int64_t)kept;
    for (int byte = 0; byte < 32; ++byte) {
      int v = 0;
      for (int bit = 0; bit < 8; ++bit) {
        const int t = byte * 8 + bit;
        v |= (c[off[2 * t]] < c[off[2 * t + 1]]) << bit;
      }
      d[byte] = (uint8_t)v;
    }
    kept_idx[kept++] = i;
  }
  return kept;
}
