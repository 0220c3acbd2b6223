// K5 -- ORB keypoint detection per azimuthal mask, and K6' -- ORB descriptors for oriented multi-level keypoints.
//
//   cv2.ORB_create(nfeatures=N).detect(pano, mask)   omnistereo/camera_models.py:1640, :1755; pose_est_tools.py:478, :547
//   .compute(pano, keypoints)                        camera_models.py:1765; pose_est_tools.py:553
//
// Batched over images (view-major) x azimuthal masks; problem p = image * nmask + mask.
//   levels    8 levels x 1.2, each level bilinear (11-bit fixed point) from the previous one; all levels of one image live
//             back to back in one workspace row (level offsets in a by-value struct).  ONE launch per level
//             (orb_level_pass_kernel, round 4): a wave walks a 128-column strip of level l (two pixels per lane) down its rows and produces, from one
//             read of the level, the FAST-9/16 local-maximum flags (+ the flagged pixels' scores), the 7x7 blur of the rows a
//             descriptor can read, and level l + 1.  FAST score: one polarity per lane on raw pixel values, van Herk arc network.
//             (Images taller than the row table's LDS area, and keypoints handed in from outside, take the separate
//             resize / FAST / blur kernels.)
//   select    one workgroup per problem walks the levels: flags + 31-px border + mask -> candidates in LDS, keep the best
//             2 n_l by FAST score through a 256-bin LDS histogram (ties kept), Harris response per candidate (a LANE per
//             candidate: separable Sobel on a three-row ring), rank sort (response desc, then y, x), keep n_l (+ ties),
//             intensity-centroid angle (a lane per keypoint: v_dot4_u32_u8 with the disc as byte weights), fastAtan2 polynomial
//             in float32 with a pinned operation order.  The rectangle of the level under the mask's candidates is staged in
//             LDS: the Harris blocks and orientation patches read it instead of memory;
//   describe  border rule on level-0 coordinates + stable compaction, then one wave per keypoint: the rotated offsets are
//             evaluated per lane in float32, four 64-bit ballots are the 32 descriptor bytes.  Per level the bounding box of
//             the problem's keypoints grown by the pattern's reach is staged in LDS once; every test is an LDS byte read.
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
constexpr int kCandMax = 1024;  // candidates per (problem, level) held in LDS (more: the threshold is still exact, a second walk
                                // keeps the retained ones; 1024 instead of 2048 = 19 KB of LDS: five workgroups per CU, not four)

struct Pyr {
  int h[kLevels], w[kLevels];
  long long off[kLevels];
  float scale[kLevels];
  int quota[kLevels];
  int nlev;          // levels that exist (h, w >= 1)
  long long total;   // pixels of one image's pyramid
  // detection: levels that can hold a keypoint (a quota and more than the 31-px border); local-maximum flags of their
  // FAST score maps, one u64 per (row, 56-column strip): foff = first word of a level inside an image's ftotal words
  int det[kLevels], ndet;  // ndet = 1 + the highest such level (the pyramid is only built that far for detection)
  int fstrips[kLevels];    // 56-column strips of a level (fast_score_rolling_kernel: one flag word each)
  int fstrips2[kLevels];   // 120-column strips (orb_level_pass_kernel: TWO flag words each, even and odd columns)
  long long foff[kLevels], ftotal;  // (a level's rows hold max(fstrips, 2 fstrips2) words: either layout fits)
  int toff[kLevels], ttotal;  // resize tap table: level l >= 1 holds w[l] x taps, then h[l] y taps, at toff[l]
};

// image of level l: level 0 is the caller's gray batch itself (no copy), levels >= 1 live in the pyramid workspace
struct LevelSrc {
  const uint8_t* gray;  // [nimg, rows * cols]
  const uint8_t* pyr;   // [nimg, total] (level 0's slot is unused)
  long long npix0, total;
};
__device__ __forceinline__ const uint8_t* level_image(const LevelSrc& S, const Pyr& P, int img, int l) {
  return l == 0 ? S.gray + (size_t)img * S.npix0 : S.pyr + (size_t)img * S.total + P.off[l];
}

// Workgroup -> problem (image, mask), XCD-aware (as detect.hip's xcd_problem): workgroups are dealt round-robin over the 8
// XCDs by linear id, so p = blockIdx.x spreads the 12 masks of one image over all eight L2s and every L2 fetches that
// image's pyramid (round 3's PMC pass: 6.8 MB of fetches per frame pair in the selection alone).  Here XCD x takes the images
// x, x + 8, ... with all their masks: grid = 8 * ceil(nimg / 8) * nmask, -1 for the padding ids.
__device__ __forceinline__ int orb_xcd_problem(int nimg, int nmask) {
  const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
  const int img = (slot / nmask) * 8 + xcd, m = slot - (slot / nmask) * nmask;
  return img < nimg ? img * nmask + m : -1;
}
inline unsigned orb_xcd_grid(int nimg, int nmask) { return (unsigned)(8 * ((nimg + 7) / 8) * nmask); }

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

// Tap table of the bilinear resizes (depends on the image size only): entry = i0 | w1 << 16 (i1 = min(i0 + 1, n0 - 1),
// w0 = 2048 - w1), computed once per call instead of twice per output pixel (double-precision arithmetic).
__global__ __launch_bounds__(kThreads) void resize_taps_kernel(Pyr P, uint32_t* __restrict__ taps, int32_t* __restrict__ lvl_rows,
                                                               const int8_t* __restrict__ pattern) {
  const int i = blockIdx.x * kThreads + threadIdx.x, l = blockIdx.y + 1;
  // (piggy-backed: the rows of every level's BLURRED image a descriptor of a detected keypoint can read -- the detector keeps
  // 31 px of border on the level, a rotated and rounded test point of THIS pattern stays within `reach` px, computed as
  // orb_describe_levels_kernel computes its margin -- for the row-restricted blur)
  if (lvl_rows && blockIdx.x == 0 && blockIdx.y == 0) {  // uniform per workgroup
    __shared__ int s_cmax;
    if (threadIdx.x == 0) s_cmax = 0;
    __syncthreads();
    int c = 0;
    for (int k = threadIdx.x; k < 1024; k += kThreads) c = max(c, abs((int)pattern[k]));
    atomicMax(&s_cmax, c);
    __syncthreads();
    const int reach = (int)((float)s_cmax * 1.4143f) + 2;
    if (threadIdx.x < kLevels) {
      const int t = threadIdx.x, hh = t < P.nlev ? P.h[t] : 0;
      lvl_rows[2 * t] = min(max(kEdge - reach, 0), hh);
      lvl_rows[2 * t + 1] = max(min(hh - kEdge + reach, hh), 0);
    }
  }
  if (l >= P.nlev || i >= P.w[l] + P.h[l]) return;
  const bool isx = i < P.w[l];
  const int d = isx ? i : i - P.w[l];
  const int n0 = isx ? P.w[l - 1] : P.h[l - 1], n1 = isx ? P.w[l] : P.h[l];
  const ResizeTap t = resize_tap(d, (double)n0 / n1, n0);
  taps[P.toff[l] + i] = (uint32_t)t.i0 | ((uint32_t)t.w1 << 16);
}

// level l from level l - 1: a thread produces FOUR neighbouring output pixels of one row (grid: output row x image, one
// workgroup of up to 1024 threads per row, sized to the row: level 1 of a 1440-wide panorama is 300 quads = 5 waves, not
// two workgroups of 256 threads with the second one 83 % idle; the row's y tap is scalar).  The quads are laid out from the
// first DWORD-ALIGNED byte of the output row (level widths are rarely multiples of four, so most rows do not start on
// one): every interior quad is ONE dword store; its eight source columns span at most 7 bytes (scale 1.2), so each of
// the two source rows is ONE unaligned 8-byte load (clamped to stay inside the row) -- against 16 byte loads and 4 byte
// stores; rows narrower than 8 pixels and the (up to three) pixels before / after the aligned quads go pixel by pixel.
typedef unsigned long long __attribute__((aligned(1))) orb_u64_unaligned;
__global__ __launch_bounds__(1024) void resize_level_kernel(const uint8_t* __restrict__ src, long long src_stride, int h0,
                                                            int w0, uint8_t* __restrict__ dst, long long dst_stride, int h1,
                                                            int w1, const uint32_t* __restrict__ taps) {
  SOSVO_STREAMING_PRIO();
  const int dy = blockIdx.y, img = blockIdx.z;
  uint8_t* row_out = dst + (size_t)img * dst_stride + (size_t)dy * w1;
  const int lead = (int)((4 - ((uintptr_t)row_out & 3)) & 3);  // pixels before the first aligned dword of this row (uniform)
  const int q = blockIdx.x * blockDim.x + threadIdx.x;         // quad 0 = the leading pixels, quad k >= 1 starts at lead + 4 (k - 1)
  const int dx4 = q == 0 ? 0 : lead + 4 * (q - 1);
  const int cnt = q == 0 ? lead : 4;                           // pixels of this thread
  if (dx4 >= w1 || cnt == 0) return;
  const uint32_t ty = taps[w1 + dy];
  const int y0 = (int)(ty & 0xFFFFu), y1 = y0 + 1 < h0 ? y0 + 1 : h0 - 1;
  const uint32_t wy1 = ty >> 16, wy0 = 2048u - wy1;
  const uint8_t* s = src + (size_t)img * src_stride;
  const uint8_t* r0 = s + (size_t)y0 * w0;
  const uint8_t* r1 = s + (size_t)y1 * w0;
  uint8_t* out = row_out + dx4;
  // (wy0 * top + wy1 * bot + 2^21) >> 22 with top, bot <= 2048 * 255: 32-bit arithmetic is exact
  auto blend = [&](uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t wx1) {
    const uint32_t wx0 = 2048u - wx1;
    return ((wy0 * (wx0 * a0 + wx1 * a1) + wy1 * (wx0 * b0 + wx1 * b1)) + (1u << 21)) >> 22;
  };
  if (w0 >= 8 && q > 0 && dx4 + 3 < w1) {
    uint32_t t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = taps[dx4 + k];
    const int base = min((int)(t[0] & 0xFFFFu), w0 - 8);
    const unsigned long long va = *reinterpret_cast<const orb_u64_unaligned*>(r0 + base);
    const unsigned long long vb = *reinterpret_cast<const orb_u64_unaligned*>(r1 + base);
    uint32_t res = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x0 = (int)(t[k] & 0xFFFFu), x1 = x0 + 1 < w0 ? x0 + 1 : w0 - 1;
      const int o0 = 8 * (x0 - base), o1 = 8 * (x1 - base);
      res |= blend((uint32_t)(va >> o0) & 0xFFu, (uint32_t)(va >> o1) & 0xFFu, (uint32_t)(vb >> o0) & 0xFFu,
                   (uint32_t)(vb >> o1) & 0xFFu, t[k] >> 16)
             << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(out) = res;
    return;
  }
  for (int k = 0; k < cnt && dx4 + k < w1; ++k) {
    const uint32_t tx = taps[dx4 + k];
    const int x0 = (int)(tx & 0xFFFFu), x1 = x0 + 1 < w0 ? x0 + 1 : w0 - 1;
    out[k] = (uint8_t)blend(r0[x0], r0[x1], r1[x0], r1[x1], tx >> 16);
  }
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

// ---- FAST-9/16 score map (+ local-maximum flags) ----------------------------------------------------------------
// Neighbouring lane's value through the DPP operand path (a VALU move; 0 beyond the wave's ends: halo lanes only).
__device__ __forceinline__ int fs_from_left(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ int fs_from_right(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t min16(uint32_t a, uint32_t b) { return (uint32_t)min((uint16_t)a, (uint16_t)b); }
__device__ __forceinline__ uint32_t max16(uint32_t a, uint32_t b) { return (uint32_t)max((uint16_t)a, (uint16_t)b); }

// e[k] = 256 + circle pixel k - centre pixel (1 .. 511, so that 16-bit UNSIGNED minima / maxima -- the fast-rate VALU
// forms -- order them) -> corner score: the largest threshold for which the pixel is still a FAST-9 corner, minus 1
// (0: not a corner at `thr`): best = max(A - 256, 256 - B) with A = max over the 16 arcs of 9 consecutive circle pixels of
// the arc's minimum (brighter), B = min over the arcs of the arc's maximum (darker).
//
// max over the arcs of the arc's minimum, van Herk's way: the circle is two blocks of eight; an arc that starts at k inside
// a block is the block's suffix from k plus the k + 1 first values of the other block, so 4 x 7 running minima (two
// suffix, two prefix chains) + 16 + 15 instructions serve all 16 arcs (59; a doubling network needs 80).
__device__ __forceinline__ uint32_t fast_arc_max_of_min(const uint32_t (&g)[16]) {
  uint32_t suf0[8], pre1[8], suf1[8], pre0[8];
  suf0[7] = g[7];
  pre1[0] = g[8];
  suf1[7] = g[15];
  pre0[0] = g[0];
#pragma unroll
  for (int k = 6; k >= 0; --k) {
    suf0[k] = min16(g[k], suf0[k + 1]);
    suf1[k] = min16(g[8 + k], suf1[k + 1]);
  }
#pragma unroll
  for (int j = 1; j < 8; ++j) {
    pre1[j] = min16(g[8 + j], pre1[j - 1]);
    pre0[j] = min16(g[j], pre0[j - 1]);
  }
  uint32_t A = min16(suf0[0], pre1[0]);
  A = max16(A, min16(suf1[0], pre0[0]));
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    A = max16(A, min16(suf0[k], pre1[k]));
    A = max16(A, min16(suf1[k], pre0[k]));
  }
  return A;
}

// ONE polarity per lane, on the RAW circle values (round 4; rounds 1-3 biased all 16 values by 256 - centre and ran a brighter
// and a darker network on every pixel: 16 + 160 instructions).  Minima / maxima commute with the bias, so the network runs on
// the pixel values themselves and the centre is subtracted once at the end.  Any arc of 9 holds at least two of the four compass
// pixels (0, 4, 8, 12), so a pixel can only be a brighter corner at `thr` when the SECOND LARGEST of them exceeds centre + thr
// (cb) and a darker one when the second smallest is below centre - thr (cd) -- both from one 8-instruction sorting step; a side
// that cannot be a corner scores <= thr and never decides the result.  Darker lanes run the same network on 255 - v = v ^ 0xFF:
// best = A' - centre (brighter), A' - (255 - centre) (darker), i.e. A' - (centre ^ mask).  Lanes with BOTH (0.3 % of the pixels
// of a noisy panorama, some lane in ~12 % of a wave's rows) take a second pass with the other polarity, decided per wave.
// Equality with the two-sided network on biased values: tests/fast_network_check.cpp (plain C++, in the CPU suite) on random
// and adversarial circles, and every detector test.
// v[k]: circle pixel k (0 .. 255), c: the centre; valid: the lane's pixel can be a corner at all (interior column).
__device__ __forceinline__ int fast_score_raw(const uint32_t (&v)[16], int c, int thr, bool valid) {
  const uint32_t hiA = max16(v[0], v[4]), loA = min16(v[0], v[4]), hiB = max16(v[8], v[12]), loB = min16(v[8], v[12]);
  const uint32_t X = min16(hiA, hiB), Y = max16(loA, loB);
  const bool cb = valid && (int)max16(X, Y) > c + thr, cd = valid && (int)min16(X, Y) < c - thr;
  if (__ballot(cb || cd) == 0ULL) return 0;  // uniform: no lane of the wave can hold a corner (most rows of a blurred panorama)
  const uint32_t mm = (cd && !cb) ? 0xFFu : 0u;
  uint32_t g[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) g[k] = v[k] ^ mm;
  int best = (int)fast_arc_max_of_min(g) - (int)((uint32_t)c ^ mm);
  if (__ballot(cb && cd) != 0ULL) {  // uniform
#pragma unroll
    for (int k = 0; k < 16; ++k) g[k] = v[k] ^ 0xFFu;
    best = max(best, (int)fast_arc_max_of_min(g) - (int)((uint32_t)c ^ 0xFFu));
  }
  return ((cb || cd) && best > thr) ? best - 1 : 0;
}

// FAST-9/16 score map of nimg images that lie img_stride bytes apart (a dense batch, or one pyramid level), rows
// [r0, r1) only: a wave owns a 64-column strip (56 output columns, 4 halo columns each side) and walks down the rows
// with the last seven rows of its column in a register ring.  A row's value travels to the six neighbouring lanes that
// will need it (lane +-1, +-2, +-3: six chained DPP moves when the row arrives) and stays there for the seven steps the
// row lives, so the 16 circle pixels of every step are plain register reads.  3-pixel image border scores 0.
// flags (optional): bit = lane of a u64 per (row, strip): score > 0 and strictly greater than its 8 neighbours (the 3x3
// non-maximum suppression of the detectors), for rows (r0, r1 - 1); the halo of 4 makes the scores of an output lane's
// neighbours exact.
constexpr int kFsHalo = 4, kFsStripW = 64 - 2 * kFsHalo;
constexpr int kLp2Halo = 4, kLp2StripW = 128 - 2 * kLp2Halo;  // two pixels per lane (orb_level_pass_kernel)
__global__ __launch_bounds__(kThreads) void fast_score_rolling_kernel(const uint8_t* __restrict__ in, long long img_stride,
                                                                      int nimg, int rows, int cols, int strips, int thr,
                                                                      int r0, int r1, uint8_t* __restrict__ out,
                                                                      long long out_stride,
                                                                      unsigned long long* __restrict__ flags,
                                                                      long long flags_stride) {
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
  uint8_t* o = out + (size_t)img * out_stride;
  unsigned long long* fo = flags ? flags + (size_t)img * flags_stride : nullptr;
  // ring slot = (row - t_first) mod 7, compile-time after unrolling; per slot the centre value and its six shifted copies
  int vc[7], vl1[7], vl2[7], vl3[7], vr1[7], vr2[7], vr3[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) vc[k] = vl1[k] = vl2[k] = vl3[k] = vr1[k] = vr2[k] = vr3[k] = 0;
  int hm_a = 0, hm_b = 0, s_b = 0, lr_b = 0;  // NMS state: hm = max over (x-1, x, x+1) of rows y-2 / y-1, s / lr of row y-1
  const int t_first = r0 - 3, t_last = r1 - 1 + 3;
  auto src_row = [&](int t) { return min(max(t, 0), rows - 1); };  // clamped rows are never used as circle pixels
  // rows in two half-batches into the slots just emptied, three to six steps ahead of their use (as orb_level_pass_kernel)
  int c_pre[7];
  auto request = [&](int t0, int k0, int k1) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (k >= k0 && k < k1) c_pre[k] = (int)g[(uint32_t)(src_row(min(t0 + k, t_last)) * cols) + (uint32_t)xs];
  };
  request(t_first, 0, 7);
  auto step = [&](auto phase_tag, const int t) __attribute__((always_inline)) {
    constexpr int P = decltype(phase_tag)::value;
    if (t > t_last) return;  // uniform
    {
      const int c = c_pre[P];  // row t
      vc[P] = c;
      vr1[P] = fs_from_right(c);
      vr2[P] = fs_from_right(vr1[P]);
      vr3[P] = fs_from_right(vr2[P]);
      vl1[P] = fs_from_left(c);
      vl2[P] = fs_from_left(vl1[P]);
      vl3[P] = fs_from_left(vl2[P]);
    }
    const int y = t - 3;
    if (y < r0) return;  // uniform
    int s = 0;
    if (y >= 3 && y < rows - 3) {  // uniform
      // row y + dy lives in slot (P + 4 + dy) mod 7; circle pixel k = (rx[k], ry[k]) as in OpenCV's table
      constexpr int s3 = (P + 7) % 7, s2 = (P + 6) % 7, s1 = (P + 5) % 7, s0 = (P + 4) % 7, m1 = (P + 3) % 7, m2 = (P + 2) % 7,
                    m3 = (P + 1) % 7;
      const uint32_t v[16] = {
          (uint32_t)vc[s3],   // ( 0,  3)
          (uint32_t)vr1[s3],  // ( 1,  3)
          (uint32_t)vr2[s2],  // ( 2,  2)
          (uint32_t)vr3[s1],  // ( 3,  1)
          (uint32_t)vr3[s0],  // ( 3,  0)
          (uint32_t)vr3[m1],  // ( 3, -1)
          (uint32_t)vr2[m2],  // ( 2, -2)
          (uint32_t)vr1[m3],  // ( 1, -3)
          (uint32_t)vc[m3],   // ( 0, -3)
          (uint32_t)vl1[m3],  // (-1, -3)
          (uint32_t)vl2[m2],  // (-2, -2)
          (uint32_t)vl3[m1],  // (-3, -1)
          (uint32_t)vl3[s0],  // (-3,  0)
          (uint32_t)vl3[s1],  // (-3,  1)
          (uint32_t)vl2[s2],  // (-2,  2)
          (uint32_t)vl1[s3],  // (-1,  3)
      };
      s = fast_score_raw(v, vc[s0], thr, interior_x);
    }
    if (out_lane) o[(uint32_t)(y * cols) + (uint32_t)xc] = (uint8_t)s;
    if (fo) {  // uniform
      const int lr = max(fs_from_left(s), fs_from_right(s)), hm = max(lr, s);
      if (y - 1 > r0 && y - 1 < r1 - 1) {  // flag row y - 1 (uniform)
        const bool is_max = out_lane && s_b > 0 && s_b > lr_b && s_b > hm_a && s_b > hm;
        const unsigned long long bal = __ballot(is_max);
        if (lane == 0) fo[(uint32_t)((y - 1) * strips + strip)] = bal;
      }
      hm_a = hm_b;
      hm_b = hm;
      s_b = s;
      lr_b = lr;
    }
  };
  for (int t = t_first; t <= t_last; t += 7) {
    step(std::integral_constant<int, 0>{}, t);
    step(std::integral_constant<int, 1>{}, t + 1);
    step(std::integral_constant<int, 2>{}, t + 2);
    step(std::integral_constant<int, 3>{}, t + 3);
    request(t + 7, 0, 4);
    step(std::integral_constant<int, 4>{}, t + 4);
    step(std::integral_constant<int, 5>{}, t + 5);
    step(std::integral_constant<int, 6>{}, t + 6);
    request(t + 7, 4, 7);
  }
}

// rows [r0, r1) of the score map; flags (may be null) for rows (r0, r1 - 1)
static int32_t launch_fast_score(sosvo_ctx* ctx, const uint8_t* in, long long img_stride, int nimg, int rows, int cols, int thr,
                                 int r0, int r1, uint8_t* out, long long out_stride, unsigned long long* flags,
                                 long long flags_stride) {
  const int strips = cdiv(cols, kFsStripW);
  SOSVO_LAUNCH(ctx, fast_score_rolling_kernel, dim3(cdiv(nimg * strips, kThreads / 64)), dim3(kThreads), 0, ctx->stream, in,
               img_stride, nimg, rows, cols, strips, thr, r0, r1, out, out_stride, flags, flags_stride);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

// ---- one pass over a pyramid level: FAST flags + blur + next level (round 4) ---------------------------------------
// Rounds 1-3 read every level three times (resize -> HBM -> FAST score -> HBM, and again for the 7x7 blur) in 4 + 5 + 5
// launches per call; the three consumers want the same thing -- the last seven rows of a column strip and the three
// neighbours to each side -- which a rolling FAST kernel's register ring already holds.  Here the wave that walks a
// column strip of level l down its rows produces, from ONE read of the level:
//   * FAST-9 score -> 3x3 local-maximum flags (as fast_score_rolling_kernel), the score stored ONLY at the flagged pixels
//     (the selection reads nothing else of the map);
//   * the 7x7 Gaussian of the rows a descriptor can read (gauss7_kernel's integer arithmetic: a ring of horizontal sums;
//     mirrored columns in the halo lanes of the two edge strips, mirrored rows by walking the reflected row indices);
//   * level l + 1 (resize_level_kernel's arithmetic): output row dy is due when its lower source row arrives; an output
//     column belongs to the strip that owns its left tap x0(dx) and fetches its taps from the owning lanes (ds_bpermute).
// What a row t of the walk (t = -3 .. rows + 2: three mirrored rows each side for the blur) has to do is the same for every
// wave of the launch, so it comes from a table written once per call (orb_rowtab_kernel), held in LDS and read one row
// ahead: the first version evaluated ~100 scalar instructions of row-range tests per row and stalled on a dependent load of
// the next y tap after every output row -- as many issue slots as the arithmetic.  Steps of the kernel's history (ms per 256
// frame pairs, all five levels): separate kernels 3.23; one pass, one pixel per lane 3.02; row table 2.32; rows requested
// seven ahead 2.14; mirrored halo lanes instead of an edge-strip variant (113 VGPRs: four waves per SIMD) 1.84; TWO pixels
// per lane (below) 1.40, at four waves per SIMD 1.33.
constexpr uint32_t kRtEmit = 1u << 31, kRtTopSelf = 1u << 28;  // word 0: dy | wy1 << 16 | flags
constexpr uint32_t kRtHsum = 1u, kRtBlurOut = 2u, kRtFastRow = 4u, kRtScore = 8u, kRtFlagRow = 16u;  // word 1
struct LevelPass {
  const uint8_t* in;  // level l of image i at in + i * in_stride
  long long in_stride;
  int nimg, rows, cols, strips, thr;
  const uint2* rowtab;  // [rows + 6]: entry t + 3 for row t of the walk
  uint8_t* score;
  long long score_stride;
  unsigned long long* flags;
  long long flags_stride;
  uint8_t* blur;  // (rows per the table)
  long long blur_stride;
  uint8_t* next;  // level l + 1 (null: none), h1 x w1; xtaps = its w1 x taps
  long long next_stride;
  int w1;
  const uint32_t* xtaps;
};

__device__ __forceinline__ uint32_t orb_mad24(uint32_t k, uint32_t x, uint32_t acc) {  // (detect.hip's mad24)
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(k), "v"(acc));
  return r;
}
__device__ __forceinline__ uint32_t orb_mul24(uint32_t a, uint32_t b) {  // a, b < 2^24, product < 2^32: full-rate multiply
  uint32_t r;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// The walk's row table of every detection level.  grid = (cdiv(rows0 + 6, kThreads), ndet).  Per level l: entry t + 3,
//   word 0: the output row dy of level l + 1 whose LOWER source row is t (resize_tap: y1 = min(y0 + 1, h - 1); distinct for
//           distinct dy because h[l] > h[l + 1], which the host checks): kRtEmit | dy | wy1 << 16, kRtTopSelf when y0 == t;
//   word 1: kRtHsum (row t enters a blurred row), kRtBlurOut (blurred row t - 3 is due), kRtFastRow (FAST row y = t - 3 in
//           [fr0, fr1)), kRtScore (y has its 3-px border), kRtFlagRow (flags of row y - 1 are due).
// Blurred rows: all when `pattern` is given and restrict_rows is 0, run_orb_describe's lvl_rows when it is 1 (also written
// here), none without a pattern.
__global__ __launch_bounds__(kThreads) void orb_rowtab_kernel(Pyr P, const int8_t* __restrict__ pattern, int restrict_rows,
                                                              int tab_stride, uint2* __restrict__ rowtab,
                                                              int32_t* __restrict__ lvl_rows) {
  __shared__ int s_cmax;
  const int l = blockIdx.y, idx = blockIdx.x * kThreads + threadIdx.x;
  const int h = P.h[l];
  int br0 = 0, br1 = 0;
  if (pattern) {  // uniform
    if (threadIdx.x == 0) s_cmax = 0;
    __syncthreads();
    int c = 0;
    for (int k = threadIdx.x; k < 1024; k += kThreads) c = max(c, abs((int)pattern[k]));
    atomicMax(&s_cmax, c);
    __syncthreads();
    const int reach = (int)((float)s_cmax * 1.4143f) + 2;  // (as resize_taps_kernel / orb_describe_levels_kernel)
    br0 = restrict_rows ? min(max(kEdge - reach, 0), h) : 0;
    br1 = restrict_rows ? max(min(h - kEdge + reach, h), 0) : h;
    if (lvl_rows && blockIdx.x == 0 && threadIdx.x == 0) {
      lvl_rows[2 * l] = br0;
      lvl_rows[2 * l + 1] = br1;
    }
  }
  if (idx >= h + 6) return;
  const int t = idx - 3;
  const int fr0 = P.det[l] ? kEdge - 1 : 0, fr1 = P.det[l] ? h - kEdge + 1 : 0;
  uint32_t w0 = 0u, w1 = 0u;
  if (l + 1 < P.ndet && t >= 0 && t < h) {
    const int h1 = P.h[l + 1];
    const double scale = (double)h / h1;
    int dy = max((int)((double)(t - 2) / scale) - 1, 0);
    for (int k = 0; k < 8 && dy < h1; ++k, ++dy) {
      const ResizeTap tp = resize_tap(dy, scale, h);
      if (tp.i1 == t) {
        w0 = kRtEmit | (uint32_t)dy | ((uint32_t)tp.w1 << 16) | (tp.i0 == t ? kRtTopSelf : 0u);
        break;
      }
      if (tp.i1 > t) break;
    }
  }
  if (br1 > br0 && t >= br0 - 3 && t <= br1 + 2) w1 |= kRtHsum;
  if (br1 > br0 && t - 3 >= br0 && t - 3 < br1) w1 |= kRtBlurOut;
  const int y = t - 3;
  if (fr1 > fr0 && y >= fr0 && y < fr1) {
    w1 |= kRtFastRow;
    if (y >= 3 && y < h - 3) w1 |= kRtScore;
    if (y - 1 > fr0 && y - 1 < fr1 - 1) w1 |= kRtFlagRow;
  }
  rowtab[(size_t)l * tab_stride + idx] = make_uint2(w0, w1);
}

constexpr int kRowTabLds = 1024;  // rows + 6 entries of the level's table in LDS (higher images take the separate kernels)
// ---- two pixels per lane ---------------------------------------------------------------------------------------------------
// With one pixel per lane the pass was priced per wave-instruction (issue-bound at four waves per SIMD) and per byte-wide
// memory instruction: taken apart (compile-time variants), FAST alone cost 0.91 ms, blur + next level alone 0.79, of which the
// walk itself -- a byte load, six lane moves, byte stores per row -- ~0.4 each.  So a lane owns the pixel PAIR (x, x + 1) as the
// two 16-bit halves of one register: the FAST network runs on v_pk_min_u16 / v_pk_max_u16 (about the same instruction count per
// wave and row now covers 120 owned columns instead of 56), the polarity mask is one 32-bit xor for both halves, predicates
// are saturating packed subtractions (nonzero half = true), loads and blur stores are 16 bits wide.  Strip = 128 columns (120
// owned, halo 4); shifted copies of a row: the neighbour lanes' pairs by DPP, the odd shifts by v_alignbit of two of them.
// Flags: TWO words per (row, strip) -- bit b of word h is column strip * 120 - 4 + 2 b + h.  127 VGPRs: four waves per SIMD
// (with all seven shifted copies in rings and an unpacked ring of horizontal sums it was 144 and three waves: 1.40 ms against
// 1.33; forced to 128 by the compiler that form spilled 17 registers and ran 36 % slower).
typedef unsigned short orb_us2 __attribute__((ext_vector_type(2)));
typedef uint16_t __attribute__((aligned(1))) u16_unaligned;
__device__ __forceinline__ orb_us2 pk_u(uint32_t x) { return __builtin_bit_cast(orb_us2, x); }
__device__ __forceinline__ uint32_t pk_w(orb_us2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return pk_w(__builtin_elementwise_min(pk_u(a), pk_u(b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return pk_w(__builtin_elementwise_max(pk_u(a), pk_u(b))); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return pk_w(__builtin_elementwise_sub_sat(pk_u(a), pk_u(b))); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return pk_w(pk_u(a) + pk_u(b)); }
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, uint32_t b) { return pk_w(pk_u(a) * pk_u(b)); }
// nonzero half -> 0x00FF, zero half -> 0
__device__ __forceinline__ uint32_t pk_mask8(uint32_t d) { return pk_mul(pk_min(d, 0x00010001u), 0x00FF00FFu); }

__device__ __forceinline__ uint32_t fast_arc_max_of_min_pk(const uint32_t (&g)[16]) {  // fast_arc_max_of_min on both halves
  uint32_t suf0[8], pre1[8], suf1[8], pre0[8];
  suf0[7] = g[7];
  pre1[0] = g[8];
  suf1[7] = g[15];
  pre0[0] = g[0];
#pragma unroll
  for (int k = 6; k >= 0; --k) {
    suf0[k] = pk_min(g[k], suf0[k + 1]);
    suf1[k] = pk_min(g[8 + k], suf1[k + 1]);
  }
#pragma unroll
  for (int j = 1; j < 8; ++j) {
    pre1[j] = pk_min(g[8 + j], pre1[j - 1]);
    pre0[j] = pk_min(g[j], pre0[j - 1]);
  }
  uint32_t A = pk_min(suf0[0], pre1[0]);
  A = pk_max(A, pk_min(suf1[0], pre0[0]));
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    A = pk_max(A, pk_min(suf0[k], pre1[k]));
    A = pk_max(A, pk_min(suf1[k], pre0[k]));
  }
  return A;
}

// fast_score_raw on both halves: v[k], c: packed pixel pairs (0 .. 255 per half); thr_pk = thr | thr << 16 (thr <= 254),
// thrm1_pk likewise for thr - 1; valid_pk: 0x00FF per half whose pixel can be a corner.  -> packed scores.
__device__ __forceinline__ uint32_t fast_score_pk(const uint32_t (&v)[16], uint32_t c, uint32_t thr_pk, uint32_t thrm1_pk,
                                                  uint32_t valid_pk) {
  const uint32_t hiA = pk_max(v[0], v[4]), loA = pk_min(v[0], v[4]), hiB = pk_max(v[8], v[12]), loB = pk_min(v[8], v[12]);
  const uint32_t X = pk_min(hiA, hiB), Y = pk_max(loA, loB);
  // second largest compass pixel > c + thr / second smallest < c - thr, as saturating differences (nonzero = true)
  const uint32_t mB = pk_mask8(pk_subs(pk_max(X, Y), pk_add(c, thr_pk))) & valid_pk;
  const uint32_t mD = pk_mask8(pk_subs(c, pk_add(pk_min(X, Y), thr_pk))) & valid_pk;
  if (__ballot((mB | mD) != 0u) == 0ULL) return 0u;  // uniform
  const uint32_t mm = mD & ~mB;
  uint32_t g[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) g[k] = v[k] ^ mm;
  uint32_t t = pk_subs(fast_arc_max_of_min_pk(g), pk_add(c ^ mm, thr_pk));                 // best - thr where positive
  uint32_t s = pk_add(t, pk_mul(pk_min(t, 0x00010001u), thrm1_pk)) & (mB | mD);             // best - 1 where best > thr
  if (__ballot((mB & mD) != 0u) != 0ULL) {  // uniform: some half has two brighter AND two darker compass pixels
#pragma unroll
    for (int k = 0; k < 16; ++k) g[k] = v[k] ^ mD;   // the darker polarity wherever it can hold a corner
    t = pk_subs(fast_arc_max_of_min_pk(g), pk_add(c ^ mD, thr_pk));
    s = pk_max(s, pk_add(t, pk_mul(pk_min(t, 0x00010001u), thrm1_pk)) & mD);
  }
  return s;
}

__global__ __launch_bounds__(kThreads) void orb_level_pass_kernel(const LevelPass A) {
  SOSVO_STREAMING_PRIO();
  __shared__ uint2 s_tab[kRowTabLds];
  const int rows = A.rows, cols = A.cols, strips = A.strips;
  for (int i = threadIdx.x; i < rows + 6; i += kThreads) s_tab[i] = A.rowtab[i];  // (rows + 6 <= kRowTabLds: the host checks)
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * kThreads + threadIdx.x) >> 6));
  if (wave >= A.nimg * strips) return;  // wave-uniform (after the workgroup's only barrier)
  const int img = wave / strips, strip = wave - img * strips;
  const int xb = strip * kLp2StripW - kLp2Halo;
  const int x_lo = xb + 2 * lane, x_hi = x_lo + 1;
  // the pair's source: mirrored columns (reflect-101) are consecutive too, in descending order -> one 16-bit load, halves swapped
  auto mirror = [&](int x) { return min(max(x < 0 ? -x : (x >= cols ? 2 * (cols - 1) - x : x), 0), cols - 1); };
  const int m_lo = mirror(x_lo), m_hi = mirror(x_hi);
  const bool swapped = m_hi < m_lo;
  const int src_col = min(min(m_lo, m_hi), cols - 2);
  const uint32_t unpack_sel = swapped ? 0x0C000C01u : 0x0C010C00u;  // v_perm_b32: bytes (b0, 0, b1, 0) / (b1, 0, b0, 0)
  const bool own = lane >= kLp2Halo / 2 && lane < 64 - kLp2Halo / 2;
  const bool out_lo = own && x_lo < cols, out_hi = own && x_hi < cols;
  const uint32_t out_pk = (out_lo ? 0x0000FFFFu : 0u) | (out_hi ? 0xFFFF0000u : 0u);
  const uint32_t valid_pk = ((x_lo >= 3 && x_lo < cols - 3) ? 0x000000FFu : 0u) | ((x_hi >= 3 && x_hi < cols - 3) ? 0x00FF0000u : 0u);
  const uint32_t thr_pk = (uint32_t)A.thr * 0x00010001u, thrm1_pk = (uint32_t)(A.thr - 1) * 0x00010001u;
  const uint8_t* g = A.in + (size_t)img * A.in_stride;
  uint8_t* sc = A.score + (size_t)img * A.score_stride;
  unsigned long long* fo = A.flags + (size_t)img * A.flags_stride;
  uint8_t* bo = A.blur + (size_t)img * A.blur_stride;
  uint8_t* no = A.next + (size_t)img * A.next_stride;
  // next level: up to two output columns per lane (a strip's 120 columns map to <= 120 outputs: j = lane, 64 + lane)
  const int w1 = A.w1;
  int dxo[2] = {0, 0}, la[2] = {0, 0}, sh[2] = {0, 0};
  uint32_t wx1[2] = {0u, 0u};
  bool rz[2] = {false, false};
  if (A.next) {  // uniform
    const int xlo = strip * kLp2StripW;
    int d0 = max((int)((long long)xlo * w1 / cols) - 2, 0);
    while (d0 < w1 && (int)(A.xtaps[d0] & 0xFFFFu) < xlo) ++d0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      dxo[q] = d0 + 64 * q + lane;
      const uint32_t tx = A.xtaps[min(dxo[q], w1 - 1)];
      const int x0 = (int)(tx & 0xFFFFu);
      rz[q] = dxo[q] < w1 && x0 < xlo + kLp2StripW;
      const int rel = min(max(x0 - xb, 0), 125);
      la[q] = 4 * (rel >> 1);    // lane that holds the pair with x0
      sh[q] = 16 * (rel & 1);    // x0 is that pair's high half: the window (pair, next pair) >> 16
      wx1[q] = tx >> 16;
    }
  }
  // ring slot = (t + 3) mod 7; per slot the centre pair and its six shifted copies (packed)
  // (the even shifts -- the neighbour lanes' pairs themselves -- are NOT kept: one lane move from vc where FAST needs them, and
  // fourteen registers fewer: with the packed ring of horizontal sums 127 VGPRs, four waves per SIMD instead of three)
  uint32_t vc[7], vl1[7], vl3[7], vr1[7], vr3[7];
  uint32_t hsp[7];  // horizontal sums of the last seven rows, packed (<= 255 * 256 per half)
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    vc[k] = vl1[k] = vl3[k] = vr1[k] = vr3[k] = 0u;
    hsp[k] = 0u;
  }
  uint32_t hm_a = 0u, hm_b = 0u, s_b = 0u, lr_b = 0u;  // NMS state (packed), as the one-pixel form's
  uint32_t hx_prev[2] = {0u, 0u};
  const int t_last = rows + 2;
  auto src_row = [&](int t) { return min(abs(t), 2 * (rows - 1) - abs(t)); };
  uint32_t c_pre[7];
  auto request = [&](int t0, int k0, int k1) __attribute__((always_inline)) {  // rows t0 + k0 .. t0 + k1 - 1 -> their slots
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (k >= k0 && k < k1)
        c_pre[k] = (uint32_t)*reinterpret_cast<const u16_unaligned*>(g + (uint32_t)(src_row(min(t0 + k, t_last)) * cols) + (uint32_t)src_col);
  };
  request(-3, 0, 7);
  uint2 e_next = s_tab[0];
  auto step = [&](auto phase_tag, const int t) __attribute__((always_inline)) {
    constexpr int P = decltype(phase_tag)::value;
    if (t > t_last) return;  // uniform
    const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e_next.x);
    const uint32_t e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e_next.y);
    uint32_t row_nx, row_pv;  // row t's pairs of the neighbour lanes: (x+2, x+3), (x-2, x-1)
    {
      const uint32_t c = __builtin_amdgcn_perm(0u, c_pre[P], unpack_sel);  // row t: (x, x + 1) as 16-bit halves
      e_next = s_tab[min(t + 1, t_last) + 3];
      const uint32_t nx = (uint32_t)fs_from_right((int)c), nx2 = (uint32_t)fs_from_right((int)nx);   // (x+2, x+3), (x+4, x+5)
      const uint32_t pv = (uint32_t)fs_from_left((int)c), pv2 = (uint32_t)fs_from_left((int)pv);     // (x-2, x-1), (x-4, x-3)
      vc[P] = c;
      vr1[P] = __builtin_amdgcn_alignbit(nx, c, 16);    // (x+1, x+2)
      vr3[P] = __builtin_amdgcn_alignbit(nx2, nx, 16);  // (x+3, x+4)
      vl1[P] = __builtin_amdgcn_alignbit(c, pv, 16);    // (x-1, x)
      vl3[P] = __builtin_amdgcn_alignbit(pv, pv2, 16);  // (x-3, x-2)
      row_nx = nx;
      row_pv = pv;
    }
    // ---- next level ----
    if (A.next) {  // uniform
      uint32_t hx[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_ds_bpermute(la[q], (int)vc[P]);
        const uint32_t q1 = (uint32_t)__builtin_amdgcn_ds_bpermute(la[q], (int)row_nx);
        const uint32_t win = __builtin_amdgcn_alignbit(q1, q0, (uint32_t)sh[q]);  // (pixel x0, pixel x0 + 1)
        hx[q] = orb_mad24(wx1[q], win >> 16, orb_mul24(2048u - wx1[q], win & 0xFFFFu));
      }
      if (e0 & kRtEmit) {  // uniform
        const uint32_t wy1 = (e0 >> 16) & 0xFFFu, wy0 = 2048u - wy1, dy = e0 & 0xFFFFu;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint32_t top = (e0 & kRtTopSelf) ? hx[q] : hx_prev[q];
          const uint32_t res = (orb_mul24(wy0, top) + orb_mul24(wy1, hx[q]) + (1u << 21)) >> 22;
          if (rz[q]) no[dy * (uint32_t)w1 + (uint32_t)dxo[q]] = (uint8_t)res;
        }
      }
      hx_prev[0] = hx[0];
      hx_prev[1] = hx[1];
    }
    // ---- blur ----
    if (e1 & kRtHsum) {  // uniform
      // horizontal sums fit 16 bits (255 * 256): packed multiply-adds
      const orb_us2 hp = pk_u(vc[P]) * (unsigned short)54 + (pk_u(vl1[P]) + pk_u(vr1[P])) * (unsigned short)49 +
                         (pk_u(row_pv) + pk_u(row_nx)) * (unsigned short)34 + (pk_u(vl3[P]) + pk_u(vr3[P])) * (unsigned short)18;
      hsp[P] = pk_w(hp);
      if (e1 & kRtBlurOut) {  // uniform; rows t-6 .. t live in slots P+1 .. P+7 (mod 7)
        // (the ring stays packed -- seven registers, not fourteen --; the vertical sums need 32 bits per half)
        auto lo = [](uint32_t x) { return x & 0xFFFFu; };
        auto hi = [](uint32_t x) { return x >> 16; };
        const uint32_t h1 = hsp[(P + 1) % 7], h2 = hsp[(P + 2) % 7], h3 = hsp[(P + 3) % 7], h4 = hsp[(P + 4) % 7], h5 = hsp[(P + 5) % 7],
                       h6 = hsp[(P + 6) % 7], h7 = hsp[P];
        const uint32_t vl = orb_mad24(18u, lo(h1) + lo(h7), orb_mad24(34u, lo(h2) + lo(h6), orb_mad24(49u, lo(h3) + lo(h5), orb_mul24(54u, lo(h4)))));
        const uint32_t vh = orb_mad24(18u, hi(h1) + hi(h7), orb_mad24(34u, hi(h2) + hi(h6), orb_mad24(49u, hi(h3) + hi(h5), orb_mul24(54u, hi(h4)))));
        const uint32_t bl = (vl + 32768u) >> 16, bh = (vh + 32768u) >> 16;
        uint8_t* o = bo + (uint32_t)((t - 3) * cols) + (uint32_t)x_lo;
        if (out_hi) {
          *reinterpret_cast<u16_unaligned*>(o) = (uint16_t)(bl | (bh << 8));
        } else if (out_lo) {
          *o = (uint8_t)bl;
        }
      }
    }
    // ---- FAST score of row y = t - 3, flags (and the flagged pixels' scores) of row y - 1 ----
    if (!(e1 & kRtFastRow)) return;  // uniform
    const int y = t - 3;
    uint32_t s = 0u;
    if (e1 & kRtScore) {  // uniform
      constexpr int s3 = (P + 7) % 7, s2 = (P + 6) % 7, s1 = (P + 5) % 7, s0 = (P + 4) % 7, m1 = (P + 3) % 7, m2 = (P + 2) % 7,
                    m3 = (P + 1) % 7;
      const uint32_t r2s = (uint32_t)fs_from_right((int)vc[s2]), r2m = (uint32_t)fs_from_right((int)vc[m2]);  // (x+2, x+3) of rows y+2, y-2
      const uint32_t l2s = (uint32_t)fs_from_left((int)vc[s2]), l2m = (uint32_t)fs_from_left((int)vc[m2]);    // (x-2, x-1)
      const uint32_t v[16] = {vc[s3],  vr1[s3], r2s, vr3[s1], vr3[s0], vr3[m1], r2m, vr1[m3],
                              vc[m3],  vl1[m3], l2m, vl3[m1], vl3[s0], vl3[s1], l2s, vl1[s3]};
      s = fast_score_pk(v, vc[s0], thr_pk, thrm1_pk, valid_pk);
    }
    // packed scores of the left / right neighbours: (x-1, x) and (x+1, x+2)
    const uint32_t sn = (uint32_t)fs_from_right((int)s), sp = (uint32_t)fs_from_left((int)s);
    const uint32_t lr = pk_max(__builtin_amdgcn_alignbit(s, sp, 16), __builtin_amdgcn_alignbit(sn, s, 16)), hm = pk_max(lr, s);
    if (e1 & kRtFlagRow) {  // flag row y - 1 (uniform)
      // strictly greater than the eight neighbours (and hence > 0): saturating difference to their maximum, nonzero = flagged
      const uint32_t f = pk_subs(s_b, pk_max(pk_max(lr_b, hm_a), hm)) & out_pk;
      const bool f_lo = (f & 0xFFFFu) != 0u, f_hi = (f >> 16) != 0u;
      const unsigned long long bal_lo = __ballot(f_lo), bal_hi = __ballot(f_hi);
      if (lane == 0) {
        fo[(uint32_t)(((y - 1) * strips + strip) * 2)] = bal_lo;
        fo[(uint32_t)(((y - 1) * strips + strip) * 2 + 1)] = bal_hi;
      }
      if (f_lo) sc[(uint32_t)((y - 1) * cols) + (uint32_t)x_lo] = (uint8_t)s_b;
      if (f_hi) sc[(uint32_t)((y - 1) * cols) + (uint32_t)x_hi] = (uint8_t)(s_b >> 16);
    }
    hm_a = hm_b;
    hm_b = hm;
    s_b = s;
    lr_b = lr;
  };
  for (int t = -3; t <= t_last; t += 7) {
    step(std::integral_constant<int, 0>{}, t);
    step(std::integral_constant<int, 1>{}, t + 1);
    step(std::integral_constant<int, 2>{}, t + 2);
    step(std::integral_constant<int, 3>{}, t + 3);
    request(t + 7, 0, 4);
    step(std::integral_constant<int, 4>{}, t + 4);
    step(std::integral_constant<int, 5>{}, t + 5);
    step(std::integral_constant<int, 6>{}, t + 6);
    request(t + 7, 4, 7);
  }
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

// ---- AGAST (OAST 9/16) as a detector (feature_detection_method "AGAST") ---------------------------------------------
// cv2.AgastFeatureDetector_create() + setNonmaxSuppression(True) + detect(image, mask) (omnistereo/camera_models.py:
// 1670-1671, :1755; pose_est_tools.py:508-509): the 9-of-16 segment test by an optimal decision tree -- the corner set and
// the response of FAST-9/16 (the score map above, threshold 10) -- followed by AGAST's own suppression: corners that touch
// vertically or horizontally form blocks, a raster scan with a link table keeps one maximum per block (oracle/orb.c,
// orc_agast_nms, for the exact rule).  That scan is sequential by nature: one workgroup per image builds the corner list
// in raster order (all lanes), lane 0 walks it with everything in LDS (entry = pixel index | response << 24, link table,
// row starts: 132 KB), then all lanes compact the survivors per mask.
constexpr int kAgastCap = 16384, kAgastMaxRows = 1024;
__global__ __launch_bounds__(64) void agast_nms_collect_kernel(const uint8_t* __restrict__ score,
                                                               const uint32_t* __restrict__ mask_bits, int images_per_maskset,
                                                               int nmask, int rows, int cols, int cap, float* __restrict__ kp,
                                                               int32_t* __restrict__ n_out, int32_t* __restrict__ status) {
  __shared__ uint32_t ent[kAgastCap];
  __shared__ int32_t link[kAgastCap];
  __shared__ int32_t rowptr[kAgastMaxRows + 1];
  const int lane = threadIdx.x, img = blockIdx.x;
  const uint8_t* sc = score + (size_t)img * rows * cols;
  const uint32_t* mb = mask_bits + (size_t)(img / images_per_maskset) * rows * cols;
  // A. corners (score > 0) in raster order
  int total = 0;  // wave-uniform
  bool overflow = false;
  for (int y = 0; y < rows; ++y) {
    if (lane == 0) rowptr[y] = min(total, kAgastCap);
    for (int x0 = 0; x0 < cols; x0 += 64) {
      const int x = x0 + lane;
      const int sv = x < cols ? (int)sc[(size_t)y * cols + x] : 0;
      const unsigned long long bal = __ballot(sv > 0);
      const int pos = total + __popcll(bal & ((1ULL << lane) - 1ULL));
      if (sv > 0) {
        if (pos < kAgastCap) {
          ent[pos] = (uint32_t)(y * cols + x) | ((uint32_t)sv << 24);
          link[pos] = -1;
        } else {
          overflow = true;
        }
      }
      total += __popcll(bal);
    }
  }
  const int n = min(total, kAgastCap);
  if (lane == 0) rowptr[rows] = n;
  __syncthreads();
  // B. block maxima, sequentially (the pointer `ap` walks the previous row's corners: amortised O(1) per corner)
  if (lane == 0) {
    int row = -1, ap = 0, ap_end = 0;
    for (int cur = 0; cur < n; ++cur) {
      const uint32_t e = ent[cur];
      const int pix = (int)(e & 0xFFFFFFu), resp = (int)(e >> 24);
      const int y = pix / cols, x = pix - y * cols;
      if (y != row) {
        row = y;
        ap = y > 0 ? rowptr[y - 1] : 0;
        ap_end = y > 0 ? rowptr[y] : 0;
      }
      const int above_pix = pix - cols;
      while (ap < ap_end && (int)(ent[ap] & 0xFFFFFFu) < above_pix) ++ap;
      if (ap < ap_end && (int)(ent[ap] & 0xFFFFFFu) == above_pix) {
        int w = ap;
        while (link[w] != -1) w = link[w];
        if (resp < (int)(ent[w] >> 24)) link[cur] = w;
        else link[w] = cur;
      }
      if (x > 0 && cur > 0 && (int)(ent[cur - 1] & 0xFFFFFFu) == pix - 1) {
        int t = cur - 1;
        const int above = link[cur];
        while (link[t] != -1) t = link[t];
        if (above == -1) {
          if (t != cur) {
            if (resp < (int)(ent[t] >> 24)) link[cur] = t;
            else link[t] = cur;
          }
        } else if (t != above) {
          if ((int)(ent[above] >> 24) < (int)(ent[t] >> 24)) {
            link[above] = t;
            link[cur] = t;
          } else {
            link[t] = above;
            link[cur] = above;
          }
        }
      }
    }
  }
  __syncthreads();
  // C. survivors (link == -1) per mask, raster order
  for (int m = 0; m < nmask; ++m) {
    const int p = img * nmask + m;
    int cnt = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int c = c0 + lane;
      bool keep = false;
      int pix = 0;
      if (c < n && link[c] == -1) {
        pix = (int)(ent[c] & 0xFFFFFFu);
        keep = ((mb[pix] >> m) & 1u) != 0u;
      }
      const unsigned long long bal = __ballot(keep);
      const int pos = cnt + __popcll(bal & ((1ULL << lane) - 1ULL));
      if (keep && pos < cap) {
        const int y = pix / cols;
        kp[((size_t)p * cap + pos) * 2] = (float)(pix - y * cols);
        kp[((size_t)p * cap + pos) * 2 + 1] = (float)y;
      }
      cnt += __popcll(bal);
    }
    if (lane == 0) {
      n_out[p] = min(cnt, cap);
      if (status) status[p] = (cnt > cap ? 1 : 0) | ((__ballot(overflow) != 0ULL || total > kAgastCap) ? 2 : 0);
    }
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

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint32_t u32x3_unaligned __attribute__((ext_vector_type(3), aligned(1)));
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));

// Harris response of the 7x7 block around (cx, cy), k = 0.04, by ONE LANE (round 4; rounds 1-3 spent a whole wave on a
// candidate: 49 of 64 lanes busy, three 6-step cross-lane reductions and an LDS patch per candidate -- ~120 wave-instructions
// each, the largest share of this VALU-bound kernel).  The block's 3x3 Sobel pairs read the 9 x 9 pixels around the
// candidate: nine rows of three unaligned dwords (candidates lie >= 31 px inside the level, so no tap leaves the image),
// Sobel taken apart into its row pass -- hd = p[x+1] - p[x-1], hs = p[x-1] + 2 p[x] + p[x+1] -- and its column pass
// (Ix = hd above + 2 hd + hd below, Iy = hs below - hs above) on a three-row ring with compile-time slots: ~680
// instructions for the 64 candidates of a wave.  The sums are integers (exact, order-free); the float expression keeps the
// oracle's operation order.
__device__ __forceinline__ float harris_response_lane(const uint8_t* __restrict__ im, int w, int cx, int cy) {
  const uint8_t* p = im + (size_t)(cy - 4) * w + (cx - 4);
  int hd[3][7], hs[3][7];
  int a = 0, b = 0, c = 0;
#pragma unroll 1
  for (int r0 = 0; r0 < 9; r0 += 3) {  // three rows per trip: the ring slot of a row is compile-time; fully unrolled the 27
                                       // loads are hoisted and the kernel spills at the 128 VGPRs four workgroups per CU leave
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      const int r = r0 + sl;
      // (ONE 12-byte load per row: every lane has its own candidate, i.e. its own cache line, and the texture addresser takes a
      // wave-load's lines one per clock -- three dword loads per row were three such walks)
      const u32x3_unaligned q = *reinterpret_cast<const u32x3_unaligned*>(p + (size_t)r * w);
      const uint32_t d0 = q.x, d1 = q.y, d2 = q.z;
      const int px[9] = {(int)(d0 & 255u), (int)((d0 >> 8) & 255u), (int)((d0 >> 16) & 255u), (int)(d0 >> 24),
                         (int)(d1 & 255u), (int)((d1 >> 8) & 255u), (int)((d1 >> 16) & 255u), (int)(d1 >> 24), (int)(d2 & 255u)};
#pragma unroll
      for (int x = 0; x < 7; ++x) {
        hd[sl][x] = px[x + 2] - px[x];
        hs[sl][x] = px[x] + 2 * px[x + 1] + px[x + 2];
      }
      if (r >= 2) {  // (uniform) block row r - 2: rows r - 2 (above), r - 1, r (below) of the patch
        constexpr int kAbove[3] = {1, 2, 0}, kMid[3] = {2, 0, 1};
        const int t = kAbove[sl], m = kMid[sl];
#pragma unroll
        for (int x = 0; x < 7; ++x) {
          const int Ix = hd[t][x] + 2 * hd[m][x] + hd[sl][x];
          const int Iy = hs[sl][x] - hs[t][x];
          a += Ix * Ix;
          b += Iy * Iy;
          c += Ix * Iy;
        }
      }
    }
  }
  const float scale = 1.f / (4 * 7 * 255.f);
  const float s4 = (scale * scale) * (scale * scale);
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  return (((fa * fb) - (fc * fc)) - ((0.04f * (fa + fb)) * (fa + fb))) * s4;
}

// Intensity-centroid moments of the radius-15 disc around (cx, cy) by ONE LANE: patch row v = -15 .. 15 is eight unaligned
// dwords (columns cx - 15 .. cx + 16), the disc |u| <= umax(|v|) (OpenCV's: 15 15 15 15 14 14 14 13 13 12 11 10 9 8 6 3)
// becomes byte weights of v_dot4_u32_u8: 1 inside the disc for the row sum, u + 15 inside the disc for sum (u + 15) I -- two
// dot instructions per dword, the weights a constant table read through the SCALAR unit (the row index is uniform), a
// dword that misses the disc has weight 0.  m01 = sum v rowsum(v), m10 = sum (u + 15) I - 15 sum I.
// ~750 instructions for 64 keypoints (a wave per keypoint took ~150 each).  A real loop over the rows (two per trip):
// unrolled 31 times the 208 loads were all hoisted and the kernel ran at one wave per SIMD.
struct IcWeights {
  uint32_t w1[32][8], wu[32][8];  // (row 31: padding for the two-rows-per-trip loop, never read)
};
constexpr IcWeights make_ic_weights() {
  IcWeights t{};
  const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  for (int r = 0; r < 31; ++r) {
    const int v = r - 15, d = umax[v < 0 ? -v : v];
    for (int k = 0; k < 8; ++k) {
      uint32_t w1 = 0, wu = 0;
      for (int b = 0; b < 4; ++b) {
        const int u = 4 * k + b - 15;
        if ((u < 0 ? -u : u) <= d) {
          w1 |= 1u << (8 * b);
          wu |= (uint32_t)(u + 15) << (8 * b);
        }
      }
      t.w1[r][k] = w1;
      t.wu[r][k] = wu;
    }
  }
  return t;
}
__constant__ IcWeights kIcW = make_ic_weights();

// rows first, first + step, ... of the patch (the workgroup's waves share a keypoint's 31 rows: four dependent load rounds
// per wave instead of sixteen) -> partial sums: su = sum (u + 15) I, stot = sum I, m01 = sum v I over those rows
__device__ __forceinline__ void ic_moments_rows(const uint8_t* __restrict__ im, int w, int cx, int cy, int first, int step,
                                                uint32_t& su_out, uint32_t& stot_out, int& m01_out) {
  const uint8_t* p = im + (size_t)(cy - kHalfPatch) * w + (cx - kHalfPatch);
  uint32_t su = 0u, stot = 0u;
  int m01 = 0;
#pragma unroll 2
  for (int r = first; r < 2 * kHalfPatch + 1; r += step) {
    // (two 16-byte loads per row instead of eight dword loads: fewer requests; the time is the L2's -- every 32-byte row
    // segment of a patch pulls its own 128-byte line: 2 M keypoints x ~40 lines per launch)
    const u32x4_unaligned qa = *reinterpret_cast<const u32x4_unaligned*>(p + (size_t)r * w);
    const u32x4_unaligned qb = *reinterpret_cast<const u32x4_unaligned*>(p + (size_t)r * w + 16);
    const uint32_t x[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};  // (a dword outside the disc has weight 0)
    uint32_t rs = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      rs = __builtin_amdgcn_udot4(x[k], kIcW.w1[r][k], rs, false);
      su = __builtin_amdgcn_udot4(x[k], kIcW.wu[r][k], su, false);
    }
    m01 += (r - kHalfPatch) * (int)rs;
    stot += rs;
  }
  su_out = su;
  stot_out = stot;
  m01_out = m01;
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

#ifdef SOSVO_DEBUG_TIMING
// (debug build only, `make EXTRA=-DSOSVO_DEBUG_TIMING`: phase clocks of orb_select_kernel, 100 MHz ticks summed over workgroups)
__device__ unsigned long long g_sel_ticks[8];
#define SEL_TICK(k)                                                                      \
  do {                                                                                   \
    __syncthreads();                                                                     \
    if (tid == 0) {                                                                      \
      const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                  \
      atomicAdd(&g_sel_ticks[k], now_ - t_prev_);                                        \
      t_prev_ = now_;                                                                    \
    }                                                                                    \
  } while (0)
#else
#define SEL_TICK(k) do { } while (0)
#endif
// One workgroup per problem (image, mask) walks the detection levels.  Per level:
//   1. candidates: the local-maximum flags of the FAST score map (one u64 per row and 56-column strip, written by the
//      score kernel) inside the mask's bounding box and the 31-px border, mask bit set -> (position, score) in LDS;
//   2. retainBest(2 quota) by FAST score, ties kept: threshold from a 256-bin histogram, suffix sums by the first wave;
//   3. Harris response, one wave per candidate -> unique sort keys;  4. bitonic sort (response descending, then y, x);
//   5. retainBest(quota) by response, ties kept (counted in parallel: the list is sorted);
//   6. orientation, one wave per keypoint.
constexpr int kSelRegionBytes = 20 * 1024;
constexpr int kRankSortMax = 192;  // <= kThreads (one key per thread) and <= kCandMax / 2 (the sorted copy lives in `kept`)
__global__ __launch_bounds__(kThreads, 4) void orb_select_kernel(LevelSrc S, const uint8_t* __restrict__ score,
                                                              const unsigned long long* __restrict__ flags,
                                                              const uint32_t* __restrict__ mask_pyr,
                                                              const uint32_t* __restrict__ bbox, Pyr P,
                                                              int images_per_maskset, int nmask, int cap,
                                                              float* __restrict__ kp4, float* __restrict__ resp_out,
                                                              int32_t* __restrict__ n_out, int nimg_total, int flags2) {
  SOSVO_LATENCY_BOUND_PRIO();
  __shared__ uint32_t cxy[kCandMax];            // (y << 16) | x; after the sort: the sorted positions
  __shared__ uint8_t cfast[kCandMax];
  __shared__ unsigned long long ckey[kCandMax];  // ordered(harris) << 32 | (0xFFFFFFFF - linear index)
  __shared__ uint32_t kept[kCandMax];            // candidates that pass the FAST-score threshold (compacted)
  __shared__ int hist[256];
  __shared__ int ic_acc[192];                    // step 6: per keypoint of a 64-chunk, sum (u + 15) I | sum I | sum v I
  __shared__ uint32_t region32[kSelRegionBytes / 4];  // the level's pixels under the mask's candidates (steps 3 and 6)
  __shared__ int s_nc, s_thr, s_keep, s_nout, s_nk;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int p = orb_xcd_problem(nimg_total, nmask);
  if (p < 0) return;  // uniform (padding workgroup)
  const int img = p / nmask, m = p - img * nmask;
  if (tid == 0) s_nout = 0;
  __syncthreads();
#ifdef SOSVO_DEBUG_TIMING
  unsigned long long t_prev_ = __builtin_amdgcn_s_memrealtime();
#endif
  for (int l = 0; l < P.ndet; ++l) {
    if (!P.det[l]) continue;  // uniform
    const int h = P.h[l], w = P.w[l], quota = P.quota[l];
    const uint8_t* im = level_image(S, P, img, l);
    const uint8_t* sc = score + (size_t)img * P.total + P.off[l];
    const uint32_t* mk = mask_pyr + (size_t)(img / images_per_maskset) * P.total + P.off[l];
    const unsigned long long* fl = flags + (size_t)img * P.ftotal + P.foff[l];
    // flag words of a row: one per 56-column strip (bit b = column 56 s - 4 + b), or -- flags2, orb_level_pass_kernel's --
    // two per 120-column strip (bit b of word h = column 120 s - 4 + 2 b + h)
    const int strips = flags2 ? 2 * P.fstrips2[l] : P.fstrips[l];
    if (tid == 0) {
      s_nc = 0;
      s_nk = 0;
      s_keep = 0;
    }
    for (int k = tid; k < 256; k += kThreads) hist[k] = 0;
    __syncthreads();
    // 1. candidates (unordered): flag words of the rows / strips that meet the mask's bounding box on this level
    const uint32_t* bb = bbox + (((size_t)(img / images_per_maskset) * kLevels + l) * 32 + m) * 4;
    const int bx0 = max(kEdge, (int)(0xFFFFFFFFu - bb[0])), by0 = max(kEdge, (int)(0xFFFFFFFFu - bb[1]));
    const int bx1 = min(w - kEdge, (int)bb[2]), by1 = min(h - kEdge, (int)bb[3]);  // exclusive
    const int rw = bb[2] ? max(0, bx1 - bx0) : 0, rh = bb[2] ? max(0, by1 - by0) : 0;
    // The rectangle of the level every Harris block and every orientation patch of this mask lies in -- the candidates keep to
    // [bx0, bx1) x [by0, by1), a patch reaches 15 px (16 to the right: rows are read as 32 bytes) -- is requested NOW, into
    // registers (<= 20 dwords a thread, coalesced), and goes to LDS behind step 2: steps 3 and 6 then read LDS instead of
    // sending 27 + 248 scattered dwords per candidate / keypoint through the texture addresser (70 % + 21 % of the kernel's
    // 702 M accesses, round 4's counter pass).  A rectangle beyond the LDS area (one mask over a wide image): memory, as before.
    constexpr int kRegPer = kSelRegionBytes / 4 / kThreads;
    const int gx0 = bx0 - kHalfPatch, gy0 = by0 - kHalfPatch;
    const int gw = rw > 0 ? rw + 2 * kHalfPatch + 1 : 0, gh = rh > 0 ? rh + 2 * kHalfPatch : 0;
    const int gldw = (gw + 3) >> 2, gpitch_dw = gldw | 1, gpitch_b = 4 * gpitch_dw;
    const bool staged = gw > 0 && gh > 0 && gpitch_b * gh <= kSelRegionBytes;  // uniform (a mask without rows inside the border: gh 0)
    uint32_t rv[kRegPer];
    int rdst[kRegPer];
    if (staged) {
      const int step_r = kThreads / gldw, step_k = kThreads - step_r * gldw;
      int ry = tid / gldw, k = tid - ry * gldw;
      const uint8_t* org = im + (size_t)gy0 * w + gx0;
#pragma unroll
      for (int u = 0; u < kRegPer; ++u) {
        rdst[u] = ry < gh ? ry * gpitch_dw + k : -1;
        rv[u] = *reinterpret_cast<const u32_unaligned*>(org + (uint32_t)(min(ry, gh - 1) * w + 4 * k));
        k += step_k;
        ry += step_r;
        if (k >= gldw) {
          k -= gldw;
          ++ry;
        }
      }
    }
    const uint8_t* region8 = reinterpret_cast<const uint8_t*>(region32);
    // (the walk over the flag words; visit(x, y, FAST score) for every local maximum inside the mask)
    // Memory round trips are what this phase costs (a level's flag words, mask words and scores come from HBM: 128 images
    // of 577 KB share one XCD's 4 MB L2, ~2 - 3 us each under load), so the walk keeps as few of them in series as the data
    // allows: the flag words of TWO trips are requested together, and inside a word the mask word and the score of the NEXT
    // set bit are requested (both at once: neither needs the other) before the current bit's are looked at -- two loads deep
    // per word instead of flag -> mask -> score per bit, one word after the other.
    auto walk = [&](auto visit) {
      if (rw <= 0 || rh <= 0) return;
      // (word index inside a row: strip for the one-word layout, 2 strip + half for the two-word one; xstep = columns per bit)
      const int xstep = flags2 ? 2 : 1, sw = flags2 ? kLp2StripW : kFsStripW;
      const int st0 = (bx0 / sw) * xstep, ns = ((bx1 - 1) / sw) * xstep + (xstep - 1) - st0 + 1, nw = ns * rh;
      auto word = [&](int i, int& y, int& xbase) -> unsigned long long {
        const int ry = i / ns, sidx = st0 + (i - ry * ns);
        y = by0 + ry;
        xbase = flags2 ? (sidx >> 1) * kLp2StripW - kLp2Halo + (sidx & 1) : sidx * kFsStripW - kFsHalo;  // column of bit 0
        unsigned long long wd = fl[(size_t)y * strips + sidx];
        // keep the bits whose column xbase + xstep b lies inside [bx0, bx1)
        const int l_lo = max(0, (bx0 - xbase + xstep - 1) / xstep), l_hi = min(63, (bx1 - 1 - xbase) >= 0 ? (bx1 - 1 - xbase) / xstep : -1);
        return l_hi >= l_lo ? (wd >> l_lo << l_lo) & (~0ULL >> (63 - l_hi)) : 0ULL;
      };
      auto bits = [&](unsigned long long wd, int y, int xbase) __attribute__((always_inline)) {
        if (!wd) return;
        const uint32_t row = (uint32_t)y * (uint32_t)w;
        int x = xbase + xstep * (__ffsll((long long)wd) - 1);
        wd &= wd - 1ULL;
        uint32_t mword = mk[row + (uint32_t)x];
        int sval = (int)sc[row + (uint32_t)x];
        for (;;) {
          const int xc = x, sc_c = sval;
          const uint32_t mc = mword;
          const bool more = wd != 0ULL;
          if (more) {
            x = xbase + xstep * (__ffsll((long long)wd) - 1);
            wd &= wd - 1ULL;
            mword = mk[row + (uint32_t)x];
            sval = (int)sc[row + (uint32_t)x];
          }
          if ((mc >> m) & 1u) visit(xc, y, sc_c);
          if (!more) break;
        }
      };
      for (int i = tid; i < nw; i += 2 * kThreads) {
        int y0 = 0, xb0 = 0, y1 = 0, xb1 = 0;
        const unsigned long long w0 = word(i, y0, xb0);
        const unsigned long long w1 = i + kThreads < nw ? word(i + kThreads, y1, xb1) : 0ULL;
        bits(w0, y0, xb0);
        bits(w1, y1, xb1);
      }
    };
    walk([&](int x, int y, int sv) {
      atomicAdd(&hist[sv], 1);  // EVERY candidate counts for the threshold of step 2
      const int slot = atomicAdd(&s_nc, 1);
      if (slot < kCandMax) {
        cxy[slot] = ((uint32_t)y << 16) | (uint32_t)x;
        cfast[slot] = (uint8_t)sv;
      }
    });
    __syncthreads();
    SEL_TICK(0);
    const int nc_all = s_nc;
    if (nc_all == 0) continue;  // uniform: nothing on this level (s_nc is re-initialised behind the barrier of the next level's step 1)
    // 2. retainBest(2 * quota) by FAST score, ties kept: the largest score t with #(score >= t) >= 2 quota
    if (wid == 0) {
      int thr = 0;
      if (nc_all > 2 * quota) {  // uniform
        const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        int suf = h0 + h1 + h2 + h3;  // -> sum over the bins of lanes >= lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_down(suf, o);
          if (lane + o < 64) suf += v;
        }
        const int a3 = suf - h0 - h1 - h2, a2 = a3 + h2, a1 = a2 + h1, a0 = suf;  // #(score >= 4 lane + k)
        int best = -1;
        if (a0 >= 2 * quota) best = 4 * lane;
        if (a1 >= 2 * quota) best = 4 * lane + 1;
        if (a2 >= 2 * quota) best = 4 * lane + 2;
        if (a3 >= 2 * quota) best = 4 * lane + 3;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
        thr = best;
      }
      if (lane == 0) s_thr = thr;
    }
    __syncthreads();
    const int thr = s_thr;
    if (nc_all <= kCandMax) {  // uniform
      for (int i = tid; i < nc_all; i += kThreads)
        if (cfast[i] >= thr) kept[atomicAdd(&s_nk, 1)] = cxy[i];  // order irrelevant: the sort keys are unique
    } else {
      // more local maxima than the LDS list holds (a noisy level-0 image: thousands): the threshold above is exact (the
      // histogram saw them all), so a second walk keeps exactly the retained ones.  Only a retained set beyond kCandMax --
      // thousands of candidates TIED at the threshold score -- is cut (an unspecified subset of it goes on).
      walk([&](int x, int y, int sv) {
        if (sv >= thr) {
          const int slot = atomicAdd(&s_nk, 1);
          if (slot < kCandMax) kept[slot] = ((uint32_t)y << 16) | (uint32_t)x;
        }
      });
    }
    if (staged) {
#pragma unroll
      for (int u = 0; u < kRegPer; ++u)
        if (rdst[u] >= 0) region32[rdst[u]] = rv[u];
    }
    __syncthreads();
    SEL_TICK(1);
    const int nc = min(s_nk, kCandMax);
    // 3. Harris response of the survivors -> sort keys: a lane per candidate, the candidates packed into as few waves as
    // they fill (a wave's instruction count does not depend on how many of its lanes work)
    for (int j0 = wid * 64; j0 < nc; j0 += kThreads) {
      const int j = j0 + lane;
      const uint32_t xy = kept[min(j, nc - 1)];
      const float r = staged ? harris_response_lane(region8, gpitch_b, (int)(xy & 0xFFFFu) - gx0, (int)(xy >> 16) - gy0)
                             : harris_response_lane(im, w, (int)(xy & 0xFFFFu), (int)(xy >> 16));
      if (j < nc) {
        const uint32_t lin = (xy >> 16) * (uint32_t)w + (xy & 0xFFFFu);
        ckey[j] = ((unsigned long long)sosvo_float_ordered(r) << 32) | (0xFFFFFFFFu - lin);
      }
    }
    // 4. sort, descending: response, then (y, x) ascending (keys are unique; padding sorts last).  Up to 64 keys (the usual
    // case on a blurred panorama): one wave, in registers, no barriers; more: bitonic sort in LDS by the workgroup.
    __syncthreads();
    SEL_TICK(2);
    if (nc <= 64) {
      if (wid == 0) {
        unsigned long long key = lane < nc ? ckey[lane] : 0ULL;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
          for (int j = k >> 1; j > 0; j >>= 1) {
            const unsigned long long other = __shfl_xor(key, j);
            const bool desc = (lane & k) == 0, lower = (lane & j) == 0;
            // descending block: the lower lane keeps the larger key
            const bool take_max = desc == lower;
            key = take_max ? (key > other ? key : other) : (key < other ? key : other);
          }
        }
        ckey[lane] = key;
      }
      __syncthreads();
    } else if (nc <= kRankSortMax) {
      // up to 192 keys (the usual case on the three largest levels: 2 n_l = 70 .. 100 candidates + ties): every thread RANKS
      // its key against all the others (broadcast LDS reads, unique keys -> the ranks are a permutation) and writes it to its
      // place -- three barriers instead of the bitonic network's 28 (a level's sort was ~8 us of barriers with one key per
      // two threads: variants of the kernel without the phases behind the walk showed the selection spending 2/3 of its time
      // in these low-parallelism phases, not in memory)
      unsigned long long* tmp = reinterpret_cast<unsigned long long*>(kept);  // (the candidate list is dead behind step 3)
      const unsigned long long mine = tid < nc ? ckey[tid] : 0ULL;
      int rank = 0;
      for (int i = 0; i < nc; ++i) rank += ckey[i] > mine ? 1 : 0;
      if (tid < nc) tmp[rank] = mine;
      __syncthreads();
      if (tid < nc) ckey[tid] = tmp[tid];
      __syncthreads();
    } else {
      int N = 128;
      while (N < nc) N <<= 1;
      for (int i = nc + tid; i < N; i += kThreads) ckey[i] = 0ULL;
      __syncthreads();
      for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int t = tid; t < N / 2; t += kThreads) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
            const unsigned long long a = ckey[i], b = ckey[ixj];
            const bool desc = (i & k) == 0;
            if ((a < b) == desc) {
              ckey[i] = b;
              ckey[ixj] = a;
            }
          }
          __syncthreads();
        }
      }
    }
    SEL_TICK(3);
    // 5. retainBest(quota) by Harris response, ties kept: quota + the following entries that equal the quota-th response
    if (nc > quota) {  // uniform
      const uint32_t amb = (uint32_t)(ckey[quota - 1] >> 32);  // ordered(float): same order as the responses
      const float amb_f = sosvo_ordered_float(amb);
      int more = 0;
      for (int j = quota + tid; j < nc; j += kThreads) more += sosvo_ordered_float((uint32_t)(ckey[j] >> 32)) >= amb_f;
      if (more) atomicAdd(&s_keep, more);
    }
    __syncthreads();
    const int keep = nc > quota ? quota + s_keep : nc, base_out = s_nout;
    // 6. orientation: a lane per keypoint, the 31 patch rows dealt over the workgroup's waves (partial sums meet in LDS)
    const int nor = min(keep, cap - base_out);
    for (int j0 = 0; j0 < nor; j0 += 64) {  // (uniform; one trip unless a level keeps more than 64)
      if (tid < 192) ic_acc[tid] = 0;
      __syncthreads();
      const int j = j0 + lane;
      const unsigned long long key = ckey[min(j, nor - 1)];
      const uint32_t lin = 0xFFFFFFFFu - (uint32_t)key;
      const int cy = (int)(lin / (uint32_t)w), cx = (int)(lin - (uint32_t)cy * (uint32_t)w);
      uint32_t su, stot;
      int m01p;
      if (staged)
        ic_moments_rows(region8, gpitch_b, cx - gx0, cy - gy0, __builtin_amdgcn_readfirstlane(wid), kThreads / 64, su, stot, m01p);
      else
        ic_moments_rows(im, w, cx, cy, __builtin_amdgcn_readfirstlane(wid), kThreads / 64, su, stot, m01p);
      atomicAdd(&ic_acc[lane], (int)su);
      atomicAdd(&ic_acc[64 + lane], (int)stot);
      atomicAdd(&ic_acc[128 + lane], m01p);
      __syncthreads();
      if (wid == 0 && j < nor) {
        const int m10 = ic_acc[lane] - kHalfPatch * ic_acc[64 + lane], m01 = ic_acc[128 + lane];
        const size_t o = (size_t)p * cap + base_out + j;
        kp4[4 * o + 0] = (float)cx * P.scale[l];
        kp4[4 * o + 1] = (float)cy * P.scale[l];
        kp4[4 * o + 2] = fast_atan2_deg((float)m01, (float)m10);
        kp4[4 * o + 3] = (float)l;
        resp_out[o] = sosvo_ordered_float((uint32_t)(key >> 32));
      }
      __syncthreads();
    }
    __syncthreads();
    if (tid == 0) s_nout = min(cap, base_out + keep);
    __syncthreads();
    SEL_TICK(4);
  }
  if (tid == 0) n_out[p] = s_nout;
}

// ---- descriptors of oriented multi-level keypoints ------------------------------------------------------------
// The 512 test points of a keypoint are scattered bytes inside the (2R + 1)^2 patch around it (R = 19 for OpenCV's table at
// any angle).  Read straight from the blurred level they are 8 gathers of 64 unrelated bytes per keypoint, which the
// texture-address unit serialises lane by lane (measured, round 4: 1.67 ms per 256 pairs, the same before and after the
// kernel's VALU instructions were halved).  So the patch goes through LDS: PR rows of pdw unaligned dwords (7 loads per lane
// for R = 19, neighbouring lanes on neighbouring dwords), row pitch ODD in dwords so that the rows spread over all banks, and
// the tests read LDS bytes.  Keypoints too close to their level's border for the patch (none the detector returns: it
// keeps 31 px) take the direct path with reflected coordinates.
constexpr int kDescPatchMaxR = 23, kDescPatchRows = 2 * kDescPatchMaxR + 1, kDescPatchPitch = 13, kDescPatchLoads = 9;
constexpr int kDescRegionBytes = 24 * 1024;
__global__ __launch_bounds__(kThreads) void orb_describe_levels_kernel(const uint8_t* __restrict__ blur, Pyr P, int nlev_have,
                                                                       int rows, int cols, int nmask, int cap,
                                                                       float* __restrict__ kp4, int32_t* __restrict__ n_io,
                                                                       const int8_t* __restrict__ pattern,
                                                                       uint8_t* __restrict__ desc,
                                                                       float* __restrict__ kp_xy, int nimg_total) {
  SOSVO_LATENCY_BOUND_PRIO();
  extern __shared__ float lds_kp[];  // [cap][4] keypoints, then [cap][2] (cos, sin) of their angles
  float* lds_cs = lds_kp + 4 * (size_t)cap;
  __shared__ int8_t spat[1024];
  // one LDS area, two uses: the REGION of a level's blurred image under all of this problem's keypoints of that level (the
  // usual case, below), or the four waves' per-keypoint patches (keypoints whose region does not fit)
  __shared__ uint32_t region32[kDescRegionBytes / 4];
  static_assert(kDescRegionBytes >= (int)sizeof(uint32_t) * (kThreads / 64) * kDescPatchRows * kDescPatchPitch, "patches fit");
  __shared__ int s_bb[kLevels][6];  // per level, over its keypoints with a whole patch: xmin, ymin, xmax, ymax (level px), first / last index
  __shared__ int s_mode[kLevels];   // 1: the level's keypoints are described from the region
  __shared__ int s_rest;            // keypoints that are not (1: the per-keypoint paths have work)
  __shared__ int wave_off[5];
  __shared__ int s_running;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int p = orb_xcd_problem(nimg_total, nmask);
  if (p < 0) return;  // uniform (padding workgroup)
  const int img = p / nmask;
  const int n = min(n_io[p], cap);
  if (tid == 0) s_running = 0;
  for (int i = tid; i < 1024; i += kThreads) spat[i] = pattern[i];
  // the levels' geometry in LDS: indexed by a keypoint's level inside the descriptor loop, where a look-up in the by-value
  // struct is a global load from the kernel arguments whose wait would drain the patch loads in flight
  __shared__ int s_lh[kLevels], s_lw[kLevels], s_loff[kLevels];
  __shared__ float s_linv[kLevels];
  if (tid < kLevels) {
    s_lh[tid] = P.h[tid];
    s_lw[tid] = P.w[tid];
    s_loff[tid] = (int)P.off[tid];  // (< 2^28 * 3: rows * cols < 2^28)
    s_linv[tid] = 1.f / P.scale[tid];
  }
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
             l >= 0 && l < nlev_have;  // (levels whose blurred image exists)
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
  // cos / sin of every kept keypoint's angle, a LANE per keypoint (trig_core.h's double-precision sincos, the oracle's bits,
  // rounded once to float): evaluated by a whole wave per keypoint this was ~1/4 of the kernel's instructions
  for (int i = tid; i < mkept; i += kThreads) {
    float angle = lds_kp[4 * i + 2];
    angle *= (float)(3.14159265358979323846 / 180.0);
    double sd, cd;
    sv_sincos((double)angle, &sd, &cd);
    lds_cs[2 * i] = (float)cd;
    lds_cs[2 * i + 1] = (float)sd;
  }
  __syncthreads();
  if (kp_xy)
    for (int i = tid; i < mkept; i += kThreads) {
      kp_xy[((size_t)p * cap + i) * 2] = lds_kp[4 * i];
      kp_xy[((size_t)p * cap + i) * 2 + 1] = lds_kp[4 * i + 1];
    }
  if (tid == 0) n_io[p] = mkept;
  // this lane's four tests (lane, 64 + lane, 128 + lane, 192 + lane): their eight pattern points never change
  float ppx[4][2], ppy[4][2];
  int cmax = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = 2 * (64 * r + lane) + e;
      const int ix = spat[2 * i], iy = spat[2 * i + 1];
      ppx[r][e] = (float)ix;
      ppy[r][e] = (float)iy;
      cmax = max(cmax, max(abs(ix), abs(iy)));
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
  const int margin = (int)((float)cmax * 1.4143f) + 2;  // a rotated, rounded test point stays within it (13 -> 20)
  const int R = margin - 1, PR = 2 * R + 1, pdw = (PR + 3) >> 2, pstride = pdw | 1;
  const bool patch_ok = R <= kDescPatchMaxR;  // (|coordinate| <= 16: R <= 23)
  // this lane's dwords of a patch: item i = it * 64 + lane -> (row i / pdw, dword i % pdw); the same for every keypoint
  int prow[kDescPatchLoads], pk[kDescPatchLoads];
#pragma unroll
  for (int it = 0; it < kDescPatchLoads; ++it) {
    const int i = min(it * 64 + lane, PR * pdw - 1);  // (lanes past the end repeat the last item: same value, same word)
    prow[it] = i / pdw;
    pk[it] = i - prow[it] * pdw;
  }
  const int nit = patch_ok ? (PR * pdw + 63) >> 6 : 0;  // <= kDescPatchLoads
  uint32_t* patch32 = region32 + wid * (kDescPatchRows * kDescPatchPitch);
  const uint8_t* patch8 = reinterpret_cast<const uint8_t*>(patch32);
  // A keypoint's state: where its patch lies and the patch dwords themselves, REQUESTED ONE KEYPOINT AHEAD: a wave used to
  // load, wait a full memory round trip, stage and test one keypoint after the other (round 4's stats pass: 2.8 us per keypoint
  // and wave for ~150 instructions); now the next keypoint's loads fly while this one's tests run.  For the waits to count
  // loads instead of draining them, the staged loop holds no other global load and a fixed number of patch loads per keypoint
  // (NIT = 7 for OpenCV's table, 9 at most; a keypoint that is not staged -- none the detector returns -- loads a patch that
  // certainly exists, the top-left corner of its image's level 0, and is described by the second loop below).
  struct KpState {
    int cx, cy, hh, ww, l;
    float ca, sa;
    bool staged, region;
  };
  // (a keypoint belongs to a WAVE: its level, centre and image pointers are told to be scalars, so that the patch loads take a
  // scalar base plus one 32-bit offset per lane instead of seven 64-bit address chains)
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  auto locate = [&](int j, const uint8_t*& im, const uint8_t*& ctr, bool& inside) __attribute__((always_inline)) {
    KpState k;
    const int l = __builtin_amdgcn_readfirstlane((int)lds_kp[4 * j + 3]);
    k.hh = __builtin_amdgcn_readfirstlane(s_lh[l]);
    k.ww = __builtin_amdgcn_readfirstlane(s_lw[l]);
    const float inv = s_linv[l];
    k.ca = lds_cs[2 * j];
    k.sa = lds_cs[2 * j + 1];
    k.cx = __builtin_amdgcn_readfirstlane(__float2int_rn(lds_kp[4 * j] * inv));
    k.cy = __builtin_amdgcn_readfirstlane(__float2int_rn(lds_kp[4 * j + 1] * inv));
    im = blur + (size_t)img * P.total + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane(s_loff[l]);
    // a keypoint at least `margin` px inside its level image (every keypoint the detector returns: it keeps 31 px, and
    // OpenCV's table needs 20) needs no border reflection -- a wave-uniform test
    inside = k.cx >= margin && k.cy >= margin && k.cx < k.ww - margin && k.cy < k.hh - margin;
    ctr = im + (size_t)k.cy * k.ww + k.cx;
    k.l = l;
    k.region = inside && s_mode[l] != 0;              // wave-uniform
    k.staged = inside && patch_ok && !k.region;       // wave-uniform
    return k;
  };
  // ---- the usual case: a level's keypoints from ONE staged region ------------------------------------------------------
  // Round 4's counters: every 40-byte row of a keypoint's patch pulls its own 128-byte line through the L2 (~45 lines, 6 KB per
  // keypoint, 12 GB per 256 frame pairs: the kernel ran at the L2's line rate, not at its 150 instructions per keypoint) -- and
  // the ~165 keypoints of a mask lie so close together that their patches cover the same pixels nine times over.  So: per
  // level, the bounding box of the problem's keypoints grown by the pattern's reach goes to LDS once (coalesced dword loads,
  // ~20 KB for a 120-column mask of a 146-row panorama) and every test of every keypoint of that level is an LDS byte read.
  // A level whose region exceeds the LDS area (one mask over a whole wide image) keeps the per-keypoint patches below.
  if (tid < kLevels) {
    s_bb[tid][0] = s_bb[tid][1] = s_bb[tid][4] = 0x7FFFFFFF;
    s_bb[tid][2] = s_bb[tid][3] = s_bb[tid][5] = -1;
    s_mode[tid] = 0;
  }
  if (tid == 0) s_rest = 0;
  __syncthreads();
  for (int i = tid; i < mkept; i += kThreads) {
    const int l = (int)lds_kp[4 * i + 3];
    const int cx = __float2int_rn(lds_kp[4 * i] * s_linv[l]), cy = __float2int_rn(lds_kp[4 * i + 1] * s_linv[l]);
    if (cx >= margin && cy >= margin && cx < s_lw[l] - margin && cy < s_lh[l] - margin) {
      atomicMin(&s_bb[l][0], cx);
      atomicMin(&s_bb[l][1], cy);
      atomicMax(&s_bb[l][2], cx);
      atomicMax(&s_bb[l][3], cy);
      atomicMin(&s_bb[l][4], i);
      atomicMax(&s_bb[l][5], i);
    } else {
      s_rest = 1;
    }
  }
  __syncthreads();
  const uint8_t* region8 = reinterpret_cast<const uint8_t*>(region32);
  for (int l = 0; l < nlev_have; ++l) {  // (uniform)
    const int xmin = s_bb[l][0], ymin = s_bb[l][1], xmax = s_bb[l][2], ymax = s_bb[l][3], jlo = s_bb[l][4], jhi = s_bb[l][5];
    if (xmax < 0) continue;  // no keypoint with a whole patch on this level
    const int rx0 = xmin - R, ry0 = ymin - R, rw = xmax - xmin + 2 * R + 1, rh = ymax - ymin + 2 * R + 1;
    const int ldw = (rw + 3) >> 2, pitch_dw = ldw | 1;  // (an odd pitch in dwords: a patch column spreads over the banks)
    if (pitch_dw * 4 * rh > kDescRegionBytes) {
      if (tid == 0) s_rest = 1;
      continue;
    }
    const uint8_t* im = blur + (size_t)img * P.total + (size_t)(uint32_t)s_loff[l];
    const int ww = s_lw[l];
    __syncthreads();  // (the previous level's tests are done with the area)
    {  // dword i = tid, tid + 256, ... of the region (row i / ldw, dword i % ldw), EIGHT requests in flight per thread and the
       // position advanced without a division (the straightforward loop compiled to four loads per trip around ~120
       // instructions of division emulation: five dependent round trips for a 20 KB region)
      const int step_r = kThreads / ldw, step_k = kThreads - step_r * ldw;
      int ry = tid / ldw, k = tid - ry * ldw;
      const uint8_t* org = im + (size_t)ry0 * ww + rx0;
      while (__any(ry < rh)) {
        uint32_t v[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          dst[u] = ry < rh ? ry * pitch_dw + k : -1;
          v[u] = *reinterpret_cast<const u32_unaligned*>(org + (uint32_t)(min(ry, rh - 1) * ww + 4 * k));
          k += step_k;
          ry += step_r;
          if (k >= ldw) {
            k -= ldw;
            ++ry;
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (dst[u] >= 0) region32[dst[u]] = v[u];
      }
    }
    if (tid == 0) s_mode[l] = 1;
    __syncthreads();
    const uint32_t pitch_b = 4u * (uint32_t)pitch_dw;
    for (int j = jlo + wid_s; j <= jhi; j += kThreads / 64) {
      if (__builtin_amdgcn_readfirstlane((int)lds_kp[4 * j + 3]) != l) continue;  // (keypoints handed in out of level order)
      const float inv = s_linv[l];
      const int cx = __builtin_amdgcn_readfirstlane(__float2int_rn(lds_kp[4 * j] * inv));
      const int cy = __builtin_amdgcn_readfirstlane(__float2int_rn(lds_kp[4 * j + 1] * inv));
      if (!(cx >= margin && cy >= margin && cx < ww - margin && cy < s_lh[l] - margin)) continue;  // uniform: the border loop's
      const float ca = lds_cs[2 * j], sa = lds_cs[2 * j + 1];
      const uint32_t org = (uint32_t)(cy - ry0) * pitch_b + (uint32_t)(cx - rx0);
      unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((size_t)p * cap + j) * 32);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int val[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float px = ppx[r][e], py = ppy[r][e];
          const float xr = (px * ca) - (py * sa), yr = (px * sa) + (py * ca);
          const int dx = __float2int_rn(xr), dy = __float2int_rn(yr);
          val[e] = region8[org + (uint32_t)(dy * (int)pitch_b + dx)];
        }
        const unsigned long long bal = __ballot(val[0] < val[1]);
        if (lane == 0) d[r] = bal;
      }
    }
  }
  __syncthreads();  // (s_mode / s_rest complete; the area is free for the per-keypoint patches)
  if (!s_rest) return;  // uniform: every keypoint came from a region
  auto staged_loop = [&](auto nit_tag) __attribute__((always_inline)) {
    constexpr int NIT = decltype(nit_tag)::value;
    // (a kept keypoint lies >= 31 px inside level 0, so level 0 holds a (2 R + 1)-row patch with its centre at (R, R))
    const uint8_t* safe = blur + (size_t)img * P.total + (size_t)R * P.w[0] + R;
    auto fetch = [&](int j, uint32_t (&reg)[NIT]) __attribute__((always_inline)) {
      const uint8_t *im, *ctr;
      bool inside;
      const KpState k = locate(j, im, ctr, inside);
      const int pitch = k.staged ? k.ww : P.w[0];
      const uint8_t* c = (k.staged ? ctr : safe) - (size_t)(R * pitch + R);  // the patch's first byte (scalar)
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        reg[it] = *reinterpret_cast<const u32_unaligned*>(c + orb_mad24((uint32_t)prow[it], (uint32_t)pitch, (uint32_t)(4 * pk[it])));
      return k;
    };
    uint32_t reg_n[NIT];
    KpState nxt;
    if (wid_s < mkept) nxt = fetch(wid_s, reg_n);
    for (int j = wid_s; j < mkept; j += kThreads / 64) {
      const KpState k = nxt;
      uint32_t reg[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) reg[it] = reg_n[it];
      if (j + kThreads / 64 < mkept) nxt = fetch(j + kThreads / 64, reg_n);
      if (!k.staged) continue;  // (uniform; the second loop's)
      unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((size_t)p * cap + j) * 32);
      __builtin_amdgcn_wave_barrier();  // (the previous keypoint's reads of the patch are done: one wave, LDS in order)
#pragma unroll
      for (int it = 0; it < NIT; ++it) patch32[prow[it] * pstride + pk[it]] = reg[it];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int val[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float px = ppx[r][e], py = ppy[r][e];
          const float xr = (px * k.ca) - (py * k.sa), yr = (px * k.sa) + (py * k.ca);
          const int dx = __float2int_rn(xr), dy = __float2int_rn(yr);
          val[e] = patch8[orb_mad24((uint32_t)(dy + R), (uint32_t)(4 * pstride), (uint32_t)(dx + R))];
        }
        const unsigned long long bal = __ballot(val[0] < val[1]);
        if (lane == 0) d[r] = bal;
      }
    }
  };
  if (patch_ok) {  // uniform
    if (nit <= 7) staged_loop(std::integral_constant<int, 7>{}); else staged_loop(std::integral_constant<int, kDescPatchLoads>{});
  }
  // keypoints too close to their level's border for the patch, or a pattern beyond the staged radius: straight from memory
  for (int j = wid_s; j < mkept; j += kThreads / 64) {
    const uint8_t *im, *ctr;
    bool inside;
    const KpState k = locate(j, im, ctr, inside);
    if (k.staged || k.region) continue;  // uniform
    unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((size_t)p * cap + j) * 32);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
      int val[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = 2 * (64 * r + lane) + e;
        const float px = (float)spat[2 * i], py = (float)spat[2 * i + 1];
        const float xr = (px * k.ca) - (py * k.sa), yr = (px * k.sa) + (py * k.ca);
        const int dx = __float2int_rn(xr), dy = __float2int_rn(yr);
        if (inside) {
          val[e] = ctr[dy * k.ww + dx];
        } else {
          const int xx = refl101(k.cx + dx, k.ww), yy = refl101(k.cy + dy, k.hh);
          val[e] = im[(size_t)yy * k.ww + xx];
        }
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
  long long foff = 0;
  int toff = 0;
  for (int l = 0; l < P.nlev; ++l) {
    P.det[l] = P.quota[l] > 0 && P.h[l] > 2 * kEdge && P.w[l] > 2 * kEdge;
    if (P.det[l]) P.ndet = l + 1;
    P.fstrips[l] = cdiv(P.w[l], kFsStripW);
    P.fstrips2[l] = cdiv(P.w[l], kLp2StripW);
    P.foff[l] = foff;
    foff += (long long)P.h[l] * (P.fstrips[l] > 2 * P.fstrips2[l] ? P.fstrips[l] : 2 * P.fstrips2[l]);
    P.toff[l] = toff;
    if (l >= 1) toff += P.w[l] + P.h[l];
  }
  P.ftotal = foff;
  P.ttotal = toff;
  return P;
}

// levels 1 .. nlev_build - 1 of every image into `pyr` (level 0 is `gray` itself); `taps`: P.ttotal words of scratch
int32_t build_pyramid(sosvo_ctx* ctx, const uint8_t* gray, int nimg, int rows, int cols, const Pyr& P, int nlev_build,
                      uint8_t* pyr, uint32_t* taps, int32_t* lvl_rows = nullptr, const int8_t* pattern = nullptr) {
  if (nlev_build <= 1) return SOSVO_OK;
  int tmax = 0;
  for (int l = 1; l < nlev_build; ++l) tmax = tmax > P.w[l] + P.h[l] ? tmax : P.w[l] + P.h[l];
  SOSVO_LAUNCH(ctx, resize_taps_kernel, dim3(cdiv(tmax, kThreads), nlev_build - 1), dim3(kThreads), 0, ctx->stream, P, taps,
               pattern ? lvl_rows : nullptr, pattern);
  SOSVO_LAUNCH_CHECK(ctx);
  for (int l = 1; l < nlev_build; ++l) {
    const uint8_t* src = l == 1 ? gray : pyr + P.off[l - 1];
    const long long src_stride = l == 1 ? (long long)rows * cols : P.total;
    const int quads = cdiv(P.w[l], 4) + 2;  // + the leading pixels' thread, + one quad the alignment may add at the end
    const int nthr = quads < 1024 ? (quads + 63) & ~63 : 1024;
    SOSVO_LAUNCH(ctx, resize_level_kernel, dim3(cdiv(quads, nthr), P.h[l], nimg), dim3(nthr), 0, ctx->stream, src, src_stride,
                 P.h[l - 1], P.w[l - 1], pyr + P.off[l], P.total, P.h[l], P.w[l], taps + P.toff[l]);
    SOSVO_LAUNCH_CHECK(ctx);
  }
  return SOSVO_OK;
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// Scratch layout shared by the three ORB entry points: pyramid levels, FAST scores (same layout), blurred levels (same
// layout), local-maximum flags, mask bounding boxes, resize taps.
struct OrbScratch {
  uint8_t *pyr, *score, *blur;
  unsigned long long* flags;
  uint32_t *bbox, *taps;
  int32_t* lvl_rows;  // [kLevels][2]: blurred rows a detected keypoint's descriptor can read (written with the taps)
  uint2* rowtab;      // [kLevels][rowtab_stride]: orb_level_pass_kernel's row tables
  int rowtab_stride;
  size_t bbox_bytes, bytes;
};
OrbScratch orb_scratch(const Pyr& P, int nimg, int nsets, bool detect, bool describe, char* base) {
  OrbScratch o;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += al256(bytes);
    return p;
  };
  const size_t lv = (size_t)nimg * P.total;
  o.pyr = (uint8_t*)take(lv);
  o.score = (uint8_t*)take(detect ? lv : 0);
  o.blur = (uint8_t*)take(describe ? lv : 0);
  o.flags = (unsigned long long*)take(detect ? sizeof(unsigned long long) * (size_t)nimg * P.ftotal : 0);
  o.bbox_bytes = detect ? (size_t)nsets * kLevels * 32 * 4 * sizeof(uint32_t) : 0;
  o.bbox = (uint32_t*)take(o.bbox_bytes);
  o.taps = (uint32_t*)take(sizeof(uint32_t) * (size_t)(P.ttotal > 0 ? P.ttotal : 1));
  o.lvl_rows = (int32_t*)take(sizeof(int32_t) * 2 * kLevels);
  o.rowtab_stride = P.h[0] + 6;
  o.rowtab = (uint2*)take(detect ? sizeof(uint2) * (size_t)kLevels * o.rowtab_stride : 0);
  o.bytes = off;
  return o;
}

// detection on a pyramid whose levels < P.ndet exist: mask bounding boxes, FAST scores + flags of the detection levels
// (only the rows inside the 31-px border matter: candidates in [31, h - 31), their neighbours one row further), selection
int32_t run_orb_detect(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_pyr, int nimg, int images_per_maskset, int rows,
                       int cols, int nmask, int cap, const Pyr& P, const OrbScratch& W, float* kp4, float* resp, int32_t* n,
                       bool have_flags = false) {
  const int nsets = cdiv(nimg, images_per_maskset);
  SOSVO_HIP(ctx, hipMemsetAsync(W.bbox, 0, W.bbox_bytes, ctx->stream));
  int total_rows = 0;
  for (int l = 0; l < P.nlev; ++l) total_rows += P.h[l];
  SOSVO_LAUNCH(ctx, orb_mask_bbox_kernel, dim3(total_rows, nsets), dim3(kThreads), 0, ctx->stream, mask_pyr, P, nmask, W.bbox);
  SOSVO_LAUNCH_CHECK(ctx);
  for (int l = 0; l < P.ndet && !have_flags; ++l) {  // (have_flags: run_orb_level_passes wrote flags and scores)
    if (!P.det[l]) continue;
    const uint8_t* in = l == 0 ? gray : W.pyr + P.off[l];
    const long long in_stride = l == 0 ? (long long)rows * cols : P.total;
    const int32_t rc = launch_fast_score(ctx, in, in_stride, nimg, P.h[l], P.w[l], kFastThr, kEdge - 1, P.h[l] - kEdge + 1,
                                         W.score + P.off[l], P.total, W.flags + P.foff[l], P.ftotal);
    if (rc != SOSVO_OK) return rc;
  }
  LevelSrc S{gray, W.pyr, (long long)rows * cols, P.total};
  SOSVO_LAUNCH(ctx, orb_select_kernel, dim3(orb_xcd_grid(nimg, nmask)), dim3(kThreads), 0, ctx->stream, S, W.score,
               W.flags, mask_pyr, W.bbox, P, images_per_maskset, nmask, cap, kp4, resp, n, nimg,
               have_flags ? 1 : 0);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

// The detection levels in ONE pass each (orb_level_pass_kernel): flags + flagged scores of level l, its blur (describe), and
// level l + 1 -- instead of build_pyramid + the FAST launches of run_orb_detect + the blur launches of run_orb_describe.
// Needs what every real image has (a detection level is more than 62 px high and wide, and strictly higher than the next one).
bool orb_level_passes_ok(const Pyr& P) {
  if (P.ndet < 1 || P.h[0] + 6 > kRowTabLds) return false;
  for (int l = 0; l < P.ndet; ++l) {
    if (P.h[l] < 8 || P.w[l] < 8) return false;
    if (l + 1 < P.ndet && !(P.h[l] > P.h[l + 1] && P.w[l] >= P.w[l + 1])) return false;
  }
  return true;
}
int32_t run_orb_level_passes(sosvo_ctx* ctx, const uint8_t* gray, int nimg, int rows, int cols, const Pyr& P, const OrbScratch& W,
                             bool describe, const int8_t* pattern) {
  const bool restrict_rows = describe && P.ndet > 1;  // (as run_orb_describe)
  if (P.ndet > 1) {
    int tmax = 0;
    for (int l = 1; l < P.ndet; ++l) tmax = tmax > P.w[l] + P.h[l] ? tmax : P.w[l] + P.h[l];
    SOSVO_LAUNCH(ctx, resize_taps_kernel, dim3(cdiv(tmax, kThreads), P.ndet - 1), dim3(kThreads), 0, ctx->stream, P, W.taps,
                 (int32_t*)nullptr, (const int8_t*)nullptr);
    SOSVO_LAUNCH_CHECK(ctx);
  }
  SOSVO_LAUNCH(ctx, orb_rowtab_kernel, dim3(cdiv(P.h[0] + 6, kThreads), P.ndet), dim3(kThreads), 0, ctx->stream, P,
               describe ? pattern : (const int8_t*)nullptr, restrict_rows ? 1 : 0, W.rowtab_stride, W.rowtab, W.lvl_rows);
  SOSVO_LAUNCH_CHECK(ctx);
  for (int l = 0; l < P.ndet; ++l) {
    LevelPass A;
    A.in = l == 0 ? gray : W.pyr + P.off[l];
    A.in_stride = l == 0 ? (long long)rows * cols : P.total;
    A.nimg = nimg;
    A.rows = P.h[l];
    A.cols = P.w[l];
    A.strips = P.fstrips2[l];
    A.thr = kFastThr;
    A.rowtab = W.rowtab + (size_t)l * W.rowtab_stride;
    A.score = W.score + P.off[l];
    A.score_stride = P.total;
    A.flags = W.flags + P.foff[l];
    A.flags_stride = P.ftotal;
    A.blur = describe ? W.blur + P.off[l] : nullptr;
    A.blur_stride = P.total;
    const bool next = l + 1 < P.ndet;
    A.next = next ? W.pyr + P.off[l + 1] : nullptr;
    A.next_stride = P.total;
    A.w1 = next ? P.w[l + 1] : 0;
    A.xtaps = next ? W.taps + P.toff[l + 1] : nullptr;
    SOSVO_LAUNCH(ctx, orb_level_pass_kernel, dim3(cdiv(nimg * A.strips, kThreads / 64)), dim3(kThreads), 0, ctx->stream, A);
    SOSVO_LAUNCH_CHECK(ctx);
  }
  return SOSVO_OK;
}

// descriptors on a pyramid whose levels < nlev_have exist: blur those levels, border rule, rotated pattern
int32_t run_orb_describe(sosvo_ctx* ctx, const uint8_t* gray, int nimg, int rows, int cols, int nmask, int cap, const Pyr& P,
                         int nlev_have, const OrbScratch& W, float* kp4, int32_t* n, const int8_t* pattern, uint8_t* desc,
                         float* kp_xy, bool detected_keypoints, bool have_blur = false) {
  // detected_keypoints: every keypoint keeps the detector's 31-px border ON ITS LEVEL, so only the rows W.lvl_rows names are
  // ever read (keypoints handed in from outside keep that border in level-0 coordinates only: whole levels then)
  const bool restrict_rows = detected_keypoints && nlev_have > 1;  // (lvl_rows is written with the resize taps)
  for (int l = 0; l < nlev_have && !have_blur; ++l) {  // the rolling strip kernel of detect.hip, level by level (same integer arithmetic)
    const int32_t* rr = restrict_rows ? W.lvl_rows + 2 * l : nullptr;
    const int32_t rc = l == 0 ? sosvo_launch_gauss7_rows(ctx, gray, (long long)rows * cols, nimg, rows, cols, W.blur, P.total, rr, nimg)
                              : sosvo_launch_gauss7_rows(ctx, W.pyr + P.off[l], P.total, nimg, P.h[l], P.w[l], W.blur + P.off[l],
                                                         P.total, rr, nimg);
    if (rc != SOSVO_OK) return rc;
  }
  SOSVO_LAUNCH(ctx, orb_describe_levels_kernel, dim3(orb_xcd_grid(nimg, nmask)), dim3(kThreads),
               (size_t)cap * 6 * sizeof(float), ctx->stream, W.blur, P, nlev_have, rows, cols, nmask, cap, kp4, n, pattern, desc,
               kp_xy, nimg);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

}  // namespace

extern "C" {

#ifdef SOSVO_DEBUG_TIMING
int32_t sosvo_debug_orb_select_ticks(unsigned long long* out8, int32_t reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_sel_ticks), sizeof(unsigned long long) * 8) != hipSuccess) return SOSVO_ERR_HIP;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_sel_ticks), z, sizeof(z)) != hipSuccess) return SOSVO_ERR_HIP;
  }
  return SOSVO_OK;
}
#endif

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

#define SOSVO_ORB_DETECT_ARGS_OK(ctx)                                                                                       \
  SOSVO_REQUIRE(ctx, nimg >= 0 && nimg <= 65535 && images_per_maskset > 0, "nimg out of range");                            \
  SOSVO_REQUIRE(ctx, rows >= 1 && cols >= 1 && cols < 65536 && rows < 65536 && rows * (int64_t)cols < (1 << 28),            \
                "image sizes out of range");                                                                                \
  SOSVO_REQUIRE(ctx, nmask >= 1 && nmask <= 32 && nfeatures >= 1 && cap >= 1 && cap <= 4096, "bad detector parameters")

int32_t sosvo_detect_orb(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_pyr, int32_t nimg,
                         int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask, int32_t nfeatures,
                         int32_t cap, float* kp4, float* resp, int32_t* n) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, gray && mask_pyr && kp4 && resp && n, "null pointer");
  SOSVO_ORB_DETECT_ARGS_OK(ctx);
  if (nimg == 0) return SOSVO_OK;
  const Pyr P = make_pyr(rows, cols, nfeatures);
  const int nsets = cdiv(nimg, images_per_maskset);
  int32_t rc = sosvo_ws_reserve(ctx, orb_scratch(P, nimg, nsets, true, false, nullptr).bytes);
  if (rc != SOSVO_OK) return rc;
  const OrbScratch W = orb_scratch(P, nimg, nsets, true, false, (char*)ctx->ws);
  const bool fused = orb_level_passes_ok(P);
  rc = fused ? run_orb_level_passes(ctx, gray, nimg, rows, cols, P, W, false, nullptr)
             : build_pyramid(ctx, gray, nimg, rows, cols, P, P.ndet, W.pyr, W.taps);
  if (rc != SOSVO_OK) return rc;
  return run_orb_detect(ctx, gray, mask_pyr, nimg, images_per_maskset, rows, cols, nmask, cap, P, W, kp4, resp, n, fused);
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
  int32_t rc = sosvo_ws_reserve(ctx, orb_scratch(P, nimg, 1, false, true, nullptr).bytes);
  if (rc != SOSVO_OK) return rc;
  const OrbScratch W = orb_scratch(P, nimg, 1, false, true, (char*)ctx->ws);
  rc = build_pyramid(ctx, gray, nimg, rows, cols, P, P.nlev, W.pyr, W.taps);  // keypoints of any level: the whole pyramid
  if (rc != SOSVO_OK) return rc;
  return run_orb_describe(ctx, gray, nimg, rows, cols, nmask, cap, P, P.nlev, W, kp4, n, pattern, desc, kp_xy, false);
}

int32_t sosvo_detect_describe_orb(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_pyr, int32_t nimg,
                                  int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask, int32_t nfeatures,
                                  int32_t cap, float* kp4, float* resp, int32_t* n, const int8_t* pattern, uint8_t* desc,
                                  float* kp_xy) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, gray && mask_pyr && kp4 && resp && n && pattern && desc, "null pointer");
  SOSVO_ORB_DETECT_ARGS_OK(ctx);
  SOSVO_REQUIRE(ctx, cap <= 2048, "cap out of range (descriptors: <= 2048)");
  SOSVO_REQUIRE(ctx, ((uintptr_t)desc & 7) == 0, "desc must be 8-byte aligned");
  if (nimg == 0) return SOSVO_OK;
  const Pyr P = make_pyr(rows, cols, nfeatures);
  const int nsets = cdiv(nimg, images_per_maskset);
  int32_t rc = sosvo_ws_reserve(ctx, orb_scratch(P, nimg, nsets, true, true, nullptr).bytes);
  if (rc != SOSVO_OK) return rc;
  const OrbScratch W = orb_scratch(P, nimg, nsets, true, true, (char*)ctx->ws);
  // ONE pyramid for both halves, built only as far as a keypoint can come from (the levels with a quota and more than the
  // 31-px border): the detector's keypoints all lie on those levels
  const bool fused = orb_level_passes_ok(P);
  rc = fused ? run_orb_level_passes(ctx, gray, nimg, rows, cols, P, W, true, pattern)
             : build_pyramid(ctx, gray, nimg, rows, cols, P, P.ndet, W.pyr, W.taps, W.lvl_rows, pattern);
  if (rc != SOSVO_OK) return rc;
  rc = run_orb_detect(ctx, gray, mask_pyr, nimg, images_per_maskset, rows, cols, nmask, cap, P, W, kp4, resp, n, fused);
  if (rc != SOSVO_OK) return rc;
  return run_orb_describe(ctx, gray, nimg, rows, cols, nmask, cap, P, P.ndet, W, kp4, n, pattern, desc, kp_xy, true, fused);
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
    const int32_t rc2 = launch_fast_score(ctx, gray, (long long)rows * cols, nimg, rows, cols, threshold, 0, rows, score,
                                          (long long)rows * cols, nullptr, 0);
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

int32_t sosvo_detect_agast(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_bits, int32_t nimg,
                           int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask, int32_t threshold,
                           int32_t cap, float* kp, int32_t* n, int32_t* status) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, gray && mask_bits && kp && n, "null pointer");
  SOSVO_REQUIRE(ctx, nimg >= 0 && nimg <= 65535 && images_per_maskset > 0, "nimg out of range");
  SOSVO_REQUIRE(ctx, rows >= 7 && rows <= kAgastMaxRows && cols >= 7 && rows * (int64_t)cols < (1 << 24),
                "image sizes out of range (rows <= 1024, rows * cols < 2^24)");
  SOSVO_REQUIRE(ctx, nmask >= 1 && nmask <= 32, "nmask out of range (1..32)");
  SOSVO_REQUIRE(ctx, cap > 0 && cap <= 16384 && threshold >= 0 && threshold <= 254, "bad detector parameters");
  if (nimg == 0) return SOSVO_OK;
  const size_t score_bytes = ((size_t)nimg * rows * cols + 255) & ~(size_t)255;
  int32_t rc = sosvo_ws_reserve(ctx, score_bytes);
  if (rc != SOSVO_OK) return rc;
  uint8_t* score = (uint8_t*)ctx->ws;
  rc = launch_fast_score(ctx, gray, (long long)rows * cols, nimg, rows, cols, threshold, 0, rows, score, (long long)rows * cols,
                         nullptr, 0);
  if (rc != SOSVO_OK) return rc;
  SOSVO_LAUNCH(ctx, agast_nms_collect_kernel, dim3((unsigned)nimg), dim3(64), 0, ctx->stream, score, mask_bits, images_per_maskset,
               nmask, rows, cols, cap, kp, n, status);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

}  // extern "C"
