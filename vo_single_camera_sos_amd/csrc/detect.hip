// K4 -- goodFeaturesToTrack (the reference's default detector) and K6 -- ORB descriptors on given keypoints.
//
//   cv2.goodFeaturesToTrack(pano, maxCorners=N, qualityLevel=0.01, minDistance=5, mask=m, useHarris=False)
//        omnistereo/camera_models.py:1739 (per azimuthal mask), :1778 (no masks), pose_est_tools.py:544 (RGB-D)
//   ORB_create(nfeatures=N).compute(pano, keypoints)     camera_models.py:1765, :1785, pose_est_tools.py:553
//
// Batched over images (view-major) and azimuthal masks; problem p = image * nmask + mask.
//   eig      min-eigenvalue response in float32 with a fixed operation order (identical to the oracle's), the
//            per-mask maximum folded in through an order-preserving atomicMax; the response never goes to memory as a
//            map: the kernel keeps three rows in registers, tests the 3x3 local maximum there and emits the positive
//            local maxima as compact records (value, pixel) into a region of its own per wave;
//   select   one workgroup per problem: the records of the waves that cover the mask's bounding box, thresholded ->
//            sort keys (ordered(value) << 32 | pixel index) in LDS, rank sort (descending value,
//            higher address first), then the greedy minimum-distance pass on an LDS cell grid by the first wave,
//            64 candidates per step with an exact replay of the sequential acceptance rule;
//   blur     7x7 sigma=2 Gaussian in 8.8 fixed point (separable, LDS tile);
//   describe one workgroup per problem: border rule + stable compaction, then one wave per keypoint:
//            lane l evaluates tests l, 64+l, 128+l, 192+l; four 64-bit ballots ARE the 32 descriptor bytes.
// Everything is integer work or float32 with a pinned evaluation order -> bit-exact against the oracle.
#include "common.h"

#include <type_traits>

namespace {

constexpr int kThreads = 256;
constexpr int kCandCap = 4096;       // candidates kept per (image, mask): 32 KB of LDS sort keys
constexpr int kCandCapSmall = 2048;   // first pass of the two-pass selection: 16 KB of LDS keys, 1536 grid cells
constexpr int kCandCapLarge = 16384; // large-mask variant (whole-image detection, cap > 1024): 128 KB of the 160 KB LDS
constexpr int kMaxMasks = 32;

__device__ __forceinline__ int refl101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Value of the neighbouring lane through the DPP operand path (wave_shr:1 / wave_shl:1): a VALU move the compiler
// folds into the consuming add/max where it can, instead of a ds_bpermute through the LDS crossbar -- the rolling
// kernels issue 6-10 lane exchanges per row and the LDS pipe, shared by the four SIMDs of a CU, was their limit.
// Lane 0 / 63 receive 0 (they are halo lanes).  Only valid where "neighbouring lane" = "neighbouring column",
// i.e. not in the strips that touch the image border (mirrored columns): those keep the bpermute form.
__device__ __forceinline__ int lane_from_left(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ int lane_from_right(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ float lane_from_left(float v) { return __int_as_float(lane_from_left(__float_as_int(v))); }
__device__ __forceinline__ float lane_from_right(float v) { return __int_as_float(lane_from_right(__float_as_int(v))); }

// Workgroup -> problem (image, mask) for the kernels that run one workgroup per problem and read that image's data (the
// response map around the candidates, the blurred image under the descriptor patches): workgroups are dealt round-robin
// over the 8 XCDs by linear id, so a plain p = blockIdx.x spreads the 12 masks of one image over all eight L2s and
// every L2 fetches the image (measured: 4.1 MB of HBM fetches per pair for 0.84 MB of blurred images).  Here XCD x takes
// the images x, x + 8, x + 16, ... with all their masks: grid = 8 * ceil(nimg / 8) * nmask, -1 for the padding ids.
__device__ __forceinline__ int xcd_problem(int nimg, int nmask) {
  const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
  const int img = (slot / nmask) * 8 + xcd, m = slot - (slot / nmask) * nmask;
  return img < nimg ? img * nmask + m : -1;
}
static inline unsigned xcd_grid(int nimg, int nmask) { return (unsigned)(8 * cdiv(nimg, 8) * nmask); }

// ---- K4a: min-eigenvalue map, rolling over rows ---------------------------------------------------------
// One wave owns a 64-column strip (3 halo columns each side, 58 output columns) of one image and walks down a
// chunk of rows; everything a 3x3 Sobel + 3x3 box sum + 3x3 local-maximum test needs from neighbouring columns
// comes from neighbouring LANES (ds_bpermute), everything from neighbouring rows from three-deep register rings.
// Per row and lane: one byte load, ten lane exchanges, ~70 VALU ops; no LDS, no barriers.  Output: per-mask maxima and
// the candidate records -- a candidate of goodFeaturesToTrack is a pixel whose thresholded response is non-zero and
// equals its 3x3 dilation; with 0 < quality < 1 the threshold quality * max is positive wherever a candidate exists (a
// mask whose maximum is <= 0 has none: c > max * q >= max is impossible for c <= max < 0), so exactly the POSITIVE 3x3
// local maxima of the raw response qualify, whatever the threshold turns out to be.  Each wave appends those of its
// (strip, row chunk) to its own region (running position in an SGPR, no atomics, deterministic order); the selection
// kernel applies the threshold once the per-mask maxima are complete.
//   gray row t  -> hd = g[x+1] - g[x-1], hs = g[x-1] + 2 g[x] + g[x+1]                  (integers, exact)
//   P row v=t-1 -> dx = hd(v-1) + 2 hd(v) + hd(v+1), dy = hs(v+1) - hs(v-1), products dx*dx, dx*dy, dy*dy (f32)
//                  and their horizontal 3-sums (a + b) + c in the oracle's order
//   e row y=v-1 -> vertical sums (ha + hc) + hb, min eigenvalue, per-mask maximum, hm = max(e[x-1], e[x], e[x+1])
//   flag row y-1-> e == max(hm(y-2), hm(y-1), hm(y)): 3x3 local maximum; e > 0 and inside some mask -> one record
// Border rules (identical to the tile version it replaces): reflect-101 on the gray image, and a covariance
// product outside the image is the product AT the reflected position.  In x both are "take the mirrored
// lane".  In y a rolling reflect-101 of the gray rows evaluates the Sobel pair of the mirrored row with the
// row order reversed, i.e. dx unchanged and dy negated -- the dx*dy product is negated back (exact).
// Correctly rounded square root of a float that is ZERO OR NORMAL (the sum of two squares of image gradients products: zero
// or above 1e-30, far from the denormal range and from infinity): v_sqrt_f32 is good to one ulp; the result is moved down /
// up by one ulp where the residual says so -- exactly the correction the compiler emits for sqrtf, without its scaling of
// denormal inputs and its class test (7 of its 16 instructions, a seventh of this kernel).  Bit-identical to the host's
// sqrtf on these inputs; the parity tests compare every keypoint.
__device__ __forceinline__ float sqrt_rn_normal(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
  const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
  float r = (r_dn <= 0.0f) ? s_dn : s;
  r = (r_up > 0.0f) ? s_up : r;
  return r;
}

constexpr int kEigHalo = 3;
constexpr int kEigStripW = 64 - 2 * kEigHalo;  // 58

__device__ __forceinline__ void eig_flush(uint32_t* __restrict__ mstat_img, uint32_t bits, uint32_t omax) {
  while (bits) {
    const int m = __ffs(bits) - 1;
    bits &= bits - 1;
    atomicMax(&mstat_img[m * 5], omax);
  }
}

__global__ __launch_bounds__(kThreads) void min_eigen_kernel(const uint8_t* __restrict__ gray,
                                                             const uint32_t* __restrict__ mask_bits, int nimg,
                                                             int images_per_maskset, int rows, int cols, int nmask,
                                                             int strips, int nchunks, int chunk_rows, int wcap,
                                                             uint2* __restrict__ cand, uint32_t* __restrict__ wcnt,
                                                             uint32_t* __restrict__ mstat) {
  SOSVO_STREAMING_PRIO();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * kThreads + threadIdx.x) >> 6));  // in an SGPR: per-image bases become scalar
  if (wave >= nimg * strips * nchunks) return;  // wave-uniform
  const int img = wave / (strips * nchunks);
  const int rem = wave - img * strips * nchunks;
  const int chunk = rem / strips, strip = rem - chunk * strips;
  const int xb = strip * kEigStripW - kEigHalo;  // column of lane 0
  const int xc = xb + lane;
  const int xs = clampi(xc, 0, cols - 1);
  const int ll = clampi(refl101(xs - 1, cols) - xb, 0, 63);  // lane holding column x-1 (mirrored at the border)
  const int lr = clampi(refl101(xs + 1, cols) - xb, 0, 63);
  const bool out_lane = lane >= kEigHalo && lane < 64 - kEigHalo && xc < cols;  // xc >= 0 follows
  // Rows that hold a pixel of some mask of this image's mask set (mask_bbox_kernel, earlier on the stream): only they can
  // own a candidate or feed a per-mask maximum, so the chunk shrinks to them -- their 3x3 neighbourhoods still get the
  // response rows above and below (e_lo / e_hi), which need the gray rows two further out.  With the SOS panoramas the
  // azimuthal masks leave out the elevation padding: about a quarter of the rows.  (wave-uniform, scalar loads)
  uint32_t* mstat_img = mstat + (size_t)img * nmask * 5;
  int r_lo = rows, r_hi = 0;
  {
    const uint32_t* sbb = mstat + (size_t)(img / images_per_maskset) * images_per_maskset * nmask * 5;  // first image of the set
    for (int m = 0; m < nmask; ++m) {
      const uint32_t y1 = sbb[m * 5 + 4];
      if (y1) {
        r_lo = min(r_lo, (int)(0xFFFFFFFFu - sbb[m * 5 + 2]));
        r_hi = max(r_hi, (int)y1);
      }
    }
  }
  const int ys = max(chunk * chunk_rows, r_lo), ye = min(min(rows, chunk * chunk_rows + chunk_rows), r_hi);
  if (ys >= ye) {  // no mask pixel in this chunk: nothing to emit, no maximum to update
    if (lane == 0) wcnt[wave] = 0u;
    return;
  }
  const int e_lo = max(ys - 1, 0), e_hi = min(ye, rows - 1);  // e rows this chunk evaluates
  const int f_lo = max(ys, 1), f_hi = min(ye, rows - 1) - 1;   // flag rows this chunk owns
  const uint8_t* g = gray + (size_t)img * rows * cols;
  uint2* region = cand + (size_t)wave * wcap;  // this wave's candidate records (response, pixel)
  int wpos = 0;                                // records emitted so far (wave-uniform)
  uint32_t mb_prev = 0u;                       // mask word of the row the flag stage looks at (the previous e row)
  const uint32_t* mb = mask_bits + (size_t)(img / images_per_maskset) * rows * cols;
  const uint32_t mask_all = nmask < 32 ? (1u << nmask) - 1u : 0xFFFFFFFFu;
  const float scale = (float)(1.0 / 3060.0);

  // Three-deep register rings indexed by (row mod 3): the row loop is unrolled three times by a fold so that the
  // slot numbers are compile-time constants (no register shuffling); slot P holds the newest row of a phase-P step.
  int hd[3] = {0, 0, 0}, hs[3] = {0, 0, 0};                                          // gray rows t-2, t-1, t
  float pxx[3] = {0.f, 0.f, 0.f}, pxy[3] = {0.f, 0.f, 0.f}, pyy[3] = {0.f, 0.f, 0.f};  // h-sums of P rows v-2, v-1, v
  float hm[3] = {0.f, 0.f, 0.f}, ec[3] = {0.f, 0.f, 0.f};                            // e rows y-2, y-1, y
  uint32_t cur_bits = 0u, cur_max = 0u;
  const int t_first = e_lo - 2, t_last = e_hi + 2;

  // interior strips: the columns x-1 / x+1 of every lane that matters live in lanes -1 / +1
  const bool edge_strip = xb < 1 || xb + 64 > cols - 1;  // wave-uniform
  // The gray rows come three at a time, one ring revolution ahead (round 4, as orb_level_pass_kernel: a wave's rows are `cols`
  // bytes apart -- another cache line every step -- and a load used in the step that issues it leaves the step's whole
  // latency to the other waves of the SIMD).
  int c_cur[3] = {0, 0, 0}, c_pre[3] = {0, 0, 0};
  auto request = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 3; ++k) c_pre[k] = (int)g[(uint32_t)(refl101(min(t0 + k, t_last), rows) * cols) + (uint32_t)xs];
  };
  auto step = [&](auto phase_tag, auto edge_tag, const int t) __attribute__((always_inline)) {
    constexpr int P = decltype(phase_tag)::value;
    constexpr bool EDGE = decltype(edge_tag)::value;
    auto left = [&](auto v) { return EDGE ? __shfl(v, ll) : lane_from_left(v); };
    auto right = [&](auto v) { return EDGE ? __shfl(v, lr) : lane_from_right(v); };
    constexpr int i0 = (P + 1) % 3, i1 = (P + 2) % 3, i2 = P;  // slots of rows t-2, t-1, t
    if (t > t_last) return;  // uniform
    // the mask word of the e row this step finishes (y = t - 2) is requested first: its latency hides under the
    // Sobel / box-sum arithmetic below
    const int y_pre = t - 2;
    const bool y_out = y_pre >= ys && y_pre < ye && y_pre >= e_lo && out_lane;
    const uint32_t o_pre = (uint32_t)(max(y_pre, 0) * cols) + (uint32_t)xc;
    const uint32_t mb_pre = y_out ? mb[o_pre] : 0u;
    {
      const int c = c_cur[P];  // gray row t (requested a ring revolution ahead: rows * cols < 2^28)
      const int l = left(c), r = right(c);
      hd[i2] = r - l;
      hs[i2] = (l + 2 * c) + r;
    }
    const int v = t - 1;
    if (v < e_lo - 1) return;  // uniform
    {
      const int dxi = (hd[i0] + 2 * hd[i1]) + hd[i2], dyi = hs[i2] - hs[i0];
      const float dx = (float)dxi * scale, dy = (float)dyi * scale;
      const float xx = dx * dx, yy = dy * dy;
      float xy = dx * dy;
      if (v < 0 || v >= rows) xy = -xy;  // mirrored row: dy came out negated
      pxx[i2] = (left(xx) + xx) + right(xx);
      pxy[i2] = (left(xy) + xy) + right(xy);
      pyy[i2] = (left(yy) + yy) + right(yy);
    }
    const int y = v - 1;
    if (y < e_lo) return;  // uniform
    {
      const float a = ((pxx[i0] + pxx[i1]) + pxx[i2]) * 0.5f, b = (pxy[i0] + pxy[i1]) + pxy[i2];
      const float c = ((pyy[i0] + pyy[i1]) + pyy[i2]) * 0.5f;
      const float e = (a + c) - sqrt_rn_normal(((a - c) * (a - c)) + (b * b));
      if (y_out) {
        const uint32_t bits = mb_pre & mask_all;
        if (bits != cur_bits) {
          eig_flush(mstat_img, cur_bits, cur_max);
          cur_bits = bits;
          cur_max = 0u;
        }
        cur_max = max(cur_max, sosvo_float_ordered(e));
      }
      const float el = left(e), er = right(e);
      hm[i2] = fmaxf(fmaxf(el, e), er);
      ec[i2] = e;
    }
    const int f = y - 1;
    const uint32_t mb_f = mb_prev & mask_all;  // row f was the e row of the previous step
    mb_prev = mb_pre;
    if (f < f_lo || f > f_hi) return;  // uniform
    {
      const float ev = ec[i1];
      const bool is_cand = out_lane && xc >= 1 && xc <= cols - 2 && ev == fmaxf(fmaxf(hm[i0], hm[i1]), hm[i2]) && ev > 0.0f &&
                           mb_f != 0u;
      const unsigned long long bal = __ballot(is_cand);
      if (bal) {  // uniform
        const int pos = wpos + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (is_cand && pos < wcap) region[pos] = make_uint2(__float_as_uint(ev), (uint32_t)(f * cols + xc));
        wpos += __popcll(bal);
      }
    }
  };
  auto run = [&](auto edge_tag) __attribute__((always_inline)) {
    request(t_first);
    for (int t = t_first; t <= t_last; t += 3) {
#pragma unroll
      for (int k = 0; k < 3; ++k) c_cur[k] = c_pre[k];
      request(t + 3);
      step(std::integral_constant<int, 0>{}, edge_tag, t);
      step(std::integral_constant<int, 1>{}, edge_tag, t + 1);
      step(std::integral_constant<int, 2>{}, edge_tag, t + 2);
    }
  };
  if (edge_strip) run(std::true_type{}); else run(std::false_type{});
  if (lane == 0) wcnt[wave] = (uint32_t)wpos;  // may exceed wcap: the selection kernel reports the overflow
  // per-mask maxima: one set of atomics per wave when all its output lanes saw a single mask word
  const unsigned long long act = __ballot(out_lane && cur_bits != 0u);
  if (act) {
    const int first = __ffsll((long long)act) - 1;
    const uint32_t b0 = __shfl(cur_bits, first);
    if (__ballot(out_lane && cur_bits != 0u && cur_bits != b0) == 0ULL) {
      uint32_t om = (out_lane && cur_bits != 0u) ? cur_max : 0u;
      for (int s = 32; s > 0; s >>= 1) om = max(om, (uint32_t)__shfl_down((int)om, s));
      if (lane == 0) eig_flush(mstat_img, b0, om);
    } else {
      eig_flush(mstat_img, cur_bits, cur_max);
    }
  }
}

// Bounding box of every mask of every mask set (depends on the masks only): mstat fields 1..4 of the FIRST
// image of the set, each grown by atomicMax: 0xFFFFFFFF - xmin, 0xFFFFFFFF - ymin, xmax + 1, ymax + 1.
__global__ __launch_bounds__(kThreads) void mask_bbox_kernel(const uint32_t* __restrict__ mask_bits, int images_per_maskset,
                                                             int rows, int cols, int nmask, uint32_t* __restrict__ mstat) {
  __shared__ uint32_t sb[2][kMaxMasks];
  const int tid = threadIdx.x, y = blockIdx.x, set = blockIdx.y;
  if (tid < 2 * kMaxMasks) (&sb[0][0])[tid] = 0u;
  __syncthreads();
  const uint32_t* mb = mask_bits + ((size_t)set * rows + y) * cols;
  const uint32_t mask_all = nmask < 32 ? (1u << nmask) - 1u : 0xFFFFFFFFu;
  for (int x = tid; x < cols; x += kThreads) {
    uint32_t b = mb[x] & mask_all;
    while (b) {
      const int m = __ffs(b) - 1;
      b &= b - 1;
      atomicMax(&sb[0][m], 0xFFFFFFFFu - (uint32_t)x);
      atomicMax(&sb[1][m], (uint32_t)x + 1u);
    }
  }
  __syncthreads();
  if (tid < nmask && sb[1][tid]) {
    uint32_t* st = mstat + ((size_t)set * images_per_maskset * nmask + tid) * 5;
    atomicMax(&st[1], sb[0][tid]);
    atomicMax(&st[2], 0xFFFFFFFFu - (uint32_t)y);
    atomicMax(&st[3], sb[1][tid]);
    atomicMax(&st[4], (uint32_t)y + 1u);
  }
}

// LOGR consecutive steps of a bitonic merge (strides jtop, jtop / 2, ...) on 2^LOGR keys per thread: the keys
// i + m * jl (jl = the lowest stride) form a closed set under those steps, and they share the merge direction (bit k of i).
template <int LOGR, int NT>
__device__ __forceinline__ void bitonic_pass(unsigned long long* keys, int N, int k, int jtop, int tid) {
  constexpr int R = 1 << LOGR;
  const int jl = jtop >> (LOGR - 1);
  for (int t = tid; t < (N >> LOGR); t += NT) {
    const int i = ((t & ~(jl - 1)) << LOGR) | (t & (jl - 1));
    const bool desc = (i & k) == 0;
    unsigned long long v[R];
#pragma unroll
    for (int m = 0; m < R; ++m) v[m] = keys[i + m * jl];
#pragma unroll
    for (int sb = LOGR - 1; sb >= 0; --sb) {
#pragma unroll
      for (int m = 0; m < R; ++m) {
        if (m & (1 << sb)) continue;
        const unsigned long long a = v[m], b = v[m | (1 << sb)];
        const bool sw = (a < b) == desc;
        v[m] = sw ? b : a;
        v[m | (1 << sb)] = sw ? a : b;
      }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) keys[i + m * jl] = v[m];
  }
}

// One workgroup per problem (image, mask).  Phase 1 gathers the candidate records of the waves of min_eigen_kernel whose
// (strip, row chunk) meets the mask's bounding box (region counts fetched NT at a time, the waves share the regions, one
// LDS atomic per wave and 64 records), keeps those of this mask above the threshold and builds the sort keys in LDS; phase 2
// sorts them (bitonic, descending value, higher address first; three steps per pass over the keys); phase 3 -- the keys are
// dead, their LDS becomes the cell grid -- is the greedy minimum-distance pass, 64 candidates per round: the candidates are
// tested against the grid of already accepted points (9 cells x 2 slots) and against each other (a 64 x 64 conflict matrix),
// both spread over the workgroup's waves, and the first wave then replays the sequential acceptance rule exactly (accept iff
// no conflict with anything accepted before) -- in parallel for the candidates nothing earlier in the round can touch, on
// the scalar unit for the few others.  Round 3: 566 -> ~190 us per whole-image problem of 8 k candidates / 2000 corners
// (region walk 223 -> 25, sort 104 -> 84, greedy 234 -> 70).
// (NT threads per workgroup: 256, or 1024 for the whole-image variant whose 128 KB of LDS allow one workgroup per CU anyway)
// Squared distance of two points packed as y << 16 | x (coordinates below 32768): v_pk_sub_i16 + v_dot2_i32_i16
__device__ __forceinline__ uint32_t dist2_packed(uint32_t a, uint32_t b) {
  typedef short short2v __attribute__((ext_vector_type(2)));
  const short2v d = __builtin_bit_cast(short2v, a) - __builtin_bit_cast(short2v, b);
  return (uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false);
}

template <int CAND, int NT>
__global__ __launch_bounds__(NT) void gft_select_kernel(const uint2* __restrict__ cand,
                                                              const uint32_t* __restrict__ wcnt, int strips, int nchunks,
                                                              int chunk_rows, int wcap, const uint32_t* __restrict__ mask_bits,
                                                              const uint32_t* __restrict__ mstat, int images_per_maskset,
                                                              int nmask, int rows, int cols, double quality,
                                                              float min_distance, int cell, int max_corners, int cap,
                                                              uint32_t* __restrict__ sorted_g, int sorted_stride,
                                                              float* __restrict__ kp, int32_t* __restrict__ n_out,
                                                              int32_t* __restrict__ status, int32_t* __restrict__ redo,
                                                              int redo_pass, int nimg_total) {
  SOSVO_LATENCY_BOUND_PRIO();
  // Two-pass scheme (redo != nullptr): pass 0 runs the small variant (half the LDS: twice the workgroups per CU of
  // this latency-bound kernel) and hands the few problems that do not fit it -- more candidates than CAND, or a
  // mask bounding box larger than its cell grid -- to pass 1, which runs the full-size variant on those only.
  constexpr int kSelGridCells = (CAND * 8 - 1024 * 4) / 8;  // 2 slots (u32) per cell, 1024 u32 left for the list
  const int p = xcd_problem(nimg_total, nmask);
  if (p < 0) return;  // uniform (padding workgroup)
  if (redo && redo_pass == 1 && redo[p] == 0) return;  // uniform
  __shared__ unsigned long long lds_u64[CAND];
  __shared__ int s_count, s_accepted, s_overflow;
  __shared__ int s_region[NT];  // phase 1: record counts of the regions in flight
  __shared__ unsigned long long s_conf[64];  // phase 3: conflict rows of the round
  __shared__ unsigned long long s_prior;     // phase 3: candidates of the round that conflict with an earlier round's points
  __shared__ int s_ovf;                      // phase 3: accepted points that found both slots of their grid cell taken
  unsigned long long* keys = lds_u64;
  uint32_t* grid = reinterpret_cast<uint32_t*>(lds_u64);  // phase 3: 2 slots per cell, pixel index + 1 (0 = empty)
  uint32_t* acc_list = grid + 2 * kSelGridCells;          // phase 3 fallback: accepted pixel indices (<= 1024)
  const int tid = threadIdx.x;
  const int img = p / nmask, m = p - img * nmask;
  const uint32_t* st = mstat + (size_t)p * 5;                                                              // [0]: max
  const uint32_t* sb = mstat + ((size_t)(img / images_per_maskset) * images_per_maskset * nmask + m) * 5;  // bbox
  uint32_t* sorted = sorted_g + (size_t)p * sorted_stride;
  if (tid == 0) {
    s_count = 0;
    s_accepted = 0;
    s_overflow = 0;
    s_prior = 0ULL;
    s_ovf = 0;
  }
  __syncthreads();
  // ---- phase 1: candidates ------------------------------------------------------------------------------
  const bool any = st[0] != 0u;
  const int bx0 = any ? max(1, (int)(0xFFFFFFFFu - sb[1])) : 0, by0 = any ? max(1, (int)(0xFFFFFFFFu - sb[2])) : 0;
  const int bx1 = any ? min(cols - 2, (int)sb[3] - 1) : -1, by1 = any ? min(rows - 2, (int)sb[4] - 1) : -1;
  const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
  if (any && bw > 0 && bh > 0) {
    const float thr = (float)((double)sosvo_ordered_float(st[0]) * quality);
    const uint32_t* mb = mask_bits + (size_t)(img / images_per_maskset) * rows * cols;
    const int s0 = bx0 / kEigStripW, s1 = bx1 / kEigStripW;
    const int c0 = by0 / chunk_rows, c1 = by1 / chunk_rows;
    // The regions (strip, row chunk) that meet the bounding box, NT at a time: every thread fetches ONE region's count (the
    // dependent global loads of a region-by-region walk were most of this phase), then the waves take the regions round-robin,
    // 64 records per step, and claim their key slots with one LDS atomic per wave and step.
    const int ns = s1 - s0 + 1, R = (c1 - c0 + 1) * ns;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int rb = 0; rb < R; rb += NT) {
      if (rb + tid < R) {
        const int rg = rb + tid, c = c0 + rg / ns, sidx = s0 + rg % ns;
        const int have = (int)wcnt[(img * nchunks + c) * strips + sidx];  // the wave numbering of min_eigen_kernel
        if (have > wcap) s_overflow = 1;
        s_region[tid] = min(have, wcap);
      }
      __syncthreads();
      const int rend = min(NT, R - rb);
      for (int rr = wave; rr < rend; rr += NT / 64) {
        const int have = s_region[rr];
        const int rg = rb + rr, c = c0 + rg / ns, sidx = s0 + rg % ns;
        const uint2* region = cand + (size_t)((img * nchunks + c) * strips + sidx) * wcap;
        for (int i0 = 0; i0 < have; i0 += 64) {
          const int i = i0 + lane;
          uint2 r = make_uint2(0u, 0u);
          bool keep = false;
          if (i < have) {
            r = region[i];
            keep = __uint_as_float(r.x) > thr && ((mb[r.y] >> m) & 1u);
          }
          const unsigned long long km = __ballot(keep);
          if (km) {  // uniform
            const int leader = __ffsll((long long)km) - 1;
            int slot0 = 0;
            if (lane == leader) slot0 = atomicAdd(&s_count, __popcll(km));
            slot0 = __shfl(slot0, leader);
            const int slot = slot0 + __popcll(km & ((1ULL << lane) - 1ULL));
            if (keep && slot < CAND) keys[slot] = ((unsigned long long)sosvo_float_ordered(__uint_as_float(r.x)) << 32) | r.y;
          }
        }
      }
      __syncthreads();  // s_region is rewritten by the next tile
    }
  }
  __syncthreads();
  const int total = s_count;
  if (redo && redo_pass == 0) {  // uniform decision of the whole workgroup
    const int cx0_ = bx0 / cell, cy0_ = by0 / cell;
    const int gw_ = bw > 0 ? bx1 / cell - cx0_ + 1 : 0, gh_ = bh > 0 ? by1 / cell - cy0_ + 1 : 0;
    const bool fits = total <= CAND && (gw_ + 2) * (gh_ + 2) <= kSelGridCells;  // (the bordered grid of phase 3)
    if (tid == 0) redo[p] = fits ? 0 : 1;
    if (!fits) return;
  }
  const int n = min(total, CAND);
  // ---- phase 2: bitonic sort, descending (keys are unique: value, then higher address first) ---------------
  int N = 64;
  while (N < n) N <<= 1;
  for (int i = n + tid; i < N; i += NT) keys[i] = 0ULL;  // padding sorts last
  __syncthreads();
  for (int k = 2, lk = 1; k <= N; k <<= 1, ++lk) {
    // the lk steps of this merge (strides k / 2 ... 1), up to three per pass over the keys (8 keys per thread in registers:
    // a third of the LDS traffic and of the barriers of one step per pass -- the sort is LDS-bandwidth-bound)
    for (int left = lk, j = k >> 1; left > 0;) {
      const int take = left >= 3 ? 3 : left;
      if (take == 3) bitonic_pass<3, NT>(keys, N, k, j, tid);
      else if (take == 2) bitonic_pass<2, NT>(keys, N, k, j, tid);
      else bitonic_pass<1, NT>(keys, N, k, j, tid);
      j >>= take;
      left -= take;
      __syncthreads();
    }
  }
  // candidates leave as packed (y << 16 | x), in order
  for (int i = tid; i < n; i += NT) {
    const uint32_t pix = (uint32_t)keys[i];
    const uint32_t y = pix / (uint32_t)cols;
    sorted[i] = (y << 16) | (pix - y * (uint32_t)cols);
  }
  int stt = (total > CAND || s_overflow) ? 1 : 0;
  const int cx0 = bx0 / cell, cy0 = by0 / cell;
  const int gw = bw > 0 ? bx1 / cell - cx0 + 1 : 0, gh = bh > 0 ? by1 / cell - cy0 + 1 : 0;
  const bool use_grid = gw * gh <= kSelGridCells;
  const int limit_list = 1024;
  const float md2 = min_distance * min_distance;
  const bool spaced = min_distance >= 1.0f;
  int limit = cap;
  if (max_corners > 0 && max_corners < limit) limit = max_corners;
  // The usual case runs on integers (squared distances below 2^31, thresholds below 2^24: the comparisons are the float
  // ones exactly) with a one-cell border around the grid, and on ALL waves of the workgroup -- see below.
  constexpr uint32_t kEmpty = 0xFFFFFFFFu;
  const int gwp = gw + 2;
  // (workgroups of eight waves and more -- launches of few problems, where the pass is latency: the C2 batch of 4096 problems
  // keeps the one-wave form, which issues fewer instructions in total: + 0.35 % on the issue-bound step, measured)
  const bool fast = NT > 256 && spaced && n > 0 && cell <= 2048 && rows <= 32768 && cols <= 32768 && gwp * (gh + 2) <= kSelGridCells;  // uniform
  __syncthreads();  // keys are dead from here on; sorted[] is visible to the whole workgroup
  if (fast) {
    for (int i = tid; i < gwp * (gh + 2) * 2; i += NT) grid[i] = kEmpty;
  } else {
    for (int i = tid; i < (use_grid ? gw * gh * 2 : 0); i += NT) grid[i] = 0u;
  }
  __syncthreads();
  // ---- phase 3: greedy minimum-distance pass ----------------------------------------------------------------
  // 64 candidates per round, in order (every wave holds the round's candidates, one per lane).  (i) Conflicts with points
  // accepted in earlier rounds: cell grid with a border, 9 cells x 2 slots of packed coordinates, no bounds tests, empty
  // slots masked; one cell per work item, the hits OR-ed into s_prior.  (ii) Conflicts INSIDE the round: a 64 x 64 bit
  // matrix, row j = ballot(candidate closer than minDistance to candidate j), one row per work item, left in s_conf.
  // (iii) After the barrier the first wave replays the sequential rule: a survivor of (i) with no EARLIER survivor in its
  // row is accepted whatever happens to the others (decided by all lanes at once); the remaining ones are taken in order on
  // the scalar unit (find-first-bit, two v_readlane, mask tests).  The one-wave form further down recomputes a row per
  // accepted point: a dependent chain of ~10 instructions of a wave that runs alone, 90 ns per point.
  if (fast) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const uint32_t thr_i = (uint32_t)ceilf(md2);  // integer d2 < md2  <=>  d2 < ceil(md2)
    const uint32_t magic = cell > 1 ? 0xFFFFFFFFu / (uint32_t)cell + 1u : 0u;  // v / cell = mulhi(v, magic) for v < 2^16
    int accepted = 0;                                 // uniform over the workgroup
    uint32_t xy_next = lane < n ? sorted[lane] : 0u;  // one coalesced read per 64 candidates, requested a round ahead
    for (int k0 = 0; k0 < n && accepted < limit; k0 += 64) {
      const int kend = min(64, n - k0);
      const bool active = lane < kend;
      const uint32_t xy = xy_next;
      xy_next = k0 + 64 + lane < n ? sorted[k0 + 64 + lane] : 0u;
      const int y = (int)(xy >> 16), x = (int)(xy & 0xFFFFu);
      // cell in the bordered grid (every candidate of this mask lies inside the mask's bounding box)
      const int xc = active ? (int)(cell > 1 ? __umulhi((uint32_t)x, magic) : (uint32_t)x) - cx0 + 1 : 1;
      const int yc = active ? (int)(cell > 1 ? __umulhi((uint32_t)y, magic) : (uint32_t)y) - cy0 + 1 : 1;
      const int cidx = (yc * gwp + xc) * 2;
      // (i) + (ii): 9 cell tests and kend matrix rows are 9 + kend work items dealt round-robin over the waves (a lone wave
      // issues an instruction every 5+ cycles and waits for every dependent one: the round is latency, so it is spread)
      constexpr int NWV = NT / 64;
      for (int c9 = wave; c9 < 9; c9 += NWV) {  // uniform per wave
        const uint2 q2 = *reinterpret_cast<const uint2*>(&grid[cidx + (c9 / 3 - 1) * gwp * 2 + (c9 % 3 - 1) * 2]);
        const uint32_t hit = ((uint32_t)(q2.x != kEmpty) & (uint32_t)(dist2_packed(xy, q2.x) < thr_i)) |
                             ((uint32_t)(q2.y != kEmpty) & (uint32_t)(dist2_packed(xy, q2.y) < thr_i));
        const unsigned long long hb = __ballot(active && hit != 0u);  // conflict with a point accepted in an earlier round
        if (hb && lane == 0) atomicOr(&s_prior, hb);
      }
      for (int a = wave, novf = min(s_ovf, limit_list); a < novf; a += NWV) {  // (normally empty) uniform per wave
        const unsigned long long hb = __ballot(active && dist2_packed(xy, acc_list[a]) < thr_i);
        if (hb && lane == 0) atomicOr(&s_prior, hb);
      }
      {
        // rows first + t * NWV, four at a time (independent chains: the scheduler interleaves them); a wave's rows gather in
        // its lanes and leave with one LDS store
        const int first = (wave + NWV - 9 % NWV) % NWV;  // item 9 + j goes to wave (9 + j) % NWV
        uint32_t glo = 0u, ghi = 0u;
        int t = 0;
        for (int jb = first; jb < kend; jb += 4 * NWV, t += 4) {  // uniform per wave
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = min(jb + u * NWV, 63);  // (rows beyond kend: computed, never stored)
            const unsigned long long row = __ballot(active && dist2_packed(xy, (uint32_t)__builtin_amdgcn_readlane((int)xy, j)) < thr_i);
            const bool here = lane == t + u;
            glo = here ? (uint32_t)row : glo;
            ghi = here ? (uint32_t)(row >> 32) : ghi;
          }
        }
        const int jmine = first + lane * NWV;
        if (jmine < kend) s_conf[jmine] = ((unsigned long long)ghi << 32) | glo;
      }
      __syncthreads();
      if (wave == 0) {
        // (iii) a survivor without an EARLIER survivor in its row is accepted whatever happens to the others; the rest are
        // taken in order on the scalar unit: accepted iff no accepted earlier candidate is in their row
        const unsigned long long alive0 = __ballot(active) & ~s_prior;
        const unsigned long long mine = active ? s_conf[lane] : 0ULL;
        const unsigned long long lt = (1ULL << lane) - 1ULL;
        const bool surv = (alive0 >> lane) & 1ULL;
        const unsigned long long fr = __ballot(surv && (mine & alive0 & lt) == 0ULL);
        unsigned long long pending = alive0 & ~fr, acc = fr;
        const int rlo = (int)(uint32_t)mine, rhi = (int)(uint32_t)(mine >> 32);
        while (pending) {
          const int j = __ffsll((long long)pending) - 1;
          pending &= pending - 1ULL;
          const unsigned long long row = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(rhi, j) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane(rlo, j);
          if ((row & acc & ((1ULL << j) - 1ULL)) == 0ULL) acc |= 1ULL << j;
        }
        // the sequential pass stops at `limit` points: the first `room` of the accepted ones, in order
        const int room = limit - accepted;
        acc = __ballot(((acc >> lane) & 1ULL) && __popcll(acc & lt) < room);
        if ((acc >> lane) & 1ULL) {
          const int pos = accepted + __popcll(acc & ((1ULL << lane) - 1ULL));
          kp[((size_t)p * cap + pos) * 2] = (float)x;
          kp[((size_t)p * cap + pos) * 2 + 1] = (float)y;
          // a cell of round(minDistance) pixels holds two points -- or, from a distance of ~30 pixels on, three in a
          // triangle: the third goes to the overflow list every candidate is tested against
          if (atomicCAS(&grid[cidx], kEmpty, xy) != kEmpty && atomicCAS(&grid[cidx + 1], kEmpty, xy) != kEmpty) {
            const int o = atomicAdd(&s_ovf, 1);
            if (o < limit_list) acc_list[o] = xy;
          }
        }
        if (lane == 0) {
          s_accepted = accepted + __popcll(acc);
          s_prior = 0ULL;  // (read above by this wave only; the others add to it after the barrier)
        }
      }
      __syncthreads();
      accepted = s_accepted;
    }
    if (s_ovf > limit_list) stt |= 2;
  } else
  if (tid < 64) {  // one-wave form (float distances; the accepted points in a list when the grid does not fit)
    const int lane = tid;
    if (!use_grid && limit_list < limit) {
      limit = limit_list;
      stt |= 2;
    }
    int accepted = 0;
    uint32_t xy_next = lane < n ? sorted[lane] : 0u;  // one coalesced read per 64 candidates, requested a round ahead
    for (int k0 = 0; k0 < n && accepted < limit; k0 += 64) {
      const int kend = min(64, n - k0);
      const bool active = lane < kend;
      const uint32_t xy = xy_next;
      xy_next = k0 + 64 + lane < n ? sorted[k0 + 64 + lane] : 0u;
      const int y = (int)(xy >> 16), x = (int)(xy & 0xFFFFu);
      bool prior = false;  // conflict with a point accepted in an earlier round
      if (active && spaced) {
        if (use_grid) {
          const int xc = x / cell - cx0, yc = y / cell - cy0;
#pragma unroll
          for (int c9 = 0; c9 < 9; ++c9) {
            const int xx = xc + (c9 % 3) - 1, yy = yc + (c9 / 3) - 1;
            if (xx < 0 || xx >= gw || yy < 0 || yy >= gh) continue;
            const uint2 q2 = *reinterpret_cast<const uint2*>(&grid[(yy * gw + xx) * 2]);
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
              const uint32_t q = slot ? q2.y : q2.x;
              if (q) {
                const float dx = (float)(x - (int)((q - 1u) & 0xFFFFu)), dy = (float)(y - (int)((q - 1u) >> 16));
                prior = prior || (dx * dx + dy * dy < md2);
              }
            }
          }
          for (int a = 0, novf = min(s_ovf, limit_list); a < novf; ++a) {  // third points of their cells (normally none)
            const uint32_t q = acc_list[a];
            const float dx = (float)(x - (int)(q & 0xFFFFu)), dy = (float)(y - (int)(q >> 16));
            prior = prior || (dx * dx + dy * dy < md2);
          }
        } else {
          for (int a = 0; a < accepted; ++a) {
            const uint32_t q = acc_list[a];
            const float dx = (float)(x - (int)(q & 0xFFFFu)), dy = (float)(y - (int)(q >> 16));
            prior = prior || (dx * dx + dy * dy < md2);
          }
        }
      }
      unsigned long long alive = __ballot(active && !prior);
      unsigned long long acc = 0ULL;
      int room = limit - accepted;
      while (alive && room > 0) {
        const int j = __ffsll((long long)alive) - 1;
        alive &= alive - 1ULL;
        acc |= 1ULL << j;
        room--;
        if (spaced) {
          const int xj = __builtin_amdgcn_readlane(x, j), yj = __builtin_amdgcn_readlane(y, j);
          const float dx = (float)(x - xj), dy = (float)(y - yj);
          alive &= ~__ballot(dx * dx + dy * dy < md2);
        }
      }
      if ((acc >> lane) & 1ULL) {
        const int pos = accepted + __popcll(acc & ((1ULL << lane) - 1ULL));
        kp[((size_t)p * cap + pos) * 2] = (float)x;
        kp[((size_t)p * cap + pos) * 2 + 1] = (float)y;
        if (spaced) {
          if (use_grid) {
            uint32_t* c = &grid[((y / cell - cy0) * gw + (x / cell - cx0)) * 2];
            if (atomicCAS(&c[0], 0u, xy + 1u) != 0u && atomicCAS(&c[1], 0u, xy + 1u) != 0u) {  // third point of a cell
              const int o = atomicAdd(&s_ovf, 1);
              if (o < limit_list) acc_list[o] = xy;  // (an overflow is reported once, after the loop, from s_ovf)
            }
          } else {
            acc_list[pos] = xy;
          }
        }
      }
      accepted += __popcll(acc);
    }
    if (lane == 0) s_accepted = accepted;
  }
  __syncthreads();
  if (tid == 0) {
    if (s_ovf > limit_list) stt |= 2;  // the overflowing lane is not tid 0 in general: decide from the shared counter
    n_out[p] = min(s_accepted, cap);
    if (status) status[p] = stt;
  }
}

// a * b + c with 24-bit operands as ONE v_mad_u32_u24 (the compiler multiplies 32-bit values with v_mul_lo_u32 and adds
// separately; __umul24 does not survive its constant folding either).  Integer arithmetic: the order of the additions
// does not matter.
__device__ __forceinline__ uint32_t mad24(uint32_t k, uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(k), "v"(acc));
  return r;
}

// ---- K6a: 7x7 Gaussian, 8.8 fixed point ------------------------------------------------------------------
// Same rolling layout as min_eigen_kernel: a wave owns a 64-column strip (58 output columns) and walks down a
// chunk of rows; the six horizontal neighbours come from neighbouring lanes (mirrored lanes at the image
// border = reflect-101), the seven rows of horizontal sums live in a register ring.
__global__ __launch_bounds__(kThreads) void gauss7_kernel(const uint8_t* __restrict__ gray, long long img_stride, int nimg,
                                                          int rows, int cols, int strips, int nchunks, int chunk_rows,
                                                          uint8_t* __restrict__ out, long long out_stride,
                                                          const int32_t* __restrict__ row_range, int imgs_per_range) {
  SOSVO_STREAMING_PRIO();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * kThreads + threadIdx.x) >> 6));  // in an SGPR: per-image bases become scalar
  if (wave >= nimg * strips * nchunks) return;  // wave-uniform
  const int img = wave / (strips * nchunks);
  const int rem = wave - img * strips * nchunks;
  const int chunk = rem / strips, strip = rem - chunk * strips;
  const int xb = strip * kEigStripW - kEigHalo;
  const int xc = xb + lane;
  const int xs = clampi(xc, 0, cols - 1);
  int src[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) src[k] = clampi(refl101(xs + k - 3, cols) - xb, 0, 63);
  const bool out_lane = lane >= kEigHalo && lane < 64 - kEigHalo && xc < cols;
  int ys = chunk * chunk_rows, ye = min(rows, ys + chunk_rows);
  if (row_range) {  // only the rows a reader of the blurred image can reach (see sosvo_describe_orb_rows); wave-uniform
    const int v = img / imgs_per_range;
    ys = max(ys, row_range[2 * v]);
    ye = min(ye, row_range[2 * v + 1]);
    if (ys >= ye) return;
  }
  const uint8_t* g = gray + (size_t)img * img_stride;  // (img_stride = rows * cols for a dense batch; a pyramid level of
  uint8_t* o = out + (size_t)img * out_stride;        //  the ORB detector sits at a fixed offset of a larger per-image block)
  // seven-deep ring of horizontal sums indexed by (row mod 7): the row loop is unrolled seven times by a fold
  uint32_t h[7] = {0, 0, 0, 0, 0, 0, 0};
  const bool edge_strip = xb < 3 || xb + 64 > cols - 3;  // wave-uniform: some lane's x-3 .. x+3 are mirrored columns
  const int t_first = ys - 3, t_last = ye + 2;
  // gray rows in two half-batches into the ring slots just emptied, three to six steps ahead of their use (round 4, as
  // orb_level_pass_kernel: one row ahead does not cover a load that misses the L2)
  int c_pre[7];
  auto request = [&](int t0, int k0, int k1) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (k >= k0 && k < k1) c_pre[k] = (int)g[(uint32_t)(refl101(min(t0 + k, t_last), rows) * cols) + (uint32_t)xs];
  };
  request(t_first, 0, 7);
  auto step = [&](auto phase_tag, auto edge_tag, const int t) __attribute__((always_inline)) {
    constexpr int P = decltype(phase_tag)::value;
    constexpr bool EDGE = decltype(edge_tag)::value;
    if (t > t_last) return;  // uniform
    const int c = c_pre[P];  // gray row t
    int l1, l2, l3, r1, r2, r3;
    if (EDGE) {
      l3 = __shfl(c, src[0]); l2 = __shfl(c, src[1]); l1 = __shfl(c, src[2]);
      r1 = __shfl(c, src[4]); r2 = __shfl(c, src[5]); r3 = __shfl(c, src[6]);
    } else {
      l1 = lane_from_left(c); l2 = lane_from_left(l1); l3 = lane_from_left(l2);
      r1 = lane_from_right(c); r2 = lane_from_right(r1); r3 = lane_from_right(r2);
    }
    // (24-bit multiplies -- every operand is below 2^18, every sum below 2^24 -- so that each tap is ONE v_mad_u32_u24
    // instead of a 32-bit multiply plus an add)
    h[P] = mad24(18u, (uint32_t)(l3 + r3), mad24(34u, (uint32_t)(l2 + r2), mad24(49u, (uint32_t)(l1 + r1), 54u * (uint32_t)c)));
    const int y = t - 3;
    if (y < ys) return;  // uniform
    // rows y-3 .. y+3 = t-6 .. t live in slots P+1 .. P+7 (mod 7)
    const uint32_t vsum = mad24(18u, h[(P + 1) % 7] + h[P], mad24(34u, h[(P + 2) % 7] + h[(P + 6) % 7],
                                mad24(49u, h[(P + 3) % 7] + h[(P + 5) % 7], 54u * h[(P + 4) % 7])));
    if (out_lane) o[(uint32_t)(y * cols) + (uint32_t)xc] = (uint8_t)((vsum + 32768u) >> 16);
  };
  auto run = [&](auto edge_tag) __attribute__((always_inline)) {
    for (int t = t_first; t <= t_last; t += 7) {
      step(std::integral_constant<int, 0>{}, edge_tag, t);
      step(std::integral_constant<int, 1>{}, edge_tag, t + 1);
      step(std::integral_constant<int, 2>{}, edge_tag, t + 2);
      step(std::integral_constant<int, 3>{}, edge_tag, t + 3);
      request(t + 7, 0, 4);
      step(std::integral_constant<int, 4>{}, edge_tag, t + 4);
      step(std::integral_constant<int, 5>{}, edge_tag, t + 5);
      step(std::integral_constant<int, 6>{}, edge_tag, t + 6);
      request(t + 7, 4, 7);
    }
  };
  if (edge_strip) run(std::true_type{}); else run(std::false_type{});
}

// ---- K6b: border rule + descriptors ---------------------------------------------------------------------
constexpr int kPatchMaxR = 23, kPatchRows = 48, kPatchStride = 64;
constexpr int kDescRegionBytesGft = 24 * 1024;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
// (NT threads: a problem's keypoints are shared by NT / 64 waves, one keypoint per wave and step -- 256 for the C2 batch of
// 4096 problems, 512 when a launch has few problems, 1024 for the whole-image problems of up to 4096 keypoints)
template <int NT>
__global__ __launch_bounds__(NT) void orb_describe_kernel(const uint8_t* __restrict__ blurred, int rows, int cols,
                                                                int nmask, int cap, float* __restrict__ kp,
                                                                int32_t* __restrict__ n_io, float cos_a, float sin_a,
                                                                const int8_t* __restrict__ pattern, int edge,
                                                                uint8_t* __restrict__ desc, int nimg_total,
                                                                const int32_t* __restrict__ row_range, int imgs_per_range) {
  SOSVO_LATENCY_BOUND_PRIO();
  extern __shared__ float lds_kp[];  // [cap][2] compacted keypoints
  __shared__ int off[512];
  __shared__ uint16_t poff[512];
  // one LDS area, two uses (NT = 256, the batch of (image, mask) problems): the REGION of the blurred image under all of the
  // problem's keypoints (below), or the waves' per-keypoint patches; the larger workgroups keep the patches only
  constexpr int kPatchWords = (NT / 64) * (kPatchRows * kPatchStride / 4);
  constexpr int kRegionWords = NT == 256 ? kDescRegionBytesGft / 4 : 0;
  __shared__ uint32_t patch_area[kPatchWords > kRegionWords ? kPatchWords : kRegionWords];
  __shared__ int wave_off[NT / 64 + 1];
  __shared__ int s_running, s_R;
  __shared__ int s_bb[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int p = xcd_problem(nimg_total, nmask);
  if (p < 0) return;  // uniform (padding workgroup)
  const int img = p / nmask;
  const int n = min(n_io[p], cap);
  if (tid == 0) {
    s_running = 0;
    s_R = 0;
  }
  __syncthreads();
  for (int i = tid; i < 512; i += NT) {
    const float px = (float)pattern[2 * i], py = (float)pattern[2 * i + 1];
    const float xr = (px * cos_a) - (py * sin_a), yr = (px * sin_a) + (py * cos_a);
    const int dx = __float2int_rn(xr), dy = __float2int_rn(yr);
    off[i] = dy * cols + dx;
    atomicMax(&s_R, max(abs(dx), abs(dy)));
  }
  __syncthreads();
  const int R = s_R;
  // Patch path: the (2R+1)^2 neighbourhood of a keypoint goes through LDS, fetched as unaligned dwords -- a few
  // row segments per lane instead of 512 scattered byte loads.  Patch rows are `pstride` dwords apart with
  // pstride ODD, so the rows spread over all 32 banks (a 64-byte row pitch would leave half of them unused and
  // turn the scattered byte reads of the tests into 4-6-way conflicts).  The dword overrun past a patch row
  // (<= 3 bytes) stays inside the image as long as the border rule keeps R + 1 px.
  const bool patch_ok = R <= kPatchMaxR && edge >= R + 1;
  const int PR = 2 * R + 1;
  // With a row range only the blurred rows [rr_lo, rr_hi) exist, and those within 3 rows of an inner end were blurred
  // from gray rows outside the range: a keypoint whose patch rows leave the trustworthy part is REMOVED, like one too
  // close to the image border (callers that derive the range from the keypoints' masks never lose one this way).
  int y_first = 0, y_last = rows - 1;
  if (row_range) {
    const int v = img / imgs_per_range;
    const int rr_lo = row_range[2 * v], rr_hi = row_range[2 * v + 1];
    y_first = (rr_lo <= 0 ? 0 : rr_lo + 3) + R;
    y_last = (rr_hi >= rows ? rows : rr_hi - 3) - 1 - R;
  }
  const int pdw = (PR + 3) >> 2;      // dwords per patch row
  const int pstride = pdw | 1;        // row pitch in dwords
  if (patch_ok)
    for (int i = tid; i < 512; i += NT) {
      const int dy = (off[i] + R * cols + R) / cols - R;  // recover (dx, dy): |dx| <= R < cols
      const int dx = off[i] - dy * cols;
      poff[i] = (uint16_t)((dy + R) * (4 * pstride) + (dx + R));
    }
  for (int i0 = 0; i0 < n; i0 += NT) {
    const int i = i0 + tid;
    float x = 0.f, y = 0.f;
    bool keep = false;
    if (i < n) {
      x = kp[((size_t)p * cap + i) * 2];
      y = kp[((size_t)p * cap + i) * 2 + 1];
      keep = x >= (float)edge && x < (float)(cols - edge) && y >= (float)edge && y < (float)(rows - edge);
      const int yi = __float2int_rn(y);
      keep = keep && yi >= y_first && yi <= y_last;
    }
    const int pos = sosvo_block_compact_pos<NT / 64>(keep, wave_off, &s_running, tid);
    if (keep) {
      lds_kp[2 * pos] = x;
      lds_kp[2 * pos + 1] = y;
    }
  }
  __syncthreads();
  const int m = s_running;
  for (int i = tid; i < 2 * m; i += NT) kp[(size_t)p * cap * 2 + i] = lds_kp[i];
  if (tid == 0) n_io[p] = m;
  const uint8_t* im = blurred + (size_t)img * rows * cols;
  // ---- the usual case of the (image, mask) batch: every keypoint from ONE staged region (round 4) -------------------------
  // The ~165 keypoints of an azimuthal mask lie so close together that their (2R + 1)^2 patches cover the same pixels about
  // nine times over, and fetched per keypoint they are 390 scattered dwords each through the texture addresser (0.8 G dword
  // accesses per 256 frame pairs: more than any other kernel of the step).  The bounding box of the problem's keypoints grown
  // by R goes to LDS once (coalesced dword loads, ~20 KB for a 120-column mask of a 146-row panorama); the tests read it.
  if constexpr (NT == 256) {
    if (patch_ok && m > 0) {  // uniform
      if (tid < 4) s_bb[tid] = tid < 2 ? 0x7FFFFFFF : -1;
      __syncthreads();
      for (int i = tid; i < m; i += NT) {
        const int cx = __float2int_rn(lds_kp[2 * i]), cy = __float2int_rn(lds_kp[2 * i + 1]);
        atomicMin(&s_bb[0], cx);
        atomicMin(&s_bb[1], cy);
        atomicMax(&s_bb[2], cx);
        atomicMax(&s_bb[3], cy);
      }
      __syncthreads();
      const int rx0 = s_bb[0] - R, ry0 = s_bb[1] - R, rw = s_bb[2] - s_bb[0] + PR, rh = s_bb[3] - s_bb[1] + PR;
      const int ldw = (rw + 3) >> 2, pitch_dw = ldw | 1, pitch_b = 4 * pitch_dw;
      if (pitch_b * rh <= kDescRegionBytesGft) {  // uniform
        {  // eight requests in flight per thread, the position advanced without a division
          const int step_r = NT / ldw, step_k = NT - step_r * ldw;
          int ry = tid / ldw, k = tid - ry * ldw;
          const uint8_t* org = im + (size_t)ry0 * cols + rx0;
          while (__any(ry < rh)) {
            uint32_t v[8];
            int dst[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              dst[u] = ry < rh ? ry * pitch_dw + k : -1;
              v[u] = *reinterpret_cast<const u32_unaligned*>(org + (uint32_t)(min(ry, rh - 1) * cols + 4 * k));
              k += step_k;
              ry += step_r;
              if (k >= ldw) {
                k -= ldw;
                ++ry;
              }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (dst[u] >= 0) patch_area[dst[u]] = v[u];
          }
        }
        // this lane's eight test points as offsets inside the region (tests lane, 64 + lane, 128 + lane, 192 + lane)
        int ra[4], rb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int o = off[2 * (64 * r + lane) + e];
            const int dy = (o + R * cols + R) / cols - R, dx = o - dy * cols;  // recover (dx, dy): |dx| <= R < cols
            (e ? rb[r] : ra[r]) = dy * pitch_b + dx;
          }
        }
        __syncthreads();
        const uint8_t* region8 = reinterpret_cast<const uint8_t*>(patch_area);
        const int wid_s = __builtin_amdgcn_readfirstlane(wid);
        for (int j = wid_s; j < m; j += NT / 64) {
          const int cx = __builtin_amdgcn_readfirstlane(__float2int_rn(lds_kp[2 * j]));
          const int cy = __builtin_amdgcn_readfirstlane(__float2int_rn(lds_kp[2 * j + 1]));
          const int org = (cy - ry0) * pitch_b + (cx - rx0);
          unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((size_t)p * cap + j) * 32);
          unsigned long long mine = 0ULL;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned long long bal = __ballot(region8[org + ra[r]] < region8[org + rb[r]]);
            if (lane == r) mine = bal;
          }
          if (lane < 4) d[lane] = mine;  // one 32-byte store
        }
        return;
      }
      __syncthreads();  // (the per-keypoint patches below reuse the area)
    }
  }
  if (patch_ok) {
    uint32_t* patch32 = patch_area + wid * (kPatchRows * kPatchStride / 4);
    const uint8_t* patch = reinterpret_cast<const uint8_t*>(patch32);
    // 8 lanes (dwords) per patch row and 8 rows per round while a row fits 32 bytes, else 16 lanes x 4 rows.  Lanes
    // past the row's last dword / rows past the patch repeat the last one (same value to the same LDS word), so the
    // loads and the LDS writes are unconditional: no exec-mask juggling around a dozen memory instructions.
    const int lsh = pdw <= 8 ? 3 : 4;
    const int prow = lane >> lsh, pk = min(lane & ((1 << lsh) - 1), pdw - 1);
    const int rpr = 64 >> lsh, nit = (PR + rpr - 1) / rpr;  // rows per round, rounds (<= kPatchRows / 4)
    uint32_t reg[kPatchRows / 4];
    int goff[kPatchRows / 4], lidx[kPatchRows / 4];  // per round: image byte offset / LDS word of this lane's dword
#pragma unroll
    for (int it = 0; it < kPatchRows / 4; ++it) {
      const int r = min(rpr * it + prow, PR - 1);
      goff[it] = r * cols;
      lidx[it] = r * pstride + pk;
    }
    auto issue = [&](int j) {
      const uint8_t* base =
          im + ((size_t)__float2int_rn(lds_kp[2 * j + 1]) - R) * cols + (__float2int_rn(lds_kp[2 * j]) - R) + 4 * pk;
#pragma unroll
      for (int it = 0; it < kPatchRows / 4; ++it)
        if (it < nit) reg[it] = *reinterpret_cast<const u32_unaligned*>(base + goff[it]);  // uniform test
    };
    // this lane's eight patch offsets (tests lane, 64 + lane, 128 + lane, 192 + lane) never change: registers
    int pa[4], pb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pa[r] = poff[2 * (64 * r + lane)];
      pb[r] = poff[2 * (64 * r + lane) + 1];
    }
    int j = wid;
    if (j < m) issue(j);
    for (; j < m; j += NT / 64) {
#pragma unroll
      for (int it = 0; it < kPatchRows / 4; ++it)
        if (it < nit) patch32[lidx[it]] = reg[it];
      if (j + NT / 64 < m) issue(j + NT / 64);  // next keypoint's rows fly while this one is tested
      unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((size_t)p * cap + j) * 32);
      unsigned long long mine = 0ULL;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned long long bal = __ballot(patch[pa[r]] < patch[pb[r]]);
        if (lane == r) mine = bal;
      }
      if (lane < 4) d[lane] = mine;  // one 32-byte store
    }
    return;
  }
  for (int j = wid; j < m; j += NT / 64) {
    const uint8_t* c = im + (size_t)__float2int_rn(lds_kp[2 * j + 1]) * cols + __float2int_rn(lds_kp[2 * j]);
    unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((size_t)p * cap + j) * 32);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = 64 * r + lane;
      const unsigned long long bal = __ballot(c[off[2 * t]] < c[off[2 * t + 1]]);
      if (lane == 0) d[r] = bal;
    }
  }
}

// Row chunks of the rolling kernels: enough waves to fill 1024 SIMDs several times over, chunks of >= 24 rows
// (6 halo rows each).
static void rolling_chunks(int nimg, int rows, int strips, int* nchunks, int* chunk_rows) {
  int nc = cdiv(16384, nimg * strips);
  if (nc > rows / 24) nc = rows / 24;
  if (nc < 1) nc = 1;
  *chunk_rows = cdiv(rows, nc);
  *nchunks = cdiv(rows, *chunk_rows);
}

}  // namespace

// 7x7 sigma-2 blur (8.8 fixed point) of nimg images that lie img_stride bytes apart (shared with the ORB pyramid levels)
static int32_t launch_gauss7_rows(sosvo_ctx* ctx, const uint8_t* in, long long in_stride, int nimg, int rows, int cols,
                                  uint8_t* out, long long out_stride, const int32_t* row_range, int imgs_per_range) {
  const int strips = cdiv(cols, kEigStripW);
  int nchunks, chunk_rows;
  rolling_chunks(nimg, rows, strips, &nchunks, &chunk_rows);
  SOSVO_LAUNCH(ctx, gauss7_kernel, dim3(cdiv(nimg * strips * nchunks, kThreads / 64)), dim3(kThreads), 0, ctx->stream, in,
               in_stride, nimg, rows, cols, strips, nchunks, chunk_rows, out, out_stride, row_range, imgs_per_range);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}
int32_t sosvo_launch_gauss7_rows(sosvo_ctx* ctx, const uint8_t* in, long long in_stride, int nimg, int rows, int cols,
                                 uint8_t* out, long long out_stride, const int32_t* row_range, int imgs_per_range) {
  return launch_gauss7_rows(ctx, in, in_stride, nimg, rows, cols, out, out_stride, row_range, imgs_per_range);
}
int32_t sosvo_launch_gauss7_to(sosvo_ctx* ctx, const uint8_t* in, long long in_stride, int nimg, int rows, int cols,
                               uint8_t* out, long long out_stride) {
  return launch_gauss7_rows(ctx, in, in_stride, nimg, rows, cols, out, out_stride, nullptr, 1);
}
int32_t sosvo_launch_gauss7(sosvo_ctx* ctx, const uint8_t* in, long long img_stride, int nimg, int rows, int cols,
                            uint8_t* out) {
  return sosvo_launch_gauss7_to(ctx, in, img_stride, nimg, rows, cols, out, img_stride);
}

extern "C" {

int32_t sosvo_detect_gft(sosvo_ctx* ctx, const uint8_t* gray, const uint32_t* mask_bits, int32_t nimg,
                         int32_t images_per_maskset, int32_t rows, int32_t cols, int32_t nmask, double quality,
                         double min_distance, int32_t max_corners, int32_t cap, float* kp, int32_t* n,
                         int32_t* status) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, gray && mask_bits && kp && n, "null pointer");
  SOSVO_REQUIRE(ctx, nimg >= 0 && nimg <= 65535 && images_per_maskset > 0, "nimg out of range");
  SOSVO_REQUIRE(ctx, rows >= 3 && cols >= 3 && rows * (int64_t)cols < (1 << 28), "image sizes out of range");
  SOSVO_REQUIRE(ctx, nmask >= 1 && nmask <= kMaxMasks, "nmask out of range (1..32)");
  // quality >= 1 is legal as in cv2.goodFeaturesToTrack: the threshold quality * max is then >= every response (the
  // candidate records hold the POSITIVE local maxima only), so no corner passes `v > thr` and every count comes back 0
  SOSVO_REQUIRE(ctx, cap > 0 && cap <= 4096 && quality > 0 && min_distance >= 0, "bad detector parameters (quality > 0)");
  if (nimg == 0) return SOSVO_OK;
  const size_t P = (size_t)nimg * nmask;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const int strips = cdiv(cols, kEigStripW);
  int nchunks, chunk_rows;
  rolling_chunks(nimg, rows, strips, &nchunks, &chunk_rows);
  // candidate records (8 bytes: response, pixel): a region per wave of min_eigen_kernel with room for EVERY pixel of its
  // (strip, row chunk) -- 3x3 maxima of distinct values fill at most a quarter of it, but an exactly periodic texture
  // gives a plateau of equal positive responses on which every pixel is a candidate, and goodFeaturesToTrack handles
  // that; only what is written costs traffic
  const int wcap = (chunk_rows * kEigStripW + 63) & ~63;
  const size_t n_waves = (size_t)nimg * strips * nchunks;
  const size_t o_cand = carve(sizeof(uint2) * n_waves * wcap);
  const size_t o_wcnt = carve(sizeof(uint32_t) * n_waves);
  const size_t o_stat = carve(sizeof(uint32_t) * P * 5);
  const bool large = cap > 1024;  // whole-image masks (RGB-D frames): 16384 candidates, 15872 grid cells
  const int sorted_stride = large ? kCandCapLarge : kCandCap;
  const size_t o_sorted = carve(sizeof(uint32_t) * P * sorted_stride);
  const size_t o_redo = carve(sizeof(int32_t) * P);
  int32_t rc = sosvo_ws_reserve(ctx, off);
  if (rc != SOSVO_OK) return rc;
  char* ws = (char*)ctx->ws;
  uint2* cand = (uint2*)(ws + o_cand);
  uint32_t* wcnt = (uint32_t*)(ws + o_wcnt);
  uint32_t* mstat = (uint32_t*)(ws + o_stat);
  uint32_t* sorted_g = (uint32_t*)(ws + o_sorted);
  SOSVO_HIP(ctx, hipMemsetAsync(mstat, 0, sizeof(uint32_t) * P * 5, ctx->stream));

  const int nsets = cdiv(nimg, images_per_maskset);
  SOSVO_LAUNCH(ctx, mask_bbox_kernel, dim3(rows, nsets), dim3(kThreads), 0, ctx->stream, mask_bits, images_per_maskset, rows,
               cols, nmask, mstat);
  SOSVO_LAUNCH_CHECK(ctx);
  const int waves = nimg * strips * nchunks;
  SOSVO_LAUNCH(ctx, min_eigen_kernel, dim3(cdiv(waves, kThreads / 64)), dim3(kThreads), 0, ctx->stream, gray, mask_bits, nimg,
               images_per_maskset, rows, cols, nmask, strips, nchunks, chunk_rows, wcap, cand, wcnt, mstat);
  SOSVO_LAUNCH_CHECK(ctx);
  const int cell = min_distance >= 1 ? (int)lrint(min_distance) : 1;
  int32_t* redo = (int32_t*)(ws + o_redo);
  if (large) {
    SOSVO_PROFILE(ctx, "gft_select_kernel");
    hipLaunchKernelGGL((gft_select_kernel<kCandCapLarge, 1024>), dim3(xcd_grid(nimg, nmask)), dim3(1024), 0, ctx->stream, cand, wcnt,
                       strips, nchunks, chunk_rows, wcap, mask_bits, mstat, images_per_maskset, nmask, rows, cols, quality, (float)min_distance, cell,
                       max_corners, cap, sorted_g, sorted_stride, kp, n, status, nullptr, 0, nimg);
  } else {
    // Few problems (a window of a sequence, a handful of large frames): eight waves per workgroup share the rounds of the
    // greedy pass and the sort; many problems (the C2 batch: 4096 per launch): four, so that twice as many are resident.
    SOSVO_PROFILE(ctx, "gft_select_kernel");
    auto launch = [&](auto small, auto full, int nt) {
      hipLaunchKernelGGL(small, dim3(xcd_grid(nimg, nmask)), dim3(nt), 0, ctx->stream, cand, wcnt, strips, nchunks, chunk_rows, wcap,
                         mask_bits, mstat, images_per_maskset, nmask, rows, cols, quality, (float)min_distance, cell, max_corners, cap,
                         sorted_g, sorted_stride, kp, n, status, redo, 0, nimg);
      hipLaunchKernelGGL(full, dim3(xcd_grid(nimg, nmask)), dim3(nt), 0, ctx->stream, cand, wcnt, strips, nchunks, chunk_rows, wcap,
                         mask_bits, mstat, images_per_maskset, nmask, rows, cols, quality, (float)min_distance, cell, max_corners, cap,
                         sorted_g, sorted_stride, kp, n, status, redo, 1, nimg);
    };
    if (P <= 1024)
      launch(gft_select_kernel<kCandCapSmall, 512>, gft_select_kernel<kCandCap, 512>, 512);
    else
      launch(gft_select_kernel<kCandCapSmall, kThreads>, gft_select_kernel<kCandCap, kThreads>, kThreads);
  }
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_describe_orb(sosvo_ctx* ctx, const uint8_t* gray, int32_t nimg, int32_t rows, int32_t cols,
                           int32_t nmask, int32_t cap, float* kp, int32_t* n, float cos_a, float sin_a,
                           const int8_t* pattern, int32_t edge, uint8_t* desc) {
  return sosvo_describe_orb_rows(ctx, gray, nimg, rows, cols, nmask, cap, kp, n, cos_a, sin_a, pattern, edge, nullptr, desc);
}

int32_t sosvo_describe_orb_rows(sosvo_ctx* ctx, const uint8_t* gray, int32_t nimg, int32_t rows, int32_t cols,
                                int32_t nmask, int32_t cap, float* kp, int32_t* n, float cos_a, float sin_a,
                                const int8_t* pattern, int32_t edge, const int32_t* row_range, uint8_t* desc) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, row_range == nullptr || nimg % 2 == 0, "row_range needs the two views' images (nimg even, view-major)");
  SOSVO_REQUIRE(ctx, gray && kp && n && pattern && desc, "null pointer");
  SOSVO_REQUIRE(ctx, nimg >= 0 && nimg <= 65535 && nmask >= 1, "nimg / nmask out of range");
  SOSVO_REQUIRE(ctx, rows >= 1 && cols >= 1 && rows * (int64_t)cols < (1 << 28), "image sizes out of range");
  SOSVO_REQUIRE(ctx, cap > 0 && cap <= 4096, "cap out of range");
  SOSVO_REQUIRE(ctx, edge >= 19, "edge must cover the rotated 31x31 patch (>= 19)");
  SOSVO_REQUIRE(ctx, ((uintptr_t)desc & 7) == 0, "desc must be 8-byte aligned");
  if (nimg == 0) return SOSVO_OK;
  const size_t bytes = (size_t)nimg * rows * cols;
  int32_t rc = sosvo_ws_reserve(ctx, bytes);
  if (rc != SOSVO_OK) return rc;
  uint8_t* blurred = (uint8_t*)ctx->ws;
  rc = launch_gauss7_rows(ctx, gray, (long long)rows * cols, nimg, rows, cols, blurred, (long long)rows * cols, row_range,
                          nimg > 1 ? nimg / 2 : 1);
  if (rc != SOSVO_OK) return rc;
  const size_t lds_kp_bytes = (size_t)cap * 2 * sizeof(float);
  const int irange = nimg > 1 ? nimg / 2 : 1;
  SOSVO_PROFILE(ctx, "orb_describe_kernel");
  auto launch = [&](auto kernel, int nt) {
    hipLaunchKernelGGL(kernel, dim3(xcd_grid(nimg, nmask)), dim3(nt), lds_kp_bytes, ctx->stream, blurred, rows, cols, nmask, cap, kp,
                       n, cos_a, sin_a, pattern, edge, desc, nimg, row_range, irange);
  };
  if (cap > 1024)
    launch(orb_describe_kernel<1024>, 1024);
  else if ((size_t)nimg * nmask <= 1024)
    launch(orb_describe_kernel<512>, 512);
  else
    launch(orb_describe_kernel<kThreads>, kThreads);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

}  // extern "C"
