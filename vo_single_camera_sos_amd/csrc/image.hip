// Image stages of the hot path, batched over NI = 2 * nframes images (view-major: image i is
// view i / nframes of frame i % nframes).
//
//   K1 unwrap        cv2.remap at omnistereo/panorama.py:293 (+ the annulus masks of
//                    camera_models.py:2991-2996 folded into the bilinear taps)
//   K2+K3 median     cv2.medianBlur(pano, 11) at camera_models.py:1711 and cv2.cvtColor(BGR2GRAY) at :1714
//
// Both are integer/byte work.  K1 is a gather (HBM/L2-bound, coalesced on the pano rows).  K2 is the
// heavy one: 121-element medians for every pixel and channel.  It is computed WITHOUT sorting and
// without LDS: a wave owns a 54-column strip and walks down the rows; each new source row is turned
// into 8 bit-planes with 64-bit wave ballots (one SGPR pair per plane), every lane cuts its 11-bit
// horizontal window out of the ballot, and the 11 x 11 window lives in a register shift-register of
// packed bit-plane words (121 bits back to back in 4 VGPRs per plane, shifted with v_alignbit).  The median
// is then a radix select from the MSB down: AND + popcount on 4 words per bit decide whether the 61st
// smallest value has that bit set.  ~700 VALU ops per output pixel (3 channels) instead of ~5800 for
// compare-and-count.
#include "common.h"

#include <type_traits>
#include <utility>

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- K1 ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void unwrap_kernel(const uint8_t* __restrict__ omni,
                                                          const uint8_t* __restrict__ masks,
                                                          const float* __restrict__ map_x,
                                                          const float* __restrict__ map_y, int nframes, int H, int W,
                                                          int rows, int cols, uint8_t* __restrict__ pano) {
  const int npix = rows * cols;
  const int pix = blockIdx.x * kThreads + threadIdx.x;
  const int img = blockIdx.y;  // view-major
  if (pix >= npix) return;
  const int view = img / nframes, frame = img - view * nframes;
  const float mx = map_x[(size_t)view * npix + pix], my = map_y[(size_t)view * npix + pix];
  int acc0 = 0, acc1 = 0, acc2 = 0;
  if (mx == mx && my == my && mx > -4.0f && mx < (float)W + 4.0f && my > -4.0f && my < (float)H + 4.0f) {
    const int sx = __float2int_rn(mx * 32.0f), sy = __float2int_rn(my * 32.0f);  // ties to even
    const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
    const uint8_t* src = omni + (size_t)frame * H * W * 3;
    const uint8_t* msk = masks ? masks + (size_t)view * H * W : nullptr;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int x = ix + (t & 1), y = iy + (t >> 1);
      const int w = ((t & 1) ? fx : 32 - fx) * ((t >> 1) ? fy : 32 - fy);
      if (x < 0 || x >= W || y < 0 || y >= H) continue;
      const size_t o = (size_t)y * W + x;
      if (msk && !msk[o]) continue;
      acc0 += w * src[3 * o + 0];
      acc1 += w * src[3 * o + 1];
      acc2 += w * src[3 * o + 2];
    }
  }
  uint8_t* out = pano + ((size_t)img * npix + pix) * 3;
  out[0] = (uint8_t)((acc0 + 512) >> 10);
  out[1] = (uint8_t)((acc1 + 512) >> 10);
  out[2] = (uint8_t)((acc2 + 512) >> 10);
}

// ---- K1, table-driven form -----------------------------------------------------------------------------
// The float maps, the 1/32-px rounding, the border test and the annulus mask depend only on the model, so they
// are folded once into a table of one 8-byte entry per panorama pixel.  Two entry formats (bit 31 of .y tells):
//   fast  (.y bit 31 = 0): everything the per-frame kernels need, ready to use --
//         .x = off0 | w11 << 22     off0 = byte offset of tap (ix, iy) in the frame (< 2^22), the taps of the
//         .y = w00 | w01 << 11 | w10 << 21    next row are at off0 + 3 W; w.. = the four bilinear weights in 1/1024
//                                    (0 for a tap outside the image or masked out: w00 <= 1024, the others <= 992)
//         Both 8-byte row loads at off0 and off0 + 3 W lie inside the frame.  A pixel without valid taps is the
//         all-zero entry.
//   edge  (.y bit 31 = 1): a tap row starts before the frame or ends within its last 8 bytes (or the frame is
//         larger than 4 MB): .x = (iy << 16) | (ix & 0xFFFF) (int16 each), .y = fx | fy << 5 | valid << 10 | 1 << 31
//         (valid: 4 bits, tap t in-bounds and unmasked); the kernels clamp the loads and shift the bytes back.
// The per-frame kernels then do no float work, no mask loads, no bounds tests: one 8-byte table load, two
// unaligned 8-byte tap loads (the two taps of a source row are adjacent pixels = 6 consecutive bytes) and 12
// multiply-adds per pixel.
constexpr uint32_t kTabEdge = 0x80000000u;
constexpr uint32_t kTabOffMask = 0x3FFFFFu;

__global__ __launch_bounds__(kThreads) void unwrap_table_kernel(const uint8_t* __restrict__ masks,
                                                                const float* __restrict__ map_x,
                                                                const float* __restrict__ map_y, int H, int W, int npix,
                                                                uint2* __restrict__ table) {
  const int pix = blockIdx.x * kThreads + threadIdx.x, view = blockIdx.y;
  if (pix >= npix) return;
  const float mx = map_x[(size_t)view * npix + pix], my = map_y[(size_t)view * npix + pix];
  uint2 e = make_uint2(0u, 0u);
  if (mx == mx && my == my && mx > -4.0f && mx < (float)W + 4.0f && my > -4.0f && my < (float)H + 4.0f) {
    const int sx = __float2int_rn(mx * 32.0f), sy = __float2int_rn(my * 32.0f);
    const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
    const uint8_t* msk = masks ? masks + (size_t)view * H * W : nullptr;
    uint32_t valid = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int x = ix + (t & 1), y = iy + (t >> 1);
      if (x >= 0 && x < W && y >= 0 && y < H && (!msk || msk[(size_t)y * W + x])) valid |= 1u << t;
    }
    if (valid) {
      const long long off0 = 3LL * ((long long)iy * W + ix), off1 = off0 + 3LL * W, fb = 3LL * H * W;
      if (off0 >= 0 && off1 + 8 <= fb && off0 <= (long long)kTabOffMask) {
        const uint32_t w00 = (valid & 1u) ? (uint32_t)((32 - fx) * (32 - fy)) : 0u, w01 = (valid & 2u) ? (uint32_t)(fx * (32 - fy)) : 0u;
        const uint32_t w10 = (valid & 4u) ? (uint32_t)((32 - fx) * fy) : 0u, w11 = (valid & 8u) ? (uint32_t)(fx * fy) : 0u;
        e.x = (uint32_t)off0 | (w11 << 22);
        e.y = w00 | (w01 << 11) | (w10 << 21);
      } else {
        e.x = ((uint32_t)(iy & 0xFFFF) << 16) | (uint32_t)(ix & 0xFFFF);
        e.y = (uint32_t)fx | ((uint32_t)fy << 5) | (valid << 10) | kTabEdge;
      }
    }
  }
  table[(size_t)view * npix + pix] = e;
}

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef unsigned long long __attribute__((aligned(1))) u64_unaligned;

// Tap loads of one table entry, unconditional and branch-free so that they can be issued a whole row ahead of
// their use: a fast entry loads at off0 and off0 + 3 W; an edge entry loads at the clamped offsets (the blend
// shifts the bytes back).  `row_bytes` = 3 W.
__device__ __forceinline__ int edge_tap_offset(uint2 e, int r, int W) {
  const int ix = (int)(int16_t)(e.x & 0xFFFFu), iy = (int)(int16_t)(e.x >> 16);
  return 3 * ((iy + r) * W + ix);  // byte offset of tap (ix, iy + r) inside the frame; H * W * 3 < 2^31
}
__device__ __forceinline__ void unwrap_gather(const uint8_t* __restrict__ src, int frame_bytes, int W, uint2 e,
                                              unsigned long long v[2]) {
  uint32_t o0 = e.x & kTabOffMask, o1 = o0 + 3u * (uint32_t)W;
  const bool edge = (e.y & kTabEdge) != 0u;
  if (__ballot(edge) != 0ULL) {  // wave-uniform branch, rarely taken: some lane has frame-edge taps
    if (edge) {
      o0 = (uint32_t)min(max(edge_tap_offset(e, 0, W), 0), frame_bytes - 8);
      o1 = (uint32_t)min(max(edge_tap_offset(e, 1, W), 0), frame_bytes - 8);
    }
  }
  v[0] = *reinterpret_cast<const u64_unaligned*>(src + o0);
  v[1] = *reinterpret_cast<const u64_unaligned*>(src + o1);
}

// 1/32-px fixed-point blend of the gathered taps -> B | G << 8 | R << 16
__device__ __forceinline__ uint32_t unwrap_blend(int frame_bytes, int W, uint2 e, const unsigned long long vin[2]) {
  unsigned long long v0 = vin[0], v1 = vin[1];
  uint32_t w00 = e.y & 0x7FFu, w01 = (e.y >> 11) & 0x3FFu, w10 = (e.y >> 21) & 0x3FFu, w11 = e.x >> 22;
  const bool edge = (e.y & kTabEdge) != 0u;
  if (__ballot(edge) != 0ULL && edge) {  // wave-uniform skip; rare: undo the clamp of the loads, weights from (fx, fy, valid)
    const uint32_t valid = (e.y >> 10) & 15u;
    const uint32_t fx = e.y & 31u, fy = (e.y >> 5) & 31u;
    w00 = (valid & 1u) ? (32u - fx) * (32u - fy) : 0u;
    w01 = (valid & 2u) ? fx * (32u - fy) : 0u;
    w10 = (valid & 4u) ? (32u - fx) * fy : 0u;
    w11 = (valid & 8u) ? fx * fy : 0u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      unsigned long long& v = r ? v1 : v0;
      const int off = edge_tap_offset(e, r, W);
      const int delta = off - min(max(off, 0), frame_bytes - 8);  // bytes by which the load was moved
      if (delta > 0) v >>= 8 * delta;
      if (delta < 0) v <<= 8 * -delta;
    }
  }
  const uint32_t a0 = (uint32_t)v0, a1 = (uint32_t)(v0 >> 32), b0 = (uint32_t)v1, b1 = (uint32_t)(v1 >> 32);
  // bytes: a0 = [B G R | B'] a1 = [G' R' . .] of row 0 (left tap, right tap), b0 / b1 the same of row 1
  const uint32_t acc0 = w00 * (a0 & 0xFFu) + w01 * (a0 >> 24) + w10 * (b0 & 0xFFu) + w11 * (b0 >> 24);
  const uint32_t acc1 = w00 * ((a0 >> 8) & 0xFFu) + w01 * (a1 & 0xFFu) + w10 * ((b0 >> 8) & 0xFFu) + w11 * (b1 & 0xFFu);
  const uint32_t acc2 = w00 * ((a0 >> 16) & 0xFFu) + w01 * ((a1 >> 8) & 0xFFu) + w10 * ((b0 >> 16) & 0xFFu) + w11 * ((b1 >> 8) & 0xFFu);
  return ((acc0 + 512u) >> 10) | (((acc1 + 512u) >> 10) << 8) | (((acc2 + 512u) >> 10) << 16);
}

// One panorama pixel per lane (neighbouring lanes = neighbouring pixels, so a wave's taps fall on a short arc
// of the omni image: few cache lines per load instruction).  The 768 output bytes of a workgroup are staged
// in LDS and leave as 192 aligned dwords.
__global__ __launch_bounds__(kThreads) void unwrap_lut_kernel(const uint8_t* __restrict__ omni,
                                                              const uint2* __restrict__ table, int nframes, int H, int W,
                                                              int npix, uint8_t* __restrict__ pano) {
  __shared__ uint32_t stage[kThreads * 3 / 4];
  const int tid = threadIdx.x;
  // XCD-aware 1-D grid: workgroups are dealt round-robin over the 8 XCDs by linear id, so image = 8 * group +
  // (id % 8) keeps all blocks of one panorama (its annulus of the omni frame, its half of the table) on ONE
  // XCD's L2 instead of pulling every frame through all eight.
  const int gx = (npix + kThreads - 1) / kThreads;
  const int q = blockIdx.x >> 3;
  const int grp = q / gx, blk = q - grp * gx;
  const int img = grp * 8 + (blockIdx.x & 7);
  if (img >= 2 * nframes) return;  // uniform
  const int pix = blk * kThreads + tid;
  const int view = img / nframes, frame = img - view * nframes;
  const uint8_t* src = omni + (size_t)frame * H * W * 3;
  uint32_t bgr = 0u;
  if (pix < npix) {
    const uint2 e = table[(size_t)view * npix + pix];
    unsigned long long v[2];
    unwrap_gather(src, H * W * 3, W, e, v);
    bgr = unwrap_blend(H * W * 3, W, e, v);
  }
  const size_t out0 = ((size_t)img * npix + (size_t)blk * kThreads) * 3;  // first output byte of this workgroup
  const int nvalid = min(kThreads, npix - blk * kThreads);
  if ((out0 & 3) == 0 && nvalid == kThreads) {
    uint8_t* sb = reinterpret_cast<uint8_t*>(stage);
    sb[3 * tid + 0] = (uint8_t)bgr;
    sb[3 * tid + 1] = (uint8_t)(bgr >> 8);
    sb[3 * tid + 2] = (uint8_t)(bgr >> 16);
    __syncthreads();
    if (tid < kThreads * 3 / 4) reinterpret_cast<uint32_t*>(pano + out0)[tid] = stage[tid];
  } else if (pix < npix) {
    uint8_t* out = pano + out0 + 3 * tid;
    out[0] = (uint8_t)bgr;
    out[1] = (uint8_t)(bgr >> 8);
    out[2] = (uint8_t)(bgr >> 16);
  }
}

__device__ __forceinline__ uint8_t bgr2gray(int b, int g, int r) {
  return (uint8_t)((1868 * b + 9617 * g + 4899 * r + 8192) >> 14);
}

__global__ __launch_bounds__(kThreads) void gray_kernel(const uint8_t* __restrict__ img, size_t npix_total,
                                                        uint8_t* __restrict__ gray) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= npix_total) return;
  gray[i] = bgr2gray(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
}

// K1 + K3 without a median (median_win_size 0 / 1: the RGB-D frames' setting, and the setting under which the ORB detector
// finds its quota on panoramas): every lane unwraps its pixel from the table and converts it to gray at once -- the colour
// panoramas (2.5 MB per frame pair) are neither written nor read.  Same arithmetic as unwrap_lut_kernel + gray_kernel.
// NK pixels per thread, their table entries and gathers all in flight before the first blend: with one pixel per thread the
// kernel was a stream of 840 000 one-shot workgroups per 256 frame pairs, bound by workgroup dispatch and by one exposed load
// latency each (round 4).  What bounds it now is the texture addresser (round 4's counter pass: busy 80 % of the kernel's
// cycles, ~1.1 cache accesses per lane and gather -- one per DISTINCT dword a wave asks for; neighbouring lanes' taps overlap
// and are merged): a bilinear tap pair of BGR bytes per output pixel is what it is.  A wave gathers a 16 x 4 TILE of the
// panorama per load rather than 64 pixels of one row (64 neighbouring columns lie on a ~40-px arc of the omni frame that
// crosses up to ~40 image rows; the tile lands in a ~10 x 3 px patch: a third of the cache lines), and the gray bytes of the
// workgroup's 64 x (4 NK) tile meet in LDS and leave as 64-byte row segments.  Measured on its own the three mappings are
// within 2 % of each other (row 0.83 ms, tile with row-piece stores 0.87, tile + LDS 0.85 per 256 pairs): the dword count
// does not depend on the mapping -- but the STEP does: with three streams sharing the chip the tile + LDS form gives 49.5 k
// pairs/s against 48.1 k for the row form (A/B, same build otherwise): a third of the lines through the L2 leaves more of it to
// the other streams' kernels.  NK = 4 or 2: the tile height that wastes fewer rows.
typedef uint32_t __attribute__((aligned(1))) u32_unaligned_img;
template <int NK>
__global__ __launch_bounds__(kThreads) void unwrap_gray_kernel(const uint8_t* __restrict__ omni, const uint2* __restrict__ table,
                                                               int nframes, int H, int W, int rows, int cols, uint8_t* __restrict__ gray) {
  constexpr int TW = 64, TH = 4 * NK;
  __shared__ uint32_t tile[TH][TW / 4];
  const int nbx = (cols + TW - 1) / TW, nby = (rows + TH - 1) / TH;
  const int gx = nbx * nby, npix = rows * cols;   // XCD-aware 1-D grid as in unwrap_lut_kernel: one image on one XCD
  const int q = blockIdx.x >> 3;
  const int grp = q / gx, blk = q - grp * gx;
  const int img = grp * 8 + (blockIdx.x & 7);
  if (img >= 2 * nframes) return;  // uniform
  const int view = img / nframes, frame = img - view * nframes;
  const uint8_t* src = omni + (size_t)frame * H * W * 3;
  const int by = blk / nbx, bx = blk - by * nbx;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lx = wid * 16 + (lane & 15), ly = lane >> 4;  // this thread's column and first row inside the tile
  const int col = min(bx * TW + lx, cols - 1), row0 = by * TH + ly;
  uint2 e[NK];
  unsigned long long v[NK][2];
#pragma unroll
  for (int k = 0; k < NK; ++k) e[k] = table[(size_t)view * npix + (size_t)min(row0 + 4 * k, rows - 1) * cols + col];
#pragma unroll
  for (int k = 0; k < NK; ++k) unwrap_gather(src, H * W * 3, W, e[k], v[k]);
  uint8_t* tile8 = reinterpret_cast<uint8_t*>(&tile[0][0]);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const uint32_t bgr = unwrap_blend(H * W * 3, W, e[k], v[k]);
    tile8[(ly + 4 * k) * TW + lx] = bgr2gray((int)(bgr & 255u), (int)((bgr >> 8) & 255u), (int)((bgr >> 16) & 255u));
  }
  __syncthreads();
  // TH rows x 16 dwords: a wave stores whole 64-byte row segments
  for (int i = threadIdx.x; i < TH * (TW / 4); i += kThreads) {
    const int r = i / (TW / 4), c4 = i - r * (TW / 4);
    const int row = by * TH + r, c = bx * TW + 4 * c4;
    if (row >= rows || c >= cols) continue;
    uint8_t* o = gray + (size_t)img * npix + (size_t)row * cols + c;
    const uint32_t w4 = tile[r][c4];
    if (c + 3 < cols) {
      *reinterpret_cast<u32_unaligned_img*>(o) = w4;
    } else {
      for (int b = 0; c + b < cols; ++b) o[b] = (uint8_t)(w4 >> (8 * b));
    }
  }
}

// 64-bit wave ballot of "byte `byte` of x has its top bit set", as ONE SDWA compare (all lanes must be active).
__device__ __forceinline__ unsigned long long ballot_byte_sign(uint32_t x, int byte, uint32_t vzero) {
  unsigned long long bal;
  if (byte == 0) asm("v_cmp_lt_i16_sdwa %0, sext(%1), %2 src0_sel:BYTE_0 src1_sel:DWORD" : "=s"(bal) : "v"(x), "v"(vzero));
  else if (byte == 1) asm("v_cmp_lt_i16_sdwa %0, sext(%1), %2 src0_sel:BYTE_1 src1_sel:DWORD" : "=s"(bal) : "v"(x), "v"(vzero));
  else asm("v_cmp_lt_i16_sdwa %0, sext(%1), %2 src0_sel:BYTE_2 src1_sel:DWORD" : "=s"(bal) : "v"(x), "v"(vzero));
  return bal;
}
// popcount(x) + acc as ONE v_bcnt (kept as a chain: the compiler otherwise splits it into four independent
// popcounts plus an add3 to shorten the dependency chain, one more issue slot per bit in an issue-bound kernel)
__device__ __forceinline__ uint32_t popc_add(uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
// c & ~(w ^ m): the candidates whose window bit equals the chosen bit (m = all-ones / zero), one v_bitop3
__device__ __forceinline__ uint32_t keep_equal(uint32_t c, uint32_t m, uint32_t w) {
  uint32_t r;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x90" : "=v"(r) : "v"(c), "v"(m), "v"(w));
  return r;
}

// (a & ~m) | (x & m) as ONE v_bitop3 (table 0xD8: m ? x : a).  v_bfi_b32 computes the same but issues at the slow
// VALU rate (4.3 SIMD-cycles against 2.7, scripts/valu_rate.hip); m is a wave-uniform literal (an SGPR).
__device__ __forceinline__ uint32_t merge_field(uint32_t a, uint32_t x, uint32_t m) {
  uint32_t r;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xd8" : "=v"(r) : "v"(a), "v"(x), "s"(m));
  return r;
}
// a | (b & k): one fast v_bitop3 (table 0xF8) instead of v_and_or_b32
__device__ __forceinline__ uint32_t or_and(uint32_t a, uint32_t b, uint32_t k) {
  uint32_t r;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xf8" : "=v"(r) : "v"(a), "v"(b), "s"(k));
  return r;
}
// ~(a ^ b): v_xnor_b32 issues at the slow rate, v_bitop3 (table 0xC3) at the fast one
__device__ __forceinline__ uint32_t xnor_fast(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_bitop3_b32 %0, %1, %2, %2 bitop3:0xc3" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// Writes lane l's K-bit slice of a row ballot -- bits [l, l + K) of `bal` -- into ring slot `slot` of one bit-plane's
// window: slot s occupies bits [K * s, K * s + K) of the NW-word register.  `slot` is a compile-time constant after
// unrolling: masks and shifts are literals.
//   slot inside one word: the ballot is shifted LEFT by sh_up = 64 - K - l, which puts the slice into the top K bits of
//     the high word, and a 32-bit RIGHT shift (fast VALU rate; v_lshlrev_b32 issues at the slow one) drops it onto the
//     slot; whatever lands below the slot is cut by the merge mask;
//   slot straddling two words: slice = bal >> l, then a shift + merge per word.
template <int K, int NW>
__device__ __forceinline__ void ring_insert(uint32_t (&Wp)[NW], unsigned long long bal, int lane, int sh_up, int slot) {
  const int bit = K * slot, w = bit >> 5, pos = bit & 31;
  const uint32_t field = (1u << K) - 1u;
  if (pos + K <= 32) {
    const uint32_t hi = (uint32_t)((bal << sh_up) >> 32);
    const uint32_t t = (32 - K - pos) ? hi >> (32 - K - pos) : hi;
    Wp[w] = merge_field(Wp[w], t, field << pos);
  } else {
    const uint32_t x = (uint32_t)(bal >> lane);
    Wp[w] = merge_field(Wp[w], x << pos, field << pos);
    Wp[w + 1] = merge_field(Wp[w + 1], x >> (32 - pos), field >> (32 - pos));
  }
}

template <typename F, int... I>
__device__ __forceinline__ void for_each_slot(F& f, int r_base, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}, r_base + I), ...);
}

// ---- K2 + K3 (optionally with K1 in front) --------------------------------------------------------------
// K x K median per channel (replicated border) followed by BGR->gray.  K odd, 3 <= K <= 11.
// FUSED = false: img is the BGR panorama batch [nimg, rows, cols, 3].
// FUSED = true : img is the omni batch [nframes, H, W, 3] and every lane unwraps its source pixel from the table
//                on the fly (same integer arithmetic as unwrap_lut_kernel), so the colour panoramas never touch
//                HBM: the per-row gather hides under the other waves' VALU work.
// Window registers: per (channel, bit-plane) the K * K window bits are a RING of K row slots of K bits in NW words
// (the median does not care about the order of the window's elements): a row step overwrites the oldest slot with
// the lane's slice of the ballot -- one shift + one v_bfi with literal mask -- instead of shifting the whole
// register; the row loop is unrolled K times so that the slot number is a compile-time constant.
template <int K, bool FUSED>
__global__ __launch_bounds__(kThreads) void median_gray_kernel(const uint8_t* __restrict__ img,
                                                               const uint2* __restrict__ table, int nframes, int H, int W,
                                                               int nimg, int rows, int cols, int strips,
                                                               const int32_t* __restrict__ row_range,
                                                               uint8_t* __restrict__ gray) {
  constexpr int R = K / 2;
  constexpr int NB = K * K;
  constexpr int NW = (NB + 31) / 32;      // words per bit-plane: 4 for 11 x 11
  constexpr int OUTW = 64 - 2 * R;        // output columns per wave
  constexpr uint32_t ABOVE = NB - (NB / 2 + 1);  // window elements ranked above the median
  const int lane = threadIdx.x & 63;
  // XCD-aware grid: workgroups are dealt round-robin over the 8 XCDs by linear id.  XCD x = id % 8 walks its own
  // list of (image, strip) pairs -- images x, x + 8, x + 16, ... strip by strip, 4 waves per workgroup -- so all
  // strips of one image (which share the cache lines along their common sector borders and the image's part of the
  // table) go through ONE XCD's L2, and no wave slot idles.  Everything here is scalar (SGPRs).
  const int L = (int)(blockIdx.x >> 3) * (kThreads / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int il = L / strips, strip = L - il * strips;
  const int im = il * 8 + (int)(blockIdx.x & 7);
  if (im >= nimg) return;  // wave-uniform
  const int x0 = strip * OUTW;
  const int x_src = clampi(x0 - R + lane, 0, cols - 1);
  const int x_out = x0 + lane;
  const bool out_ok = lane < OUTW && x_out < cols;
  uint8_t* dst = gray + (size_t)im * rows * cols;

  // ---- source pixel of (row, x_src) as B | G << 8 | R << 16, fetched one row ahead of its use
  const int view = FUSED ? im / nframes : 0, frame = FUSED ? im - view * nframes : 0;
  const uint8_t* src = FUSED ? img + (size_t)frame * H * W * 3 : img + (size_t)im * rows * cols * 3;
  // output rows [out_lo, out_hi) of this image's view (row_range: the rows a consumer can reach, see
  // sosvo_gray_rows_needed; nullptr = all): rows outside are neither computed nor written
  int out_lo = 0, out_hi = rows;
  if (FUSED && row_range) {
    out_lo = max(0, min(rows, row_range[2 * view]));
    out_hi = max(out_lo, min(rows, row_range[2 * view + 1]));
  }
  if (out_hi <= out_lo) return;  // wave-uniform
  const uint2* tab = FUSED ? table + (size_t)view * rows * cols + x_src : nullptr;
  const int frame_bytes = H * W * 3;
  const bool last_image = im == nimg - 1;
  uint2 e_cur = make_uint2(0u, 0u), e_nxt = make_uint2(0u, 0u);
  unsigned long long taps[2] = {0ULL, 0ULL};
  uint32_t raw = 0u;
  auto row_of = [&](int r) { return clampi(r, 0, rows - 1); };
  auto load_raw = [&](int r) -> uint32_t {
    const int rr = row_of(r);
    const uint8_t* px = src + ((size_t)rr * cols + x_src) * 3;
    if (!(last_image && rr == rows - 1 && x_src == cols - 1)) return *reinterpret_cast<const u32_unaligned*>(px);
    return (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);  // the buffer's last 3 bytes
  };
  if (FUSED) {
    e_cur = tab[(size_t)row_of(out_lo - R) * cols];
    e_nxt = tab[(size_t)row_of(out_lo - R + 1) * cols];
    unwrap_gather(src, frame_bytes, W, e_cur, taps);
  } else {
    raw = load_raw(-R);
  }

  uint32_t vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));  // SDWA compares take no constant operand
  const int sh_up = (64 - K - lane) & 63;         // see ring_insert (lanes beyond OUTW hold no output)
  uint32_t Wp[3][8][NW];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int j = 0; j < NW; ++j) Wp[c][b][j] = 0u;

  // one source row: ring slot = compile-time constant (the row loop below is unrolled K times by a fold)
  auto row_step = [&](auto slot_tag, const int r_src) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_tag)::value;
    if (r_src >= out_hi + R) return;  // uniform (the tail of the last group)
    uint32_t pix;
    if (FUSED) {
      pix = unwrap_blend(frame_bytes, W, e_cur, taps);
      e_cur = e_nxt;
      unwrap_gather(src, frame_bytes, W, e_cur, taps);   // row r_src + 1
      e_nxt = tab[(size_t)row_of(r_src + 2) * cols];  // row r_src + 2
    } else {
      pix = raw;
      raw = load_raw(r_src + 1);
    }
    // overwrite the oldest row slot with the new row's 24 bit-plane slices
    uint32_t sh = pix;  // bit b of every channel sits in the sign position of its byte
#pragma unroll
    for (int b = 7; b >= 0; --b) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        ring_insert<K, NW>(Wp[c][b], ballot_byte_sign(sh, c, vzero), lane, sh_up, slot);
      // next plane: sh <<= 1 as an add (fast VALU rate; the shift issues at the slow one)
      if (b > 0) asm("v_add_u32 %0, %1, %1" : "=v"(sh) : "v"(sh));
    }
    const int r_out = r_src - R;
    if (r_out < out_lo) return;  // window not complete yet (uniform)
    int med[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // radix select from the MSB down.  `above` = candidates ranked above the wanted element; the candidates
      // whose current bit is 1 rank above those whose bit is 0.
      uint32_t C[NW];
#pragma unroll
      for (int j = 0; j < NW; ++j) C[j] = 0xFFFFFFFFu;
      // (the unused top bits of the last word are never set in a window register, so no mask is needed)
      uint32_t above = ABOVE, res = 0u;
#pragma unroll
      for (int b = 7; b >= 0; --b) {
        uint32_t n1 = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) n1 = popc_add(C[j] & Wp[c][b][j], n1);
        const uint32_t d = above - n1;
        const uint32_t ones = (uint32_t)((int32_t)d >> 31);  // all-ones: the wanted element has bit b set
        // above = min(above, d) as a 16-bit minimum (v_min_u16 issues at the fast rate, v_min_u32 at the slow one):
        // above <= NB, and a negative d has its low half >= 0x10000 - NB > above
        above = (uint32_t)min((uint16_t)above, (uint16_t)d);
        res = or_and(res, ones, 1u << b);
#pragma unroll
        for (int j = 0; j < NW; ++j) C[j] = b == 7 ? xnor_fast(Wp[c][b][j], ones) : keep_equal(C[j], ones, Wp[c][b][j]);
      }
      med[c] = (int)res;
    }
    if (out_ok) dst[(size_t)r_out * cols + x_out] = bgr2gray(med[0], med[1], med[2]);
  };
  for (int r_base = out_lo - R; r_base < out_hi + R; r_base += K)
    for_each_slot(row_step, r_base, std::make_integer_sequence<int, K>{});
}

// launches the K-templated strip kernel; FUSED takes the omni batch + unwrap table instead of panoramas
template <bool FUSED>
int32_t launch_median(sosvo_ctx* ctx, const uint8_t* img, const uint2* table, int nframes, int H, int W, int nimg,
                             int rows, int cols, int ksize, const int32_t* row_range, uint8_t* gray) {
  const int outw = 64 - 2 * (ksize / 2);
  const int strips = cdiv(cols, outw);
  dim3 grid(8 * cdiv(cdiv(nimg, 8) * strips, kThreads / 64)), block(kThreads);  // see the kernel's XCD-aware mapping
  // (named aliases so that the profile labels tell the two forms apart)
  constexpr auto k11 = median_gray_kernel<11, FUSED>;
  constexpr auto k5 = median_gray_kernel<5, FUSED>;
  constexpr auto k3 = median_gray_kernel<3, FUSED>;
  SosvoProfScope prof(ctx, FUSED ? "unwrap_median_gray_kernel" : "median_gray_kernel");
  // SOSVO_HINT_SHARED_DEVICE: the median's 127 VGPRs x 4 waves fill a SIMD's register file, so while it runs the other
  // streams' kernels only get wave slots as its workgroups retire.  41 KB of (unused) dynamic LDS per workgroup caps it at
  // THREE workgroups per CU: a quarter of every SIMD's registers stays free for them.  Alone the kernel loses 2.5 % (2.38 ->
  // 2.44 ms per 256 pairs: three waves still saturate the SIMD's issue port), the three-stream step gains 1.3 % (measured:
  // 12.57 -> 12.41 ms per 768 pairs; two workgroups per CU: 13.1 ms).
  const size_t lds_pad = ctx->hint_shared_device && ksize == 11 ? 41000 : 0;
  hipLaunchKernelGGL(ksize == 11 ? k11 : (ksize == 5 ? k5 : k3), grid, block, lds_pad, ctx->stream, img, table, nframes, H, W, nimg,
                     rows, cols, strips, row_range, gray);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

// Rows of the gray panoramas that the detector + descriptor stages can reach, per mask set (= view):
//   goodFeaturesToTrack reads the response at masked pixels and their 3x3 neighbours (dilation), the response reads
//   a 3x3 box of 3x3 Sobel products: gray rows [mlo - 3, mhi + 3] for mask rows mlo..mhi;
//   ORB.compute keeps keypoints (integer rows inside the masks) with edge <= y < rows - edge and reads the rotated
//   pattern (radius Rp, computed exactly as orb_describe_kernel does) on the 7x7-blurred image:
//   gray rows [max(edge, mlo) - Rp - 3, min(rows - edge - 1, mhi) + Rp + 3].
// One workgroup per set; out[2 * set] = first row, out[2 * set + 1] = last row + 1 (0, 0 for an empty mask set).
__global__ __launch_bounds__(1024) void gray_rows_kernel(const uint32_t* __restrict__ mask_bits, int rows, int cols,
                                                         int nmask, int edge, const int8_t* __restrict__ pattern,
                                                         float cos_a, float sin_a, int32_t* __restrict__ out) {
  __shared__ int s_lo, s_hi, s_R;
  const int tid = threadIdx.x, set = blockIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) {
    s_lo = rows;
    s_hi = -1;
    s_R = 0;
  }
  __syncthreads();
  const uint32_t mask_all = nmask < 32 ? (1u << nmask) - 1u : 0xFFFFFFFFu;
  const uint32_t* mb = mask_bits + (size_t)set * rows * cols;
  for (int y = wid; y < rows; y += 16) {  // one wave per row, 16 rows in flight: the loads of a row are independent
    uint32_t acc = 0u;
    for (int x = lane; x < cols; x += 64) acc |= mb[(size_t)y * cols + x];
    if (__ballot((acc & mask_all) != 0u) != 0ULL && lane == 0) {
      atomicMin(&s_lo, y);
      atomicMax(&s_hi, y);
    }
  }
  for (int i = tid; i < 512; i += 1024) {
    const float px = (float)pattern[2 * i], py = (float)pattern[2 * i + 1];
    const float xr = (px * cos_a) - (py * sin_a), yr = (px * sin_a) + (py * cos_a);
    atomicMax(&s_R, max(abs(__float2int_rn(xr)), abs(__float2int_rn(yr))));
  }
  __syncthreads();
  if (tid == 0) {
    int lo = 0, hi = 0;
    if (s_hi >= s_lo) {
      lo = s_lo - 3;
      hi = s_hi + 3;
      const int ylo = max(edge, s_lo), yhi = min(rows - edge - 1, s_hi);
      if (yhi >= ylo) {
        lo = min(lo, ylo - s_R - 3);
        hi = max(hi, yhi + s_R + 3);
      }
      lo = max(lo, 0);
      hi = min(hi + 1, rows);
    }
    out[2 * set] = lo;
    out[2 * set + 1] = hi;
  }
}

}  // namespace

extern "C" {

int32_t sosvo_unwrap(sosvo_ctx* ctx, const uint8_t* omni, const uint8_t* masks, const float* map_x,
                     const float* map_y, int32_t nframes, int32_t H, int32_t W, int32_t rows, int32_t cols,
                     uint8_t* pano) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, omni && map_x && map_y && pano, "null pointer");
  SOSVO_REQUIRE(ctx, nframes >= 0 && nframes <= 32767, "nframes out of range");
  SOSVO_REQUIRE(ctx, H > 0 && W > 0 && H <= 16384 && W <= 16384 && rows > 0 && cols > 0 && rows * (int64_t)cols < (1 << 28),
                "image sizes out of range");
  if (nframes == 0) return SOSVO_OK;
  SOSVO_LAUNCH(ctx, unwrap_kernel, dim3(cdiv(rows * cols, kThreads), 2 * nframes), dim3(kThreads), 0, ctx->stream, omni,
               masks, map_x, map_y, nframes, H, W, rows, cols, pano);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_unwrap_prepare(sosvo_ctx* ctx, const uint8_t* masks, const float* map_x, const float* map_y, int32_t H,
                             int32_t W, int32_t rows, int32_t cols, uint32_t* table) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, map_x && map_y && table, "null pointer");
  SOSVO_REQUIRE(ctx, H > 0 && W > 0 && H <= 16384 && W <= 16384 && rows > 0 && cols > 0 && rows * (int64_t)cols < (1 << 28),
                "image sizes out of range");
  SOSVO_REQUIRE(ctx, ((uintptr_t)table & 7) == 0, "table must be 8-byte aligned");
  SOSVO_LAUNCH(ctx, unwrap_table_kernel, dim3(cdiv(rows * cols, kThreads), 2), dim3(kThreads), 0, ctx->stream, masks, map_x,
               map_y, H, W, rows * cols, reinterpret_cast<uint2*>(table));
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_unwrap_table(sosvo_ctx* ctx, const uint8_t* omni, const uint32_t* table, int32_t nframes, int32_t H,
                           int32_t W, int32_t rows, int32_t cols, uint8_t* pano) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, omni && table && pano, "null pointer");
  SOSVO_REQUIRE(ctx, nframes >= 0 && nframes <= 32767, "nframes out of range");
  SOSVO_REQUIRE(ctx, H > 0 && W > 0 && H <= 16384 && W <= 16384 && rows > 0 && cols > 0 && rows * (int64_t)cols < (1 << 28),
                "image sizes out of range");
  SOSVO_REQUIRE(ctx, ((uintptr_t)table & 7) == 0 && ((uintptr_t)pano & 3) == 0, "table / pano alignment");
  if (nframes == 0) return SOSVO_OK;
  SOSVO_LAUNCH(ctx, unwrap_lut_kernel, dim3((unsigned)(cdiv(rows * cols, kThreads) * 8 * cdiv(2 * nframes, 8))), dim3(kThreads), 0,
               ctx->stream, omni, reinterpret_cast<const uint2*>(table), nframes, H, W, rows * cols, pano);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_median_gray(sosvo_ctx* ctx, const uint8_t* img, int32_t nimg, int32_t rows, int32_t cols,
                          int32_t ksize, uint8_t* gray) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, img && gray, "null pointer");
  SOSVO_REQUIRE(ctx, nimg >= 0 && nimg <= 65535 && rows > 0 && cols > 0 && rows * (int64_t)cols < (1 << 28),
                "sizes out of range");
  SOSVO_REQUIRE(ctx, ksize <= 1 || ksize == 3 || ksize == 5 || ksize == 11, "ksize must be 0/1 (none), 3, 5 or 11");
  if (nimg == 0) return SOSVO_OK;
  if (ksize <= 1) {
    const size_t n = (size_t)nimg * rows * cols;
    SOSVO_LAUNCH(ctx, gray_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, img, n,
                 gray);
    SOSVO_LAUNCH_CHECK(ctx);
    return SOSVO_OK;
  }
  return launch_median<false>(ctx, img, nullptr, 0, 0, 0, nimg, rows, cols, ksize, nullptr, gray);
}

int32_t sosvo_gray_rows_needed(sosvo_ctx* ctx, const uint32_t* mask_bits, int32_t nsets, int32_t rows, int32_t cols,
                               int32_t nmask, int32_t edge, const int8_t* pattern, float cos_a, float sin_a,
                               int32_t* row_range) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, mask_bits && pattern && row_range, "null pointer");
  SOSVO_REQUIRE(ctx, nsets >= 1 && nsets <= 65535 && nmask >= 1 && nmask <= 32, "nsets / nmask out of range");
  SOSVO_REQUIRE(ctx, rows > 0 && cols > 0 && rows * (int64_t)cols < (1 << 28) && edge >= 0, "image sizes out of range");
  SOSVO_LAUNCH(ctx, gray_rows_kernel, dim3((unsigned)nsets), dim3(1024), 0, ctx->stream, mask_bits, rows, cols, nmask, edge,
               pattern, cos_a, sin_a, row_range);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_unwrap_median_gray(sosvo_ctx* ctx, const uint8_t* omni, const uint32_t* table, int32_t nframes, int32_t H,
                                 int32_t W, int32_t rows, int32_t cols, int32_t ksize, uint8_t* gray) {
  return sosvo_unwrap_median_gray_rows(ctx, omni, table, nframes, H, W, rows, cols, ksize, nullptr, gray);
}

int32_t sosvo_unwrap_median_gray_rows(sosvo_ctx* ctx, const uint8_t* omni, const uint32_t* table, int32_t nframes, int32_t H,
                                      int32_t W, int32_t rows, int32_t cols, int32_t ksize, const int32_t* row_range,
                                      uint8_t* gray) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, omni && table && gray, "null pointer");
  SOSVO_REQUIRE(ctx, nframes >= 0 && nframes <= 32767, "nframes out of range");
  SOSVO_REQUIRE(ctx, H > 0 && W > 0 && H <= 16384 && W <= 16384 && rows > 0 && cols > 0 && rows * (int64_t)cols < (1 << 28),
                "image sizes out of range");
  SOSVO_REQUIRE(ctx, ((uintptr_t)table & 7) == 0, "table must be 8-byte aligned");
  SOSVO_REQUIRE(ctx, ksize == 0 || ksize == 1 || ksize == 3 || ksize == 5 || ksize == 11, "ksize must be 0/1 (no median), 3, 5 or 11");
  if (nframes == 0) return SOSVO_OK;
  if (ksize <= 1) {  // no median: unwrap straight to gray (every row: nothing is skipped, row_range is not consulted)
    // (tile height 16 or 8: the one that covers the rows with less padding)
    const bool th16 = cdiv(rows, 16) * 16 <= cdiv(rows, 8) * 8;
    const int nblk = cdiv(cols, 64) * cdiv(rows, th16 ? 16 : 8);
    const dim3 grid((unsigned)(nblk * 8 * cdiv(2 * nframes, 8)));
    if (th16)
      SOSVO_LAUNCH(ctx, unwrap_gray_kernel<4>, grid, dim3(kThreads), 0, ctx->stream, omni, reinterpret_cast<const uint2*>(table),
                   nframes, H, W, rows, cols, gray);
    else
      SOSVO_LAUNCH(ctx, unwrap_gray_kernel<2>, grid, dim3(kThreads), 0, ctx->stream, omni, reinterpret_cast<const uint2*>(table),
                   nframes, H, W, rows, cols, gray);
    SOSVO_LAUNCH_CHECK(ctx);
    return SOSVO_OK;
  }
  return launch_median<true>(ctx, omni, reinterpret_cast<const uint2*>(table), nframes, H, W, 2 * nframes, rows, cols, ksize,
                             row_range, gray);
}

}  // extern "C"
