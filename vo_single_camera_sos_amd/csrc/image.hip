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
// The float maps, the 1/32-px rounding, the border test and the annulus mask depend only on the model, so
// they are folded once into a packed table: word0 = (iy << 16) | (ix & 0xFFFF) (int16 each), word1 =
// fx | fy << 5 | valid << 10 (4 bits: tap t in-bounds and unmasked).  The per-frame kernel then does no
// float work and no mask loads: one 8-byte table load and four 4-byte tap loads per pixel, four pixels per
// lane so that the 12 output bytes leave as three aligned dwords.
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
    e.x = ((uint32_t)(iy & 0xFFFF) << 16) | (uint32_t)(ix & 0xFFFF);
    e.y = (uint32_t)fx | ((uint32_t)fy << 5) | (valid << 10);
  }
  table[(size_t)view * npix + pix] = e;
}

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef unsigned long long __attribute__((aligned(1))) u64_unaligned;

// One panorama pixel per lane (neighbouring lanes = neighbouring pixels, so a wave's taps fall on a short arc
// of the omni image: few cache lines per load instruction).  The two taps of a source row are adjacent pixels
// = 6 consecutive bytes: ONE unaligned 8-byte load per row.  The 768 output bytes of a workgroup are staged
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
  const size_t npx_src = (size_t)H * W;
  int acc0 = 0, acc1 = 0, acc2 = 0;
  if (pix < npix) {
    const uint2 e = table[(size_t)view * npix + pix];
    const uint32_t valid = e.y >> 10;
    if (valid) {
      const int ix = (int)(int16_t)(e.x & 0xFFFFu), iy = (int)(int16_t)(e.x >> 16);
      const int fx = (int)(e.y & 31u), fy = (int)((e.y >> 5) & 31u);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const uint32_t vr = (valid >> (2 * r)) & 3u;  // bit 0: tap (ix, iy + r), bit 1: tap (ix + 1, iy + r)
        if (!vr) continue;
        const int wy = r ? fy : 32 - fy;
        const long long o = (long long)(iy + r) * W + ix;  // may be -1 when only the right tap is inside
        unsigned long long v = 0ULL;
        if (o >= 0 && (size_t)o + 3 <= npx_src) {  // 8 bytes from 3 * o stay inside the frame
          v = *reinterpret_cast<const u64_unaligned*>(src + 3 * o);
        } else {
          if (vr & 1u) v |= (unsigned long long)src[3 * o] | ((unsigned long long)src[3 * o + 1] << 8) | ((unsigned long long)src[3 * o + 2] << 16);
          if (vr & 2u) v |= ((unsigned long long)src[3 * o + 3] << 24) | ((unsigned long long)src[3 * o + 4] << 32) | ((unsigned long long)src[3 * o + 5] << 40);
        }
        if (vr & 1u) {
          const int w = (32 - fx) * wy;
          acc0 += w * (int)(v & 0xFFu);
          acc1 += w * (int)((v >> 8) & 0xFFu);
          acc2 += w * (int)((v >> 16) & 0xFFu);
        }
        if (vr & 2u) {
          const int w = fx * wy;
          acc0 += w * (int)((v >> 24) & 0xFFu);
          acc1 += w * (int)((v >> 32) & 0xFFu);
          acc2 += w * (int)((v >> 40) & 0xFFu);
        }
      }
    }
  }
  const uint32_t b0 = (uint32_t)((acc0 + 512) >> 10), b1 = (uint32_t)((acc1 + 512) >> 10), b2 = (uint32_t)((acc2 + 512) >> 10);
  const size_t out0 = ((size_t)img * npix + (size_t)blk * kThreads) * 3;  // first output byte of this workgroup
  const int nvalid = min(kThreads, npix - blk * kThreads);
  if ((out0 & 3) == 0 && nvalid == kThreads) {
    uint8_t* sb = reinterpret_cast<uint8_t*>(stage);
    sb[3 * tid + 0] = (uint8_t)b0;
    sb[3 * tid + 1] = (uint8_t)b1;
    sb[3 * tid + 2] = (uint8_t)b2;
    __syncthreads();
    if (tid < kThreads * 3 / 4) reinterpret_cast<uint32_t*>(pano + out0)[tid] = stage[tid];
  } else if (pix < npix) {
    uint8_t* out = pano + out0 + 3 * tid;
    out[0] = (uint8_t)b0;
    out[1] = (uint8_t)b1;
    out[2] = (uint8_t)b2;
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

// ---- K2 + K3 ----------------------------------------------------------------------------------------
// K x K median per channel (replicated border) followed by BGR->gray.  K odd, 3 <= K <= 15.
template <int K>
__global__ __launch_bounds__(kThreads) void median_gray_kernel(const uint8_t* __restrict__ img, int nimg, int rows,
                                                               int cols, int strips, uint8_t* __restrict__ gray) {
  constexpr int R = K / 2;
  constexpr int NB = K * K;               // window bits per bit-plane, rows packed back to back (oldest row first)
  constexpr int NW = (NB + 31) / 32;      // words per bit-plane: 4 for 11 x 11
  constexpr int POS = (K - 1) * K - 32 * (NW - 1);  // where the newest row goes inside the last word
  static_assert(POS >= 0 && POS + K <= 32, "newest row must not straddle words");
  constexpr int OUTW = 64 - 2 * R;        // output columns per wave
  constexpr uint32_t FIELD = (1u << K) - 1u;
  constexpr uint32_t LASTMASK = (NB - 32 * (NW - 1)) == 32 ? 0xFFFFFFFFu : ((1u << (NB - 32 * (NW - 1))) - 1u);
  constexpr int HALF = NB / 2 + 1;        // rank of the median, 1-based
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kThreads + threadIdx.x) >> 6;
  if (wave >= nimg * strips) return;  // wave-uniform
  const int im = wave / strips, strip = wave - im * strips;
  const int x0 = strip * OUTW;
  const int x_src = clampi(x0 - R + lane, 0, cols - 1);
  const int x_out = x0 + lane;
  const bool out_ok = lane < OUTW && x_out < cols;
  const uint8_t* src = img + (size_t)im * rows * cols * 3;
  uint8_t* dst = gray + (size_t)im * rows * cols;

  uint32_t Wp[3][8][NW];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int j = 0; j < NW; ++j) Wp[c][b][j] = 0u;

  for (int r_src = -R; r_src < rows + R; ++r_src) {
    const int rr = clampi(r_src, 0, rows - 1);
    const uint8_t* px = src + ((size_t)rr * cols + x_src) * 3;
    const uint32_t pix[3] = {px[0], px[1], px[2]};
    // shift the window up by one row and append the new row's 8 bit-plane fields
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long bal = __ballot((pix[c] >> b) & 1u);
        const uint32_t field = (uint32_t)(bal >> lane) & FIELD;
        // drop the oldest row (K bits) of the NB-bit window, append the new one at the top
#pragma unroll
        for (int j = 0; j + 1 < NW; ++j) Wp[c][b][j] = __funnelshift_r(Wp[c][b][j], Wp[c][b][j + 1], K);
        Wp[c][b][NW - 1] = (Wp[c][b][NW - 1] >> K) | (field << POS);
      }
    }
    const int r_out = r_src - R;
    if (r_out < 0) continue;  // window not complete yet (uniform)
    int med[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      uint32_t C[NW];
#pragma unroll
      for (int j = 0; j < NW; ++j) C[j] = 0xFFFFFFFFu;
      C[NW - 1] = LASTMASK;
      int k = HALF, cntC = NB, res = 0;
#pragma unroll
      for (int b = 7; b >= 0; --b) {
        uint32_t t[NW];
        int n1 = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          t[j] = C[j] & Wp[c][b][j];
          n1 += __popc(t[j]);
        }
        const int nz = cntC - n1;           // candidates whose bit b is 0
        const bool take0 = k <= nz;          // the k-th smallest is among them
        res |= take0 ? 0 : (1 << b);
        k = take0 ? k : k - nz;
        cntC = take0 ? nz : n1;
#pragma unroll
        for (int j = 0; j < NW; ++j) C[j] = take0 ? (C[j] ^ t[j]) : t[j];
      }
      med[c] = res;
    }
    if (out_ok) dst[(size_t)r_out * cols + x_out] = bgr2gray(med[0], med[1], med[2]);
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
  } else {
    const int outw = 64 - 2 * (ksize / 2);
    const int strips = cdiv(cols, outw);
    const int waves = nimg * strips;
    dim3 grid(cdiv(waves, kThreads / 64)), block(kThreads);
    if (ksize == 11)
      SOSVO_LAUNCH(ctx, median_gray_kernel<11>, grid, block, 0, ctx->stream, img, nimg, rows, cols, strips, gray);
    else if (ksize == 5)
      SOSVO_LAUNCH(ctx, median_gray_kernel<5>, grid, block, 0, ctx->stream, img, nimg, rows, cols, strips, gray);
    else
      SOSVO_LAUNCH(ctx, median_gray_kernel<3>, grid, block, 0, ctx->stream, img, nimg, rows, cols, strips, gray);
  }
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

}  // extern "C"
