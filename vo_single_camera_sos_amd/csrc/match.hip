// K7 -- brute-force Hamming matching and the stable sort of matches by distance.
//
// Reference call sites: cv2.BFMatcher(NORM_HAMMING).match / knnMatch at
// omnistereo/camera_models.py:442 / :420 and sorted(matches, key=distance) at :444.
//
// Mapping to CDNA4: one query descriptor lives in 8 VGPRs of one lane (QPT queries per
// lane to reuse every LDS read), the train set streams through an 8 KB LDS tile and is
// read back as wave-wide broadcasts (all lanes read the same 32 bytes), distance is
// 8 x (v_xor_b32 + accumulating v_bcnt_u32_b32).  The result is carried as the packed key
// (distance << 20 | train index): an unsigned min over keys is exactly "smallest
// distance, first train index wins", so partial results of train-range splits merge with
// one atomicMin and stay bit-identical to a sequential scan.  No MFMA: there is no
// contraction here, only popcounts.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTrainTile = 256;  // descriptors per LDS tile (8 KB)

__device__ __forceinline__ uint32_t hamming256(const uint4& a0, const uint4& a1, const uint4& b0,
                                               const uint4& b1) {
  uint32_t d = __popc(a0.x ^ b0.x);
  d += __popc(a0.y ^ b0.y);
  d += __popc(a0.z ^ b0.z);
  d += __popc(a0.w ^ b0.w);
  d += __popc(a1.x ^ b1.x);
  d += __popc(a1.y ^ b1.y);
  d += __popc(a1.z ^ b1.z);
  d += __popc(a1.w ^ b1.w);
  return d;
}

template <int QPT, int K>
__global__ __launch_bounds__(kThreads) void match_hamming_kernel(
    const uint4* __restrict__ q_desc, const uint4* __restrict__ t_desc, const int32_t* __restrict__ nq,
    const int32_t* __restrict__ nt, const int32_t* __restrict__ q_slot, const int32_t* __restrict__ t_slot,
    int q_stride, int t_stride, int nsplit, uint32_t* __restrict__ keys) {
  SOSVO_LATENCY_BOUND_PRIO();
  __shared__ uint4 tile[kTrainTile * 2];
  const int tid = threadIdx.x;
  // XCD-aware grid: workgroups go round-robin over the 8 XCDs by linear id, so the problem index (all
  // workgroups busy) is the fastest dimension and the query tile (early exit beyond nq) the second.
  const int p = blockIdx.x;
  const int qs = q_slot ? q_slot[p] : p;  // which block of query rows / counts this problem uses
  const int ts = t_slot ? t_slot[p] : p;
  const int nqp = min(nq[qs], q_stride);
  const int ntp = min(nt[ts], t_stride);
  const int q0 = blockIdx.y * (kThreads * QPT);
  if (q0 >= nqp) return;  // uniform over the workgroup

  // train range of this split, whole tiles
  const int tiles = (ntp + kTrainTile - 1) / kTrainTile;
  const int tiles_per = (tiles + nsplit - 1) / nsplit;
  const int tb = blockIdx.z * tiles_per * kTrainTile;
  const int te = min(ntp, tb + tiles_per * kTrainTile);
  if (nsplit > 1 && tb >= te) return;  // keys were pre-set to NONE

  uint4 qa[QPT], qb[QPT];
  uint32_t best[QPT], second[QPT];
  // One query per lane (the small per-bucket problems): a wave whose 64 rows all lie beyond nq sits out the distance
  // loop (wave-uniform, scalar test per train tile); it still loads its share of the tile.  With 4 queries per lane the
  // same test inside the unrolled loop costs more than the ragged tail it saves (measured), so it is not applied there.
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bool live[QPT];
#pragma unroll
  for (int r = 0; r < QPT; ++r) {
    live[r] = q0 + r * kThreads + wave * 64 < nqp;
    const int qi = q0 + r * kThreads + tid;
    const bool valid = qi < nqp;
    const size_t row = (size_t)qs * q_stride + (valid ? qi : q0);
    qa[r] = q_desc[row * 2 + 0];
    qb[r] = q_desc[row * 2 + 1];
    best[r] = SOSVO_KEY_NONE;
    second[r] = SOSVO_KEY_NONE;
  }

  const uint4* tsrc = t_desc + (size_t)ts * t_stride * 2;
  for (int t0 = tb; t0 < te; t0 += kTrainTile) {
    const int lim = min(kTrainTile, te - t0);
    __syncthreads();
    for (int i = tid; i < lim * 2; i += kThreads) tile[i] = tsrc[(size_t)t0 * 2 + i];
    __syncthreads();
    if (QPT == 1 && !live[0]) continue;
#pragma unroll 4
    for (int j = 0; j < lim; ++j) {
      const uint4 ta = tile[2 * j + 0];
      const uint4 tb4 = tile[2 * j + 1];
      const uint32_t tj = (uint32_t)(t0 + j);
#pragma unroll
      for (int r = 0; r < QPT; ++r) {
        const uint32_t key = (hamming256(qa[r], qb[r], ta, tb4) << SOSVO_KEY_SHIFT) | tj;
        if (K == 2) second[r] = min(second[r], max(best[r], key));
        best[r] = min(best[r], key);
      }
    }
  }

#pragma unroll
  for (int r = 0; r < QPT; ++r) {
    const int qi = q0 + r * kThreads + tid;
    if (qi < nqp) {
      uint32_t* out = keys + ((size_t)p * q_stride + qi) * K;
      if (K == 1 && nsplit > 1) {
        atomicMin(out, best[r]);
      } else {
        out[0] = best[r];
        if (K == 2) out[1] = second[r];
      }
    }
  }
}

// ---- float descriptors (SIFT / SURF / KAZE ...): brute-force L2, cv2.BFMatcher() with its default norm
// (omnistereo/camera_models.py:396).  Not on the VO drivers' path (they use binary descriptors): one wave per 64
// queries, the query rows transposed in LDS (lane-contiguous: conflict-free), the train rows in an LDS tile read as
// broadcasts; the squared differences accumulate in float32 four at a time in index order, as OpenCV's scalar
// normL2Sqr_ does (no contraction), distance = sqrt.  Key = (distance bits << 32 | train index): non-negative floats
// order like their bit patterns, so an unsigned min is "smallest distance, first train index".
constexpr int kL2MaxDim = 128, kL2Tile = 32;
template <int K>
__global__ __launch_bounds__(64) void match_l2_kernel(const float* __restrict__ q, const float* __restrict__ t,
                                                      const int32_t* __restrict__ nq, const int32_t* __restrict__ nt,
                                                      int q_stride, int t_stride, int dim,
                                                      unsigned long long* __restrict__ keys) {
  __shared__ float qs[kL2MaxDim * 64];
  __shared__ __attribute__((aligned(16))) float tile[kL2Tile * kL2MaxDim];
  const int lane = threadIdx.x, p = blockIdx.x;
  const int nqp = min(nq[p], q_stride), ntp = min(nt[p], t_stride);
  const int q0 = blockIdx.y * 64;
  if (q0 >= nqp) {  // uniform: padding rows only -- "absent" is all ones, as include/sosvo.h says
    if (q0 + lane < q_stride)
      for (int j = 0; j < K; ++j) keys[((size_t)p * q_stride + q0 + lane) * K + j] = ~0ULL;
    return;
  }
  const int dimp = (dim + 3) & ~3;  // row pitch in the tile
  const float* qsrc = q + ((size_t)p * q_stride + q0) * dim;
  const int rows = min(64, nqp - q0);
  for (int i = lane; i < rows * dim; i += 64) {  // coalesced read, transposed store
    const int r = i / dim, d = i - r * dim;
    qs[d * 64 + r] = qsrc[i];
  }
  unsigned long long best = ~0ULL, second = ~0ULL;
  const float* tsrc = t + (size_t)p * t_stride * dim;
  for (int t0 = 0; t0 < ntp; t0 += kL2Tile) {
    const int lim = min(kL2Tile, ntp - t0);
    __syncthreads();
    for (int i = lane; i < lim * dim; i += 64) {
      const int r = i / dim, d = i - r * dim;
      tile[r * dimp + d] = tsrc[(size_t)t0 * dim + i];
    }
    __syncthreads();
    if (lane >= rows) continue;  // (the barriers above are reached by every lane)
    for (int j = 0; j < lim; ++j) {
      const float* tr = tile + j * dimp;
      float s = 0.0f;
      int i = 0;
      for (; i <= dim - 4; i += 4) {
        const float4 tv = *reinterpret_cast<const float4*>(tr + i);
        const float v0 = qs[i * 64 + lane] - tv.x, v1 = qs[(i + 1) * 64 + lane] - tv.y;
        const float v2 = qs[(i + 2) * 64 + lane] - tv.z, v3 = qs[(i + 3) * 64 + lane] - tv.w;
        s += (((v0 * v0) + (v1 * v1)) + (v2 * v2)) + (v3 * v3);
      }
      for (; i < dim; ++i) {
        const float v = qs[i * 64 + lane] - tr[i];
        s += v * v;
      }
      const unsigned long long key = ((unsigned long long)__float_as_uint(sqrtf(s)) << 32) | (uint32_t)(t0 + j);
      if (K == 2) second = min(second, max(best, key));
      best = min(best, key);
    }
  }
  if (lane < rows || q0 + lane < q_stride) {  // (rows past the query count keep best = second = all ones)
    unsigned long long* out = keys + ((size_t)p * q_stride + q0 + lane) * K;
    out[0] = best;
    if (K == 2) out[1] = second;
  }
}

// Rank sort: the sort key of query i is (distance << 20 | i); its rank is the number of
// smaller sort keys.  Keys are unique, so ranks are a permutation and ties on distance
// keep query order (stable), as Python's sorted() does at camera_models.py:444.
__global__ __launch_bounds__(kThreads) void sort_matches_kernel(const uint32_t* __restrict__ keys,
                                                                const int32_t* __restrict__ nq,
                                                                const int32_t* __restrict__ q_slot, int q_stride,
                                                                int32_t* __restrict__ order) {
  SOSVO_LATENCY_BOUND_PRIO();
  extern __shared__ uint32_t sk[];
  const int tid = threadIdx.x;
  const int p = blockIdx.x;  // problem fastest, tile second (see match_hamming_kernel)
  const int n = min(nq[q_slot ? q_slot[p] : p], q_stride);
  const int i0 = blockIdx.y * kThreads;
  if (i0 >= n) return;
  const int n4 = (n + 3) & ~3;
  const uint32_t* src = keys + (size_t)p * q_stride;
  for (int i = tid; i < n4; i += kThreads)
    sk[i] = i < n ? ((src[i] & ~SOSVO_KEY_IDX_MASK) | (uint32_t)i) : 0xFFFFFFFFu;
  __syncthreads();
  const int i = i0 + tid;
  const uint32_t mine = i < n ? sk[i] : 0u;
  int rank = 0;
  const uint4* sk4 = reinterpret_cast<const uint4*>(sk);
  for (int j = 0; j < n4 / 4; ++j) {
    const uint4 v = sk4[j];
    rank += (v.x < mine) + (v.y < mine) + (v.z < mine) + (v.w < mine);
  }
  if (i < n) order[(size_t)p * q_stride + rank] = i;
}

// ---- radius match ------------------------------------------------------------------------------------------
// cv2.BFMatcher.radiusMatch (omnistereo/camera_models.py:413, FeatureMatcher.use_radius_match): for every query
// ALL train descriptors within max_distance.  One wave per query: the query sits in 8 VGPRs of every lane, lane l
// tests train rows l, l + 64, ...; hits are appended to the query's LDS list in train order through a ballot
// prefix, and the list (at most `cap` keys) leaves sorted by a rank sort on the packed key (distance, then train
// index).  When more than cap rows match, the cap smallest keys are kept by replacing the list's current maximum.
constexpr int kRadiusCapMax = 512;

__global__ __launch_bounds__(kThreads) void match_radius_kernel(
    const uint4* __restrict__ q_desc, const uint4* __restrict__ t_desc, const int32_t* __restrict__ nq,
    const int32_t* __restrict__ nt, const int32_t* __restrict__ q_slot, const int32_t* __restrict__ t_slot,
    int q_stride, int t_stride, uint32_t max_distance, int cap, uint32_t* __restrict__ keys,
    int32_t* __restrict__ counts) {
  __shared__ uint32_t lists[kThreads / 64][kRadiusCapMax];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int p = blockIdx.x;
  const int qs = q_slot ? q_slot[p] : p, ts = t_slot ? t_slot[p] : p;
  const int nqp = min(nq[qs], q_stride), ntp = min(nt[ts], t_stride);
  const int i = blockIdx.y * (kThreads / 64) + wid;  // wave-uniform
  if (i >= nqp) return;
  uint32_t* list = lists[wid];
  const uint4 qa = q_desc[((size_t)qs * q_stride + i) * 2], qb = q_desc[((size_t)qs * q_stride + i) * 2 + 1];
  const uint4* tb = t_desc + (size_t)ts * t_stride * 2;
  int total = 0, kept = 0;  // wave-uniform
  for (int j0 = 0; j0 < ntp; j0 += 64) {
    const int j = j0 + lane;
    bool hit = false;
    uint32_t key = SOSVO_KEY_NONE;
    if (j < ntp) {
      const uint32_t d = hamming256(qa, qb, tb[2 * (size_t)j], tb[2 * (size_t)j + 1]);
      hit = d <= max_distance;
      key = (d << SOSVO_KEY_SHIFT) | (uint32_t)j;
    }
    unsigned long long bal = __ballot(hit);
    if (!bal) continue;
    const int nh = __popcll(bal);
    total += nh;
    if (kept + nh <= cap) {
      if (hit) list[kept + __popcll(bal & ((1ULL << lane) - 1ULL))] = key;
      kept += nh;
    } else {  // overflow: one hit at a time, each replaces the list's maximum if smaller (rare)
      while (bal) {
        const int src = __ffsll((long long)bal) - 1;
        bal &= bal - 1ULL;
        const uint32_t k1 = (uint32_t)__shfl((int)key, src);
        if (kept < cap) {
          if (lane == 0) list[kept] = k1;
          kept++;
        } else {
          uint32_t mx = 0u;
          int mi = 0;
          for (int e = lane; e < cap; e += 64) {
            const uint32_t v = list[e];
            if (v > mx) {
              mx = v;
              mi = e;
            }
          }
          for (int s = 32; s > 0; s >>= 1) {
            const uint32_t ov = (uint32_t)__shfl_down((int)mx, s);
            const int oi = __shfl_down(mi, s);
            if (ov > mx) {
              mx = ov;
              mi = oi;
            }
          }
          mx = (uint32_t)__shfl((int)mx, 0);
          mi = __shfl(mi, 0);
          if (lane == 0 && k1 < mx) list[mi] = k1;
        }
      }
    }
  }
  // rank sort of the kept keys (unique: the train index is part of the key), KEY_NONE padding
  uint32_t* out = keys + ((size_t)p * q_stride + i) * cap;
  for (int e = lane; e < cap; e += 64) {
    if (e >= kept) out[e] = SOSVO_KEY_NONE;
  }
  for (int e = lane; e < kept; e += 64) {
    const uint32_t v = list[e];
    int rank = 0;
    for (int o = 0; o < kept; ++o) rank += list[o] < v ? 1 : 0;
    out[rank] = v;
  }
  if (lane == 0) counts[(size_t)p * q_stride + i] = total;
}

}  // namespace

extern "C" {

int32_t sosvo_match_hamming(sosvo_ctx* ctx, const uint8_t* q_desc, const uint8_t* t_desc,
                            const int32_t* nq, const int32_t* nt, const int32_t* q_slot,
                            const int32_t* t_slot, int32_t nprob, int32_t q_stride, int32_t t_stride, int32_t k,
                            uint32_t* keys) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, q_desc && t_desc && nq && nt && keys, "null pointer");
  SOSVO_REQUIRE(ctx, k == 1 || k == 2, "k must be 1 or 2");
  SOSVO_REQUIRE(ctx, nprob >= 0 && nprob <= 65535, "nprob out of range");
  SOSVO_REQUIRE(ctx, q_stride > 0 && q_stride <= (1 << SOSVO_KEY_SHIFT), "q_stride out of range");
  SOSVO_REQUIRE(ctx, t_stride > 0 && t_stride <= (1 << SOSVO_KEY_SHIFT), "t_stride out of range");
  SOSVO_REQUIRE(ctx, ((uintptr_t)q_desc & 15) == 0 && ((uintptr_t)t_desc & 15) == 0,
                "descriptor arrays must be 16-byte aligned");
  if (nprob == 0) return SOSVO_OK;
  const uint4* q4 = reinterpret_cast<const uint4*>(q_desc);
  const uint4* t4 = reinterpret_cast<const uint4*>(t_desc);

  // Queries per lane: 4 when there is enough work to still fill the chip, else 1.
  const int qpt = (q_stride >= 1024) ? 4 : 1;
  const int gx = cdiv(q_stride, kThreads * qpt);
  // Split the train range over grid.z until ~2k workgroups exist (1-NN only: the split
  // results merge with atomicMin on the packed key).
  int nsplit = 1;
  if (k == 1) {
    const int tiles = cdiv(t_stride, kTrainTile);
    nsplit = cdiv(2048, gx * nprob);
    if (nsplit > tiles) nsplit = tiles;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 64) nsplit = 64;
  }
  if (nsplit > 1)
    SOSVO_HIP(ctx, hipMemsetAsync(keys, 0xFF, (size_t)nprob * q_stride * sizeof(uint32_t), ctx->stream));
  dim3 grid(nprob, gx, nsplit), block(kThreads);
  if (k == 1) {
    if (qpt == 4)
      SOSVO_LAUNCH(ctx,(match_hamming_kernel<4, 1>), grid, block, 0, ctx->stream, q4, t4, nq, nt,
                         q_slot, t_slot, q_stride, t_stride, nsplit, keys);
    else
      SOSVO_LAUNCH(ctx,(match_hamming_kernel<1, 1>), grid, block, 0, ctx->stream, q4, t4, nq, nt,
                         q_slot, t_slot, q_stride, t_stride, nsplit, keys);
  } else {
    if (qpt == 4)
      SOSVO_LAUNCH(ctx,(match_hamming_kernel<4, 2>), grid, block, 0, ctx->stream, q4, t4, nq, nt,
                         q_slot, t_slot, q_stride, t_stride, nsplit, keys);
    else
      SOSVO_LAUNCH(ctx,(match_hamming_kernel<1, 2>), grid, block, 0, ctx->stream, q4, t4, nq, nt,
                         q_slot, t_slot, q_stride, t_stride, nsplit, keys);
  }
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_sort_matches(sosvo_ctx* ctx, const uint32_t* keys, const int32_t* nq, const int32_t* q_slot,
                           int32_t nprob, int32_t q_stride, int32_t* order) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, keys && nq && order, "null pointer");
  SOSVO_REQUIRE(ctx, nprob >= 0 && nprob <= 65535, "nprob out of range");
  SOSVO_REQUIRE(ctx, q_stride > 0 && q_stride <= 16384, "q_stride out of range (max 16384)");
  if (nprob == 0) return SOSVO_OK;
  const size_t lds = (size_t)((q_stride + 3) & ~3) * sizeof(uint32_t);
  dim3 grid(nprob, cdiv(q_stride, kThreads)), block(kThreads);
  SOSVO_LAUNCH(ctx,sort_matches_kernel, grid, block, lds, ctx->stream, keys, nq, q_slot, q_stride, order);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_match_radius(sosvo_ctx* ctx, const uint8_t* q_desc, const uint8_t* t_desc, const int32_t* nq,
                           const int32_t* nt, const int32_t* q_slot, const int32_t* t_slot, int32_t nprob,
                           int32_t q_stride, int32_t t_stride, int32_t max_distance, int32_t cap, uint32_t* keys,
                           int32_t* counts) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, q_desc && t_desc && nq && nt && keys && counts, "null pointer");
  SOSVO_REQUIRE(ctx, nprob >= 0 && nprob <= 65535, "nprob out of range");
  SOSVO_REQUIRE(ctx, q_stride > 0 && q_stride <= (1 << SOSVO_KEY_SHIFT) && t_stride > 0 && t_stride <= (1 << SOSVO_KEY_SHIFT),
                "strides out of range");
  SOSVO_REQUIRE(ctx, cap > 0 && cap <= kRadiusCapMax, "cap out of range (1..512)");
  SOSVO_REQUIRE(ctx, max_distance >= 0, "max_distance must be >= 0");
  SOSVO_REQUIRE(ctx, (((uintptr_t)q_desc | (uintptr_t)t_desc) & 15) == 0, "descriptors must be 16-byte aligned");
  if (nprob == 0) return SOSVO_OK;
  SOSVO_LAUNCH(ctx, match_radius_kernel, dim3(nprob, cdiv(q_stride, kThreads / 64)), dim3(kThreads), 0, ctx->stream,
               reinterpret_cast<const uint4*>(q_desc), reinterpret_cast<const uint4*>(t_desc), nq, nt, q_slot, t_slot,
               q_stride, t_stride, (uint32_t)max_distance, cap, keys, counts);
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

int32_t sosvo_match_l2(sosvo_ctx* ctx, const float* q_desc, const float* t_desc, const int32_t* nq, const int32_t* nt,
                       int32_t nprob, int32_t q_stride, int32_t t_stride, int32_t dim, int32_t k, uint64_t* keys) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, q_desc && t_desc && nq && nt && keys, "null pointer");
  SOSVO_REQUIRE(ctx, nprob >= 0 && nprob <= 65535, "nprob out of range");
  SOSVO_REQUIRE(ctx, q_stride > 0 && q_stride <= (1 << 20) && t_stride > 0 && t_stride <= (1 << 20), "strides out of range");
  SOSVO_REQUIRE(ctx, dim >= 1 && dim <= kL2MaxDim, "dim out of range (1..128)");
  SOSVO_REQUIRE(ctx, k == 1 || k == 2, "k must be 1 or 2");
  if (nprob == 0) return SOSVO_OK;
  if (k == 1)
    SOSVO_LAUNCH(ctx, match_l2_kernel<1>, dim3(nprob, cdiv(q_stride, 64)), dim3(64), 0, ctx->stream, q_desc, t_desc, nq, nt,
                 q_stride, t_stride, dim, reinterpret_cast<unsigned long long*>(keys));
  else
    SOSVO_LAUNCH(ctx, match_l2_kernel<2>, dim3(nprob, cdiv(q_stride, 64)), dim3(64), 0, ctx->stream, q_desc, t_desc, nq, nt,
                 q_stride, t_stride, dim, reinterpret_cast<unsigned long long*>(keys));
  SOSVO_LAUNCH_CHECK(ctx);
  return SOSVO_OK;
}

}  // extern "C"
