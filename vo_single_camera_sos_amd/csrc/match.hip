// K7 -- brute-force Hamming matching and the stable sort of matches by distance.
//
// Reference call sites: cv2.BFMatcher(NORM_HAMMING).match / knnMatch at
// omnistereo/camera_models.py:442 / :420 and sorted(matches, key=distance) at :444.
//
// Mapping to CDNA4: the 1-NN / 2-NN search is an exact integer contraction on the MATRIX cores
// (hamming = |q| + |t| - 2 q.t on descriptors unpacked to one byte per bit, v_mfma_i32_32x32x32_i8; see
// match_hamming_mfma_kernel) -- the step around it is VALU-issue bound, the matrix pipe was idle.  The result is
// carried as the packed key (distance << 20 | train index): an unsigned min over keys is exactly "smallest
// distance, first train index wins", so partial results of train-range splits merge with
// one atomicMin and stay bit-identical to a sequential scan.  Round 1-2 ran this stage on the VALU (one query per
// lane in 8 VGPRs, train tile in LDS read as broadcasts, 8 x (v_xor + v_bcnt) per pair: 18 VALU wave-instructions per
// 64 pairs against ~2.5 now); the radius search (not on the VO path) still does.
#include "common.h"

namespace {

constexpr int kThreads = 256;

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

// ---- K7 on the matrix pipe ------------------------------------------------------------------------------
// hamming(q, t) = |q| + |t| - 2 q.t on 0/1 vectors: the pair distances of a (32 trains x 32 queries) tile are ONE exact
// integer contraction over K = 256, i.e. eight v_mfma_i32_32x32x32_i8 on descriptors unpacked to one byte per bit.  The
// step around this kernel is VALU-issue bound and MFMA issues beside VALU, so the 8 x (xor + popcount) per pair of the
// round-2 kernel (18 VALU wave-instructions per 64 pairs) shrink to HALF an instruction per pair-element, and the
// contraction itself runs on the otherwise idle matrix cores:
//   operand values: a train bit is the byte 16, a query bit the byte -16, so the contraction yields -256 q.t = (-2 q.t) << 7;
//     the accumulator is not cleared but STARTS at base7[row] = |t| << 7 | row-in-tile (the 16 accumulator registers of a
//     lane are 4 x ds_read_b128 of that LDS table), so after the eight MFMAs every accumulator register already IS the
//     tile-local partial key ((|t| - 2 q.t) << 7) + row -- ordered like the packed key (|q| is constant per lane): the
//     epilogue is one v_min3_i32 per TWO pair-elements.  Once per 128-train tile the winner is widened to
//     ((|t| - 2 q.t) << 20) + train index; |q| << 20 is added at the very end.  Rows past the train count start at 2^24.
//   unpacking: any permutation of the 256 bit positions applied to both sides keeps q.t, so the cheapest one is used:
//     dword w of a descriptor -> K-block w; lane half h (= lane / 32) takes the bits {s + 4 h + 8 b}: operand dword s =
//     ((h ? x : x << 4) >> s) & 0x10101010, byte b.  A operands (trains): unpacked once per workgroup into an LDS tile (row
//     pitch 272 B: conflict-free ds_read_b128 across the 16-lane groups); B operands (queries): 32 VGPRs per 32-query tile.
//   accumulator r of lane l is (train row 8 (r / 4) + 4 (l / 32) + r % 4, query column l % 32).
// Built with -mllvm -amdgpu-mfma-vgpr-form (csrc/Makefile): the accumulators live in VGPRs, no v_accvgpr_read per element.
// KEEP THE MFMA SEQUENCE BRANCH-FREE.  A first version with a wave-uniform `if` around every MFMA returned a wrong key in
// ~8 % of launches under load.  Cause (found from the ISA in round 4): a result of the 8-pass
// v_mfma_i32_32x32x32_i8 may not be touched by a VALU / LDS / VMEM instruction for 12 wait states, there is no hardware
// interlock, and ROCm 7.2's hazard recogniser measures that distance wrongly across the if-diamonds conditional MFMAs
// create (a block first reached through the longer path is not revisited through the shorter one): such forms compile to
// paths with 6 - 10 wait states instead of 12, taken exactly when a wave's second query tile is dead.  Too few wait
// states is a timing-dependent stale read -- hence "only under load".  scripts/check_mfma_hazards.py measures every path
// of this file's ISA itself and tests/test_mfma_isa.py fails the CPU suite when an edit brings such a path (or
// v_accvgpr moves, or a branch inside an accumulation chain) back.
constexpr int kMmaTrainTile = 128;                  // trains per LDS tile
constexpr int kMmaRowPitch = 272;                   // bytes per unpacked train row (256 + 16)
constexpr int32_t kMmaInvalid7 = 1 << 24;           // base7 of a row without a train
constexpr int32_t kMmaInvalid = 1 << 30;            // "no train seen yet" in the wide partial key
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

// the two smallest of {b, s, k} given b <= s
__device__ __forceinline__ void two_smallest(int32_t& b, int32_t& s, int32_t k) {
  s = max(min(b, s), min(max(b, s), k));  // median of three (v_med3_i32)
  b = min(b, k);
}

template <int QT, int K>
__global__ __launch_bounds__(kThreads) void match_hamming_mfma_kernel(
    const uint4* __restrict__ q_desc, const uint4* __restrict__ t_desc, const int32_t* __restrict__ nq,
    const int32_t* __restrict__ nt, const int32_t* __restrict__ q_slot, const int32_t* __restrict__ t_slot,
    int q_stride, int t_stride, int nsplit, uint32_t* __restrict__ keys) {
  SOSVO_LATENCY_BOUND_PRIO();
  __shared__ __attribute__((aligned(16))) uint8_t tile[kMmaTrainTile * kMmaRowPitch];
  __shared__ __attribute__((aligned(16))) int32_t base7[kMmaTrainTile];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int p = blockIdx.x;
  const int qs = q_slot ? q_slot[p] : p, ts = t_slot ? t_slot[p] : p;
  const int nqp = min(nq[qs], q_stride), ntp = min(nt[ts], t_stride);
  constexpr int kQPerWg = (kThreads / 64) * 32 * QT;
  const int q0 = blockIdx.y * kQPerWg;
  if (q0 >= nqp) return;  // uniform over the workgroup
  const int tiles = (ntp + kMmaTrainTile - 1) / kMmaTrainTile;
  const int tiles_per = (tiles + nsplit - 1) / nsplit;
  const int tb = blockIdx.z * tiles_per * kMmaTrainTile;
  const int te = min(ntp, tb + tiles_per * kMmaTrainTile);
  if (nsplit > 1 && tb >= te) return;  // keys were pre-set to NONE

  // ---- this wave's queries: unpacked B operands (-16 per set bit), resident
  v4i32 bq[QT][8];
  int32_t pq[QT];
  int32_t best[QT], second[QT];
  // (a wave whose first tile lies beyond the query count sits out the contraction; a dead SECOND tile is computed on a
  // repeated row and never stored: no branch inside the MFMA sequence)
  const bool wave_live = q0 + wave * QT * 32 < nqp;  // wave-uniform
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = q0 + (wave * QT + t) * 32 + col;
    const size_t row = (size_t)qs * q_stride + (qi < nqp ? qi : q0);
    const uint4 a = q_desc[row * 2 + 0], b = q_desc[row * 2 + 1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    int32_t pc = 0;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      pc += __popc(w[kb]);
      const uint32_t y = half ? w[kb] : w[kb] << 4;
      uint32_t n[4];
#pragma unroll
      for (int sft = 0; sft < 4; ++sft) {
        const uint32_t m = (y >> sft) & 0x10101010u;
        n[sft] = (m << 4) - m;  // 0x10 -> 0xF0 = -16 per byte (no carry between bytes)
      }
      bq[t][kb] = v4i32{(int)n[0], (int)n[1], (int)n[2], (int)n[3]};
    }
    pq[t] = pc;
    best[t] = kMmaInvalid;
    second[t] = kMmaInvalid;
  }

  const uint4* tsrc = t_desc + (size_t)ts * t_stride * 2;
  for (int t0 = tb; t0 < te; t0 += kMmaTrainTile) {
    const int lim = min(kMmaTrainTile, te - t0);
    const int nrb = (lim + 31) >> 5;  // 32-row blocks that hold a train
    __syncthreads();
    // unpack the train tile (16 per set bit): item = (train j, dword kb) -> 32 bytes (both lane halves)
    for (int it = tid; it < nrb * 32 * 8; it += kThreads) {
      const int j = it >> 3, kb = it & 7;
      const uint32_t x = j < lim ? reinterpret_cast<const uint32_t*>(tsrc + (size_t)(t0 + j) * 2)[kb] : 0u;
      uint4 lo, hi;
      lo.x = (x << 4) & 0x10101010u;
      lo.y = (x << 3) & 0x10101010u;
      lo.z = (x << 2) & 0x10101010u;
      lo.w = (x << 1) & 0x10101010u;
      hi.x = x & 0x10101010u;
      hi.y = (x >> 1) & 0x10101010u;
      hi.z = (x >> 2) & 0x10101010u;
      hi.w = (x >> 3) & 0x10101010u;
      uint4* dst = reinterpret_cast<uint4*>(tile + j * kMmaRowPitch + kb * 32);
      dst[0] = lo;
      dst[1] = hi;
    }
    for (int j = tid; j < nrb * 32; j += kThreads) {
      int32_t v = kMmaInvalid7;
      if (j < lim) {
        const uint4 a = tsrc[(size_t)(t0 + j) * 2 + 0], b = tsrc[(size_t)(t0 + j) * 2 + 1];
        const int32_t pc = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) +
                           __popc(b.w);
        v = (pc << 7) + j;
      }
      base7[j] = v;
    }
    __syncthreads();
    if (!wave_live) continue;  // (uniform per wave; the barriers above are reached by every wave)
    int32_t b7[QT], s7[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) b7[t] = s7[t] = kMmaInvalid;
    for (int rb = 0; rb < nrb; ++rb) {
      v16i32 acc[QT];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const v4i32 b4 = *reinterpret_cast<const v4i32*>(base7 + rb * 32 + 8 * g + 4 * half);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          acc[t][4 * g + 0] = b4.x;
          acc[t][4 * g + 1] = b4.y;
          acc[t][4 * g + 2] = b4.z;
          acc[t][4 * g + 3] = b4.w;
        }
      }
      const uint8_t* arow = tile + (rb * 32 + col) * kMmaRowPitch + half * 16;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const v4i32 a = *reinterpret_cast<const v4i32*>(arow + kb * 32);
#pragma unroll
        for (int t = 0; t < QT; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[t][kb], acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        if (K == 1) {
#pragma unroll
          // a CHAIN min(min(best, a), b) per pair of elements: one v_min3_i32 each (no inline asm here: the compiler's
          // hazard recogniser has to see the reads of the MFMA results to place the wait states)
          for (int r = 0; r < 16; r += 2) b7[t] = min(min(b7[t], acc[t][r]), acc[t][r + 1]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) two_smallest(b7[t], s7[t], acc[t][r]);
        }
      }
    }
    // widen the tile's winner(s): ((|t| - 2 q.t) << 7) + row  ->  ((|t| - 2 q.t) << 20) + train index
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const int32_t kb1 = b7[t] < (1 << 23) ? ((b7[t] >> 7) << SOSVO_KEY_SHIFT) + (t0 + (b7[t] & 127)) : kMmaInvalid;
      if (K == 2) {
        const int32_t ks1 = s7[t] < (1 << 23) ? ((s7[t] >> 7) << SOSVO_KEY_SHIFT) + (t0 + (s7[t] & 127)) : kMmaInvalid;
        second[t] = min(max(best[t], kb1), min(second[t], ks1));
      }
      best[t] = min(best[t], kb1);
    }
  }

  // ---- the two lane halves of a column hold the same query: merge, add |q|, store
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int32_t ob = __shfl_xor(best[t], 32), os = __shfl_xor(second[t], 32);
    const int32_t mb = min(best[t], ob);
    const int32_t ms = min(min(second[t], os), max(best[t], ob));
    const int qi = q0 + (wave * QT + t) * 32 + col;
    if (half == 0 && qi < nqp) {
      uint32_t* out = keys + ((size_t)p * q_stride + qi) * K;
      const uint32_t add = (uint32_t)pq[t] << SOSVO_KEY_SHIFT;
      const uint32_t kb1 = mb >= kMmaInvalid ? SOSVO_KEY_NONE : (uint32_t)mb + add;
      if (K == 1 && nsplit > 1) {
        atomicMin(out, kb1);
      } else {
        out[0] = kb1;
        if (K == 2) out[1] = ms >= kMmaInvalid ? SOSVO_KEY_NONE : (uint32_t)ms + add;
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

  // Matrix-pipe kernel (see match_hamming_mfma_kernel): 4 waves x QT tiles of 32 queries per workgroup.
  const int qt = (q_stride >= 1024) ? 2 : 1;
  const int q_per_wg = (kThreads / 64) * 32 * qt;
  const int gx = cdiv(q_stride, q_per_wg);
  // Split the train range over grid.z until ~2k workgroups exist (1-NN only: the split
  // results merge with atomicMin on the packed key).
  int nsplit = 1;
  if (k == 1) {
    const int tiles = cdiv(t_stride, kMmaTrainTile);
    nsplit = cdiv(2048, gx * nprob);
    if (nsplit > tiles) nsplit = tiles;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 64) nsplit = 64;
  }
  if (nsplit > 1)
    SOSVO_HIP(ctx, hipMemsetAsync(keys, 0xFF, (size_t)nprob * q_stride * sizeof(uint32_t), ctx->stream));
  dim3 grid(nprob, gx, nsplit), block(kThreads);
  if (k == 1) {
    if (qt == 2)
      SOSVO_LAUNCH(ctx, (match_hamming_mfma_kernel<2, 1>), grid, block, 0, ctx->stream, q4, t4, nq, nt, q_slot, t_slot, q_stride,
                   t_stride, nsplit, keys);
    else
      SOSVO_LAUNCH(ctx, (match_hamming_mfma_kernel<1, 1>), grid, block, 0, ctx->stream, q4, t4, nq, nt, q_slot, t_slot, q_stride,
                   t_stride, nsplit, keys);
  } else {
    if (qt == 2)
      SOSVO_LAUNCH(ctx, (match_hamming_mfma_kernel<2, 2>), grid, block, 0, ctx->stream, q4, t4, nq, nt, q_slot, t_slot, q_stride,
                   t_stride, nsplit, keys);
    else
      SOSVO_LAUNCH(ctx, (match_hamming_mfma_kernel<1, 2>), grid, block, 0, ctx->stream, q4, t4, nq, nt, q_slot, t_slot, q_stride,
                   t_stride, nsplit, keys);
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
