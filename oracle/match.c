/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product
 * path (vo_single_camera_sos_amd/, libsosvo.so); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * CPU restatement of the brute-force Hamming matching stage.
 *
 * The arithmetic lives in OpenCV (cv2.BFMatcher), a third-party dependency that is NOT
 * under the reference tree and is not pinned by it ("OpenCV 3", reference README.md:119-190).
 * The restatement follows the published semantics of BFMatcher(NORM_HAMMING) and is anchored
 * on the reference's call sites:
 *   omnistereo/camera_models.py:402   BFMatcher(normType=NORM_HAMMING), crossCheck off
 *   omnistereo/camera_models.py:442   matcher.match(query, train)       -> 1-NN
 *   omnistereo/camera_models.py:420   matcher.knnMatch(query, train, k) -> k-NN
 *   omnistereo/camera_models.py:413   matcher.radiusMatch(query, train, maxDistance) -> all within the radius
 *   omnistereo/camera_models.py:444   sorted(matches, key=distance)     -> stable sort
 * Parity status: UNPINNED against OpenCV binaries (the reference ships no tests or golden
 * vectors for this path); pinned by hand-computable known-answer
 * tests in tests/test_oracle_match.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KEY_SHIFT 20
#define KEY_NONE 0xFFFFFFFFu

static int hamming32(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int w = 0; w < 4; ++w) {
    uint64_t x, y;
    memcpy(&x, a + 8 * w, 8);
    memcpy(&y, b + 8 * w, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

/* One matching problem.  For each query row scan the train rows in index order and keep the
 * k smallest distances; a later row replaces an earlier one only when STRICTLY smaller, so
 * the lowest train index wins ties (batchDistance + "d < best" in BFMatcher).
 * keys[i*k + r] = (distance << 20) | train index of the r-th neighbour, KEY_NONE if absent. */
void orc_match_hamming(const uint8_t* q, const uint8_t* t, int32_t nq, int32_t nt, int32_t k,
                       uint32_t* keys) {
  for (int i = 0; i < nq; ++i) {
    int bd[2] = {1 << 30, 1 << 30};
    int bi[2] = {-1, -1};
    for (int j = 0; j < nt; ++j) {
      int d = hamming32(q + 32 * (size_t)i, t + 32 * (size_t)j);
      if (d < bd[0]) {
        bd[1] = bd[0];
        bi[1] = bi[0];
        bd[0] = d;
        bi[0] = j;
      } else if (k > 1 && d < bd[1]) {
        bd[1] = d;
        bi[1] = j;
      }
    }
    for (int r = 0; r < k; ++r)
      keys[(size_t)i * k + r] = bi[r] < 0 ? KEY_NONE : (((uint32_t)bd[r]) << KEY_SHIFT) | (uint32_t)bi[r];
  }
}

/* Stable merge sort of match indices by distance only (Python's sorted(key=distance),
 * camera_models.py:444): equal distances keep query order.  order[r] = query index. */
static void merge_sort(int32_t* idx, int32_t* tmp, const uint32_t* dist, int lo, int hi) {
  if (hi - lo < 2) return;
  int mid = lo + (hi - lo) / 2;
  merge_sort(idx, tmp, dist, lo, mid);
  merge_sort(idx, tmp, dist, mid, hi);
  int a = lo, b = mid, o = lo;
  while (a < mid && b < hi) {
    if (dist[idx[b]] < dist[idx[a]])
      tmp[o++] = idx[b++];
    else
      tmp[o++] = idx[a++];
  }
  while (a < mid) tmp[o++] = idx[a++];
  while (b < hi) tmp[o++] = idx[b++];
  memcpy(idx + lo, tmp + lo, (size_t)(hi - lo) * sizeof(int32_t));
}

void orc_sort_matches(const uint32_t* keys, int32_t nq, int32_t* order) {
  if (nq <= 0) return;
  uint32_t* dist = (uint32_t*)malloc((size_t)nq * sizeof(uint32_t));
  int32_t* tmp = (int32_t*)malloc((size_t)nq * sizeof(int32_t));
  for (int i = 0; i < nq; ++i) {
    dist[i] = keys[i] >> KEY_SHIFT; /* KEY_NONE -> 0xFFF: sorts last */
    order[i] = i;
  }
  merge_sort(order, tmp, dist, 0, nq);
  free(dist);
  free(tmp);
}

/* radiusMatch (camera_models.py:413): for each query row ALL train rows with distance <= max_distance
 * ("not farther than", inclusive), as keys (distance << 20 | train index) in ascending key order = ascending
 * distance, lowest train index first among equals (OpenCV's per-query std::sort leaves the order of equal
 * distances unspecified; this is the deterministic choice).  At most cap keys per query are written
 * (keys[i*cap + r]), counts[i] = number found (may exceed cap: the smallest cap keys are the ones kept). */
static int cmp_u32(const void* a, const void* b) {
  const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
void orc_match_radius(const uint8_t* q, const uint8_t* t, int32_t nq, int32_t nt, int32_t max_distance,
                      int32_t cap, uint32_t* keys, int32_t* counts) {
  uint32_t* all = (uint32_t*)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(uint32_t));
  for (int i = 0; i < nq; ++i) {
    int m = 0;
    for (int j = 0; j < nt; ++j) {
      int d = hamming32(q + 32 * (size_t)i, t + 32 * (size_t)j);
      if (d <= max_distance) all[m++] = (((uint32_t)d) << KEY_SHIFT) | (uint32_t)j;
    }
    qsort(all, (size_t)m, sizeof(uint32_t), cmp_u32);
    counts[i] = m;
    for (int r = 0; r < cap; ++r) keys[(size_t)i * cap + r] = r < m ? all[r] : KEY_NONE;
  }
  free(all);
}

/* Float descriptors (SIFT / SURF / KAZE ...): cv2.BFMatcher() with its default NORM_L2 (camera_models.py:396) --
 * distance = sqrt of the sum of squared differences accumulated in float32, four terms per step in index order
 * (OpenCV's scalar normL2Sqr_: s += v0*v0 + v1*v1 + v2*v2 + v3*v3, then the tail one by one), k smallest per query, a
 * later train row replaces an earlier one only when strictly smaller.  keys[i*k + r] = (bits of the float32 distance
 * << 32) | train index (non-negative floats order like their bit patterns), all ones if absent. */
#include <math.h>
static float l2_distance(const float* a, const float* b, int dim) {
  float s = 0.0f;
  int i = 0;
  for (; i <= dim - 4; i += 4) {
    const float v0 = a[i] - b[i], v1 = a[i + 1] - b[i + 1], v2 = a[i + 2] - b[i + 2], v3 = a[i + 3] - b[i + 3];
    s += (((v0 * v0) + (v1 * v1)) + (v2 * v2)) + (v3 * v3);
  }
  for (; i < dim; ++i) {
    const float v = a[i] - b[i];
    s += v * v;
  }
  return sqrtf(s);
}

void orc_match_l2(const float* q, const float* t, int32_t nq, int32_t nt, int32_t dim, int32_t k, uint64_t* keys) {
  for (int i = 0; i < nq; ++i) {
    float bd[2] = {INFINITY, INFINITY};
    int bi[2] = {-1, -1};
    for (int j = 0; j < nt; ++j) {
      const float d = l2_distance(q + (size_t)dim * i, t + (size_t)dim * j, dim);
      if (d < bd[0] || bi[0] < 0) {
        bd[1] = bd[0];
        bi[1] = bi[0];
        bd[0] = d;
        bi[0] = j;
      } else if (k > 1 && (d < bd[1] || bi[1] < 0)) {
        bd[1] = d;
        bi[1] = j;
      }
    }
    for (int r = 0; r < k; ++r) {
      uint32_t bits;
      memcpy(&bits, &bd[r], 4);
      keys[(size_t)i * k + r] = bi[r] < 0 ? ~(uint64_t)0 : ((uint64_t)bits << 32) | (uint32_t)bi[r];
    }
  }
}
