// Internal helpers shared by the HIP translation units of libsosvo.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "sosvo.h"

struct sosvo_prof_entry {
  const char* name;  // string literal
  hipEvent_t a, b;
};

constexpr int kSosvoProfMax = 16384;

constexpr int kSosvoMaxSubStreams = 4;

struct sosvo_ctx {
  int32_t device;
  hipStream_t stream;
  hipEvent_t ev_start, ev_stop;
  char err[512];
  // scratch workspace owned by the context (grown on demand, never inside a capture)
  void* ws;
  size_t ws_bytes;
  // optional per-kernel HIP-event profile (sosvo_profile_*): one event pair per launch
  int32_t prof_on;
  int32_t prof_n;        // entries recorded since the last enable/reset
  int32_t prof_created;  // event pairs that exist
  sosvo_prof_entry* prof;
  // sosvo_frame_pair_batch_streams: internal sub-contexts (own stream, own scratch), created on first use
  sosvo_ctx* sub[kSosvoMaxSubStreams];
  hipEvent_t sub_done[kSosvoMaxSubStreams], sub_median[kSosvoMaxSubStreams], sub_begin;
  int32_t n_sub;
  int32_t hint_score_fp64_only;  // sosvo_set_hint(SOSVO_HINT_SCORE_FP64_ONLY): no single-precision tier in ransac_score_kernel
  int32_t hint_shared_device;  // sosvo_set_hint(SOSVO_HINT_SHARED_DEVICE): other streams' kernels share the chip
  int32_t sub_last;  // parts of the most recent sosvo_frame_pair_batch_streams[_enqueue] call
  // un-joined work of an ..._enqueue call is pending on the part streams: only then does the next call chain its first
  // median behind the previous call's last one, and only then must its split (n_pairs, n_streams) equal the previous one
  int32_t sub_pending, sub_last_pairs;
};

// Brackets the kernel launches of the enclosing scope with a HIP event pair when profiling is on.
struct SosvoProfScope {
  sosvo_ctx* ctx;
  bool active;
  SosvoProfScope(sosvo_ctx* c, const char* name) : ctx(c), active(false) {
    if (!c->prof_on || !c->prof || c->prof_n >= kSosvoProfMax) return;
    sosvo_prof_entry& e = c->prof[c->prof_n];
    if (c->prof_n >= c->prof_created) {
      if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
      c->prof_created = c->prof_n + 1;
    }
    e.name = name;
    if (hipEventRecord(e.a, c->stream) != hipSuccess) return;
    active = true;
  }
  ~SosvoProfScope() {
    if (!active) return;
    (void)hipEventRecord(ctx->prof[ctx->prof_n].b, ctx->stream);
    ctx->prof_n++;
  }
};
#define SOSVO_PROFILE(ctx, name) SosvoProfScope sosvo_prof_scope_(ctx, name)

// Kernel launch on the context's stream, labelled with the kernel's name for the profile.
#define SOSVO_LAUNCH(ctx, kernel, ...)            \
  do {                                            \
    SOSVO_PROFILE(ctx, #kernel);                  \
    hipLaunchKernelGGL(kernel, __VA_ARGS__);      \
  } while (0)

#define SOSVO_WAVE 64

static inline int32_t sosvo_fail(sosvo_ctx* ctx, int32_t code, const char* what, const char* detail) {
  if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, detail ? detail : "");
  return code;
}

#define SOSVO_REQUIRE(ctx, cond, msg)                                   \
  do {                                                                  \
    if (!(cond)) return sosvo_fail((ctx), SOSVO_ERR_ARG, __func__, msg); \
  } while (0)

#define SOSVO_HIP(ctx, call)                                                          \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) return sosvo_fail((ctx), SOSVO_ERR_HIP, #call, hipGetErrorString(e_)); \
  } while (0)

// Every entry point binds the calling thread to the context's device first.
#define SOSVO_ENTER(ctx)                                   \
  do {                                                     \
    if (!(ctx)) return SOSVO_ERR_ARG;                      \
    SOSVO_HIP((ctx), hipSetDevice((ctx)->device));         \
  } while (0)

// Wave priority of the latency-bound kernels: when they share the chip with the VALU-saturating median kernel of another
// stream, their few instructions should issue ahead of its many (s_setprio: arbitration among the waves of a SIMD).
// (measured: +4 % with two streams; 3 for the latency-bound kernels, 2 for the two streaming image kernels, the median 0)
#define SOSVO_LATENCY_BOUND_PRIO() __builtin_amdgcn_s_setprio(3)
#define SOSVO_STREAMING_PRIO() __builtin_amdgcn_s_setprio(2)

#define SOSVO_LAUNCH_CHECK(ctx) SOSVO_HIP((ctx), hipGetLastError())

static inline int32_t sosvo_ws_reserve(sosvo_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return SOSVO_OK;
  if (ctx->ws) {
    SOSVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SOSVO_HIP(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  SOSVO_HIP(ctx, hipMalloc(&ctx->ws, bytes));
  ctx->ws_bytes = bytes;
  return SOSVO_OK;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Order-preserving map float -> uint32 (larger float <=> larger uint), for atomicMax and sort keys.
__host__ __device__ static inline uint32_t sosvo_float_ordered(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ static inline float sosvo_ordered_float(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

#ifdef __HIPCC__
// Stable position of a lane's element among the valid elements of a workgroup round of NW waves (256 threads unless
// said otherwise): wave ballot prefix + running base (*s_running is advanced by the round's total).  wave_off: NW + 1
// ints of LDS.
template <int NW = 4>
__device__ __forceinline__ int sosvo_block_compact_pos(bool valid, int* wave_off, int* s_running, int tid) {
  const int lane = tid & 63, wid = tid >> 6;
  const unsigned long long bal = __ballot(valid);
  __syncthreads();
  if (lane == 0) wave_off[wid + 1] = __popcll(bal);
  __syncthreads();
  if (tid == 0) {
    wave_off[0] = *s_running;
    for (int w = 0; w < NW; ++w) wave_off[w + 1] += wave_off[w];
    *s_running = wave_off[NW];
  }
  __syncthreads();
  return wave_off[wid] + __popcll(bal & ((1ULL << lane) - 1ULL));
}
#endif

// detect.hip: the rolling 7x7 blur on images that lie img_stride bytes apart (also used by orb.hip for the pyramid levels)
int32_t sosvo_launch_gauss7(sosvo_ctx* ctx, const uint8_t* in, long long img_stride, int nimg, int rows, int cols,
                            uint8_t* out);
// ... with the output images out_stride bytes apart (level 0 of the ORB pyramid is read from the dense gray batch)
int32_t sosvo_launch_gauss7_to(sosvo_ctx* ctx, const uint8_t* in, long long in_stride, int nimg, int rows, int cols,
                               uint8_t* out, long long out_stride);
// ... restricted to the output rows [row_range[2 v], row_range[2 v + 1]) (device array; v = image / imgs_per_range)
int32_t sosvo_launch_gauss7_rows(sosvo_ctx* ctx, const uint8_t* in, long long in_stride, int nimg, int rows, int cols,
                                 uint8_t* out, long long out_stride, const int32_t* row_range, int imgs_per_range);
