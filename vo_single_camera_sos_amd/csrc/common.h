// Internal helpers shared by the HIP translation units of libsosvo.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "sosvo.h"

struct sosvo_ctx {
  int32_t device;
  hipStream_t stream;
  hipEvent_t ev_start, ev_stop;
  char err[512];
  // scratch workspace owned by the context (grown on demand, never inside a capture)
  void* ws;
  size_t ws_bytes;
};

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
