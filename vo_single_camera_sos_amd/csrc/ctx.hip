// Context, error reporting and the HIP-event timer of libsosvo.so.
#include "common.h"
#include "orb_bit_pattern_31.h"

#include <stdlib.h>

extern "C" {

int32_t sosvo_abi_version(void) { return 1; }

int32_t sosvo_orb_bit_pattern_31(int8_t* pattern_host) {
  if (!pattern_host) return SOSVO_ERR_ARG;
  memcpy(pattern_host, sosvo_orb_bit_pattern_31_table, 1024);
  return SOSVO_OK;
}

int32_t sosvo_create(sosvo_ctx** out, int32_t device, void* stream) {
  if (!out) return SOSVO_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SOSVO_ERR_NODEVICE;
  if (device < 0 || device >= n) return SOSVO_ERR_NODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SOSVO_ERR_HIP;
  // The code object only holds gfx950 ISA; refuse anything else up front instead of
  // failing at the first launch.
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SOSVO_ERR_NODEVICE;
  sosvo_ctx* ctx = (sosvo_ctx*)calloc(1, sizeof(sosvo_ctx));
  if (!ctx) return SOSVO_ERR_HIP;
  ctx->device = device;
  ctx->stream = (hipStream_t)stream;
  if (hipSetDevice(device) != hipSuccess || hipEventCreate(&ctx->ev_start) != hipSuccess ||
      hipEventCreate(&ctx->ev_stop) != hipSuccess) {
    free(ctx);
    return SOSVO_ERR_HIP;
  }
  *out = ctx;
  return SOSVO_OK;
}

int32_t sosvo_set_hint(sosvo_ctx* ctx, int32_t hint, int32_t value) {
  if (!ctx) return SOSVO_ERR_ARG;
  if (hint == SOSVO_HINT_SHARED_DEVICE) {
    ctx->hint_shared_device = value != 0;
  } else if (hint == SOSVO_HINT_SCORE_FP64_ONLY) {
    ctx->hint_score_fp64_only = value != 0;
    for (int i = 0; i < ctx->n_sub; ++i)  // the internal contexts of sosvo_frame_pair_batch_streams follow
      if (ctx->sub[i]) ctx->sub[i]->hint_score_fp64_only = value != 0;
  } else {
    return sosvo_fail(ctx, SOSVO_ERR_ARG, __func__, "unknown hint");
  }
  return SOSVO_OK;
}

int32_t sosvo_destroy(sosvo_ctx* ctx) {
  if (!ctx) return SOSVO_OK;
  (void)hipSetDevice(ctx->device);
  for (int i = 0; i < ctx->n_sub; ++i) {  // internal streams of sosvo_frame_pair_batch_streams
    if (ctx->sub[i]) {
      hipStream_t st = ctx->sub[i]->stream;
      (void)hipStreamSynchronize(st);
      (void)sosvo_destroy(ctx->sub[i]);
      (void)hipStreamDestroy(st);
    }
    (void)hipEventDestroy(ctx->sub_done[i]);
    (void)hipEventDestroy(ctx->sub_median[i]);
  }
  if (ctx->n_sub) (void)hipEventDestroy(ctx->sub_begin);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->prof) {
    for (int i = 0; i < ctx->prof_created; ++i) {
      (void)hipEventDestroy(ctx->prof[i].a);
      (void)hipEventDestroy(ctx->prof[i].b);
    }
    free(ctx->prof);
  }
  (void)hipEventDestroy(ctx->ev_start);
  (void)hipEventDestroy(ctx->ev_stop);
  free(ctx);
  return SOSVO_OK;
}

int32_t sosvo_set_stream(sosvo_ctx* ctx, void* stream) {
  SOSVO_ENTER(ctx);
  ctx->stream = (hipStream_t)stream;
  return SOSVO_OK;
}

int32_t sosvo_synchronize(sosvo_ctx* ctx) {
  SOSVO_ENTER(ctx);
  SOSVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SOSVO_OK;
}

const char* sosvo_last_error(const sosvo_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int32_t sosvo_timer_start(sosvo_ctx* ctx) {
  SOSVO_ENTER(ctx);
  SOSVO_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
  return SOSVO_OK;
}

int32_t sosvo_timer_stop(sosvo_ctx* ctx) {
  SOSVO_ENTER(ctx);
  SOSVO_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
  return SOSVO_OK;
}

int32_t sosvo_profile_enable(sosvo_ctx* ctx, int32_t on) {
  SOSVO_ENTER(ctx);
  if (on && !ctx->prof) {
    ctx->prof = (sosvo_prof_entry*)calloc(kSosvoProfMax, sizeof(sosvo_prof_entry));
    if (!ctx->prof) return sosvo_fail(ctx, SOSVO_ERR_HIP, __func__, "out of host memory");
  }
  ctx->prof_on = on ? 1 : 0;
  ctx->prof_n = 0;
  return SOSVO_OK;
}

int32_t sosvo_profile_count(sosvo_ctx* ctx) { return ctx ? ctx->prof_n : 0; }

int32_t sosvo_profile_get(sosvo_ctx* ctx, int32_t i, char* name_out, int32_t name_cap, float* ms) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, ctx->prof && i >= 0 && i < ctx->prof_n && name_out && name_cap > 0 && ms, "bad arguments");
  const sosvo_prof_entry& e = ctx->prof[i];
  SOSVO_HIP(ctx, hipEventSynchronize(e.b));
  SOSVO_HIP(ctx, hipEventElapsedTime(ms, e.a, e.b));
  snprintf(name_out, (size_t)name_cap, "%s", e.name ? e.name : "");
  return SOSVO_OK;
}

int32_t sosvo_debug_fill_scratch(sosvo_ctx* ctx, int32_t byte) {
  SOSVO_ENTER(ctx);
  if (ctx->ws && ctx->ws_bytes) SOSVO_HIP(ctx, hipMemsetAsync(ctx->ws, byte & 0xFF, ctx->ws_bytes, ctx->stream));
  for (int i = 0; i < ctx->n_sub; ++i) {
    if (!ctx->sub[i]) continue;
    const int32_t rc = sosvo_debug_fill_scratch(ctx->sub[i], byte);
    if (rc != SOSVO_OK) return sosvo_fail(ctx, rc, __func__, ctx->sub[i]->err);
  }
  return SOSVO_OK;
}

int32_t sosvo_timer_elapsed_ms(sosvo_ctx* ctx, float* ms) {
  SOSVO_ENTER(ctx);
  SOSVO_REQUIRE(ctx, ms != nullptr, "ms is null");
  SOSVO_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
  SOSVO_HIP(ctx, hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
  return SOSVO_OK;
}

}  // extern "C"
