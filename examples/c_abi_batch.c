/* The C ABI of libsosvo.so used from plain C (no Python, no torch): the whole hot path for B frame pairs through
 * sosvo_unwrap_prepare + sosvo_frame_pair_batch on buffers from hipMalloc.
 *
 *   c_abi_batch <input.bin> <output.bin> [internal streams 1..4 | graph | enqueue | sequence]
 *
 *   (no mode)  one sosvo_frame_pair_batch call            "2".."4"   the batch split over internal streams of the library
 *   graph      the call captured into a HIP graph         enqueue    four back-to-back ..._streams_enqueue calls, one join
 *   sequence   the same frames as ONE sequence: sosvo_sequence_front_end in windows of three frames into a frame store,
 *              then sosvo_sequence_track on the slot pairs (2 i, 2 i + 1): the records of the pair batch, bit for bit
 *
 * input.bin (written by tests/test_gpu_c_abi_example.py): struct sosvo_batch_cfg, struct sosvo_rig, then the raw
 * arrays omni [2B,H,W,3] u8, annulus masks [2,H,W] u8, map_x, map_y [2,rows,cols] f32, mask_bits [2,rows,cols] u32,
 * pattern [512,2] i8.  output.bin: results [B,16] f64.
 * Build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_batch.c -o examples/c_abi_batch \
 *            -Lvo_single_camera_sos_amd -l:libsosvo.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../vo_single_camera_sos_amd' */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sosvo.h"

#define CHECK_HIP(x)                                                              \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 2;                                                                   \
    }                                                                             \
  } while (0)
#define CHECK_SOSVO(x)                                                            \
  do {                                                                            \
    int32_t rc_ = (x);                                                            \
    if (rc_ != SOSVO_OK) {                                                        \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, sosvo_last_error(ctx));          \
      return 3;                                                                   \
    }                                                                             \
  } while (0)

static void* upload(FILE* f, size_t bytes) {
  void* host = malloc(bytes);
  void* dev = NULL;
  if (!host || fread(host, 1, bytes, f) != bytes) {
    fprintf(stderr, "short input file\n");
    exit(4);
  }
  if (hipMalloc(&dev, bytes) != hipSuccess || hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
    fprintf(stderr, "device upload failed\n");
    exit(5);
  }
  free(host);
  return dev;
}

int main(int argc, char** argv) {
  /* the library's internal streams + this program's own: more hardware queues than the runtime's default, so that streams
     do not share (and serialise on) one -- must happen before the first HIP call */
  setenv("GPU_MAX_HW_QUEUES", "32", 0);
  if (argc != 3 && argc != 4) {
    fprintf(stderr, "usage: %s input.bin output.bin [internal streams 1..4 | graph | enqueue | sequence]\n", argv[0]);
    return 1;
  }
  const int use_graph = argc == 4 && strcmp(argv[3], "graph") == 0;
  const int use_enqueue = argc == 4 && strcmp(argv[3], "enqueue") == 0;
  const int use_sequence = argc == 4 && strcmp(argv[3], "sequence") == 0;
  const int n_streams = use_enqueue ? 2 : (argc == 4 && !use_graph && !use_sequence ? atoi(argv[3]) : 1);
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  sosvo_batch_cfg cfg;
  sosvo_rig rig;
  if (fread(&cfg, sizeof(cfg), 1, f) != 1 || fread(&rig, sizeof(rig), 1, f) != 1) return 4;
  const size_t B = (size_t)cfg.n_pairs, HW = (size_t)cfg.H * cfg.W, P = (size_t)cfg.rows * cfg.cols;
  uint8_t* omni = (uint8_t*)upload(f, 2 * B * HW * 3);
  uint8_t* masks = (uint8_t*)upload(f, 2 * HW);
  float* map_x = (float*)upload(f, 2 * P * sizeof(float));
  float* map_y = (float*)upload(f, 2 * P * sizeof(float));
  uint32_t* mask_bits = (uint32_t*)upload(f, 2 * P * sizeof(uint32_t));
  int8_t* pattern = (int8_t*)upload(f, 512 * 2);
  fclose(f);

  sosvo_ctx* ctx = NULL;
  hipStream_t stream = NULL;
  if (use_graph) CHECK_HIP(hipStreamCreate(&stream)); /* a capture needs a stream of its own */
  if (sosvo_create(&ctx, 0, stream) != SOSVO_OK) {  /* device 0, the default stream unless capturing */
    fprintf(stderr, "sosvo_create failed\n");
    return 3;
  }
  uint32_t* table = NULL;
  void* workspace = NULL;
  double* results = NULL;
  CHECK_HIP(hipMalloc((void**)&table, 2 * P * 2 * sizeof(uint32_t)));
  CHECK_SOSVO(sosvo_unwrap_prepare(ctx, masks, map_x, map_y, cfg.H, cfg.W, cfg.rows, cfg.cols, table));  /* once per model */
  /* one stream, or the batch split over the library's internal streams (same results, more overlap) */
  enum { SEQ_WINDOW = 3 };
  const int seq_slots = (int)(2 * B) > SEQ_WINDOW ? (int)(2 * B) : SEQ_WINDOW;
  const size_t ws_bytes = use_sequence ? sosvo_sequence_workspace(&cfg, SEQ_WINDOW, seq_slots)
                          : n_streams > 1 ? sosvo_frame_pair_batch_streams_workspace(&cfg, n_streams)
                                          : sosvo_frame_pair_batch_workspace(&cfg);
  if (ws_bytes == 0) {
    fprintf(stderr, "bad configuration\n");
    return 3;
  }
  CHECK_HIP(hipMalloc(&workspace, ws_bytes));
  CHECK_HIP(hipMalloc((void**)&results, B * 16 * sizeof(double)));
  if (use_sequence) {
    /* every frame's front end ONCE, three frames per call, into slots 0 .. 2B - 1 of the frame store; then the B
     * (reference, current) slot pairs in one tracking call (pair i samples with seed + i, as in the pair batch) */
    for (int f0 = 0; f0 < (int)(2 * B); f0 += SEQ_WINDOW) {
      const int n = (int)(2 * B) - f0 < SEQ_WINDOW ? (int)(2 * B) - f0 : SEQ_WINDOW;
      CHECK_SOSVO(sosvo_sequence_front_end(ctx, &rig, &cfg, SEQ_WINDOW, seq_slots, omni + (size_t)f0 * HW * 3, n, f0, table,
                                           mask_bits, pattern, workspace, ws_bytes));
    }
    int32_t* ref = (int32_t*)malloc(B * sizeof(int32_t));
    int32_t* cur = (int32_t*)malloc(B * sizeof(int32_t));
    int32_t* counts = (int32_t*)malloc(2 * B * sizeof(int32_t));
    for (size_t i = 0; i < B; ++i) {
      ref[i] = (int32_t)(2 * i);
      cur[i] = (int32_t)(2 * i + 1);
    }
    CHECK_SOSVO(sosvo_sequence_frame_counts(ctx, &cfg, SEQ_WINDOW, seq_slots, 0, (int32_t)(2 * B), workspace, ws_bytes, counts));
    printf("sequence: %d frames, valid stereo correspondences of frame 0: %d\n", (int)(2 * B), (int)counts[0]);
    CHECK_SOSVO(sosvo_sequence_track(ctx, &rig, &cfg, SEQ_WINDOW, seq_slots, ref, cur, (int32_t)B, cfg.seed, workspace, ws_bytes,
                                     results));
    free(ref);
    free(cur);
    free(counts);
  } else if (use_enqueue) {
    /* four calls enqueued back to back on the library's internal streams (they overlap; two alternating record buffers),
     * ONE join before the records are read */
    double* results2 = NULL;
    CHECK_HIP(hipMalloc((void**)&results2, B * 16 * sizeof(double)));
    for (int k = 0; k < 4; ++k)
      CHECK_SOSVO(sosvo_frame_pair_batch_streams_enqueue(ctx, &rig, &cfg, n_streams, omni, table, mask_bits, pattern, workspace,
                                                         ws_bytes, (k & 1) ? results2 : results));
    CHECK_SOSVO(sosvo_frame_pair_batch_streams_join(ctx));
    CHECK_SOSVO(sosvo_synchronize(ctx));
    double* a = (double*)malloc(B * 16 * sizeof(double));
    double* b = (double*)malloc(B * 16 * sizeof(double));
    CHECK_HIP(hipMemcpy(a, results, B * 16 * sizeof(double), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(b, results2, B * 16 * sizeof(double), hipMemcpyDeviceToHost));
    const int same = memcmp(a, b, B * 16 * sizeof(double)) == 0;
    printf("enqueue: the two record buffers %s\n", same ? "agree" : "DIFFER");
    free(a);
    free(b);
    CHECK_HIP(hipFree(results2));
    if (!same) return 7;
  } else if (use_graph) {
    /* "graph": the call only enqueues work, so it can be captured into a HIP graph and replayed.  One eager call first
     * (the library sizes its scratch memory on first use), then capture, then REPLAYS replays, each checked against the
     * eager records bit for bit, with the host time per replay. */
    enum { REPLAYS = 20 };
    CHECK_SOSVO(sosvo_frame_pair_batch(ctx, &rig, &cfg, omni, table, mask_bits, pattern, workspace, ws_bytes, results));
    CHECK_SOSVO(sosvo_synchronize(ctx));
    double* eager = (double*)malloc(B * 16 * sizeof(double));
    double* again = (double*)malloc(B * 16 * sizeof(double));
    CHECK_HIP(hipMemcpy(eager, results, B * 16 * sizeof(double), hipMemcpyDeviceToHost));
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
    CHECK_SOSVO(sosvo_frame_pair_batch(ctx, &rig, &cfg, omni, table, mask_bits, pattern, workspace, ws_bytes, results));
    CHECK_HIP(hipStreamEndCapture(stream, &graph));
    CHECK_HIP(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
    int mismatches = 0;
    struct timespec t0, t1;
    double total_ms = 0.0;
    for (int r = 0; r < REPLAYS; ++r) {
      CHECK_HIP(hipMemsetAsync(results, 0xFF, B * 16 * sizeof(double), stream));
      CHECK_HIP(hipStreamSynchronize(stream));
      clock_gettime(CLOCK_MONOTONIC, &t0);
      CHECK_HIP(hipGraphLaunch(exec, stream));
      CHECK_HIP(hipStreamSynchronize(stream));
      clock_gettime(CLOCK_MONOTONIC, &t1);
      total_ms += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
      CHECK_HIP(hipMemcpy(again, results, B * 16 * sizeof(double), hipMemcpyDeviceToHost));
      if (memcmp(again, eager, B * 16 * sizeof(double)) != 0) ++mismatches;
    }
    printf("graph: %d of %d replays differ from the eager records; %.3f ms per replay\n", mismatches, REPLAYS, total_ms / REPLAYS);
    free(eager);
    free(again);
    CHECK_HIP(hipGraphExecDestroy(exec));
    CHECK_HIP(hipGraphDestroy(graph));
    if (mismatches) return 7;
  } else if (n_streams > 1)
    CHECK_SOSVO(sosvo_frame_pair_batch_streams(ctx, &rig, &cfg, n_streams, omni, table, mask_bits, pattern, workspace, ws_bytes,
                                               results));
  else
    CHECK_SOSVO(sosvo_frame_pair_batch(ctx, &rig, &cfg, omni, table, mask_bits, pattern, workspace, ws_bytes, results));
  CHECK_SOSVO(sosvo_synchronize(ctx));
  double* host = (double*)malloc(B * 16 * sizeof(double));
  CHECK_HIP(hipMemcpy(host, results, B * 16 * sizeof(double), hipMemcpyDeviceToHost));
  FILE* o = fopen(argv[2], "wb");
  if (!o || fwrite(host, sizeof(double), B * 16, o) != B * 16) return 6;
  fclose(o);
  for (size_t i = 0; i < B; ++i)
    printf("pair %zu: %d inliers of %d correspondences, status %d\n", i, (int)host[16 * i + 12], (int)host[16 * i + 13],
           (int)host[16 * i + 14]);
  sosvo_destroy(ctx);
  return 0;
}
