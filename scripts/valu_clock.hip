// Micro-benchmark: the VALU issue cost in REAL shader cycles.
//
// scripts/valu_rate.hip converts wall time to "cycles" by assuming 2.4 GHz.  This one brackets the measured loop
// with s_memtime (shader-clock ticks, MI355X_MICROARCH.md "s_memtime tick = shader cycle") AND s_memrealtime (the
// constant 100 MHz reference counter) inside the kernel, and with HIP events outside, so that
//   * the shader clock under this load  = memtime ticks / (realtime ticks / 100 MHz)
//   * the issue cost in real cycles      = memtime ticks of a wave / (instructions the SIMD issued meanwhile)
// are measured facts.  The number of waves per SIMD is CONTROLLED: one workgroup of 256 * n threads per CU (its 4 n
// waves are dealt round-robin over the CU's 4 SIMDs) that asks for 100 KB of LDS, so that no second workgroup fits
// beside it (n = 8: two workgroups of 1024 threads and 70 KB each).  A first version launched 256 * n independent
// 256-thread workgroups and let the dispatcher place them: it does NOT spread them evenly (some CUs took 6 of them,
// others 2 -- the waves' own brackets were a third shorter than the launch), which is also what inflated the
// wall-clock figures of scripts/valu_rate.hip by a few per cent.  With the placement pinned, a SIMD issues n x
// (instructions of one wave) during its LONGEST wave bracket (the arbiter serves the oldest wave first, so the waves
// of a SIMD do not finish together).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_clock valu_clock.hip ; run: ./valu_clock [> profiles/roundN/valu_clock.txt]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

struct Stamp {
  unsigned long long cyc, real;
};

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, Stamp* stamps, int iters, uint32_t seed) {
  extern __shared__ uint32_t lds_hog[];
  if (seed == 0xFFFFFFFFu) lds_hog[threadIdx.x] = seed;  // (never: keeps the allocation)
  uint32_t a[8], b = seed ^ threadIdx.x, c = seed * 3u + 1u;
  for (int i = 0; i < 8; ++i) a[i] = seed + i * 977u + threadIdx.x;
  unsigned long long bal = 0, bal2 = 0x123456789abcdefULL + seed;
  double d[8] = {1, 2, 3, 4, 5, 6, 7, 8}, dd = 1.0000001;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#define STEP(i)                                                                                                \
  if (OP == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                     \
  if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));                                \
  if (OP == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(a[i]) : "v"(b), "v"(c));          \
  if (OP == 3) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]) :);                                         \
  if (OP == 4) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]) :);                                         \
  if (OP == 5) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                     \
  if (OP == 6) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                     \
  if (OP == 7) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dd));                                \
  if (OP == 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dd));                                    \
  if (OP == 9) asm volatile("v_cmp_lt_i16_sdwa %0, sext(%1), %2 src0_sel:BYTE_0 src1_sel:DWORD" : "=s"(bal) : "v"(a[i]), "v"(b)); \
  if (OP == 10) asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(bal) : "v"(a[i] & 63), "s"(bal2));              \
  if (OP == 11) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b)); \
  if (OP == 12) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                 \
  if (OP == 13) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));                                \
  if (OP == 14) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                    \
  if (OP == 15) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));                            \
  if (OP == 16) /* the median's select: and + bcnt + bitop3 per window word */                                 \
    asm volatile("v_and_b32 %0, %1, %2\n v_bcnt_u32_b32 %3, %0, %3\n v_bitop3_b32 %1, %1, %2, %4 bitop3:0x90"  \
                 : "=&v"(a[i]), "+v"(a[(i + 1) & 7]), "+v"(b), "+v"(c)                                         \
                 : "v"(a[(i + 2) & 7]));                                                                       \
  if (OP == 17) asm volatile("v_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                    \
  if (OP == 18) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                    \
  if (OP == 19) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dd));
      REP8(STEP)
    }
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
  uint32_t r = (uint32_t)bal;
  for (int i = 0; i < 8; ++i) r ^= (uint32_t)d[i];
  for (int i = 0; i < 8; ++i) r ^= a[i];
  if (r == 0xDEADBEEF) out[threadIdx.x] = r + c;
  if ((threadIdx.x & 63) == 0) {
    Stamp s;
    s.cyc = t1 - t0;
    s.real = r1 - r0;
    stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = s;
  }
}

template <int OP>
void run(const char* name, int per_step, uint32_t* out, Stamp* stamps) {
  const int iters = 20000;
  for (int wps : {1, 2, 3, 4, 8}) {
    const int blocks = wps == 8 ? 512 : 256, threads = wps == 8 ? 1024 : 256 * wps;
    const size_t lds = wps == 8 ? 70 * 1024 : 100 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, out, stamps, 200, 7u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, out, stamps, iters, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(blocks * 16);
    hipMemcpy(h.data(), stamps, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0, real_max = 0, cyc_max = 0;
    int nw = 0;
    for (int bl = 0; bl < blocks; ++bl)
      for (int w = 0; w < threads / 64; ++w, ++nw) {
        const Stamp& s = h[bl * 16 + w];
        cyc += (double)s.cyc;
        real += (double)s.real;
        real_max = std::max(real_max, (double)s.real);
        cyc_max = std::max(cyc_max, (double)s.cyc);
      }
    cyc /= nw;
    real /= nw;
    const double instr_per_simd = (double)iters * 32 * per_step * wps;
    const double ghz = cyc / (real * 10.0);  // realtime tick = 10 ns
    // The SIMD's arbiter is not fair (oldest wave first): waves of one SIMD finish at different times, so the SIMD's
    // issue cost is (longest bracket) / (instructions of all its waves); the mean bracket only says how unfair it was.
    printf("%-22s waves/SIMD %d: wall %.3f ms, wave brackets mean %.3f max %.3f ms, shader clock %.3f GHz, "
           "%.2f real cycles per wave-instr per SIMD\n",
           name, wps, ms, real * 1e-5, real_max * 1e-5, ghz, cyc_max / instr_per_simd);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
}

int main() {
  uint32_t* out;
  Stamp* stamps;
  hipMalloc(&out, 4096);
  hipMalloc(&stamps, sizeof(Stamp) * 512 * 16);
  run<0>("v_and_b32", 1, out, stamps);
  run<18>("v_xor_b32", 1, out, stamps);
  run<14>("v_add_u32", 1, out, stamps);
  run<2>("v_bitop3_b32", 1, out, stamps);
  run<4>("v_lshrrev_b32", 1, out, stamps);
  run<5>("v_min_u16", 1, out, stamps);
  run<17>("v_max_u16", 1, out, stamps);
  run<13>("v_fma_f32", 1, out, stamps);
  run<1>("v_bcnt_u32_b32", 1, out, stamps);
  run<3>("v_lshlrev_b32", 1, out, stamps);
  run<6>("v_min_u32", 1, out, stamps);
  run<12>("v_pk_min_u16", 1, out, stamps);
  run<15>("v_mad_u32_u24", 1, out, stamps);
  run<9>("v_cmp_lt_i16_sdwa", 1, out, stamps);
  run<10>("v_lshlrev_b64", 1, out, stamps);
  run<11>("v_mov_dpp wave_shr", 1, out, stamps);
  run<7>("v_fma_f64", 1, out, stamps);
  run<8>("v_add_f64", 1, out, stamps);
  run<19>("v_mul_f64", 1, out, stamps);
  run<16>("and+bcnt+bitop3 (x3)", 3, out, stamps);
  return 0;
}
