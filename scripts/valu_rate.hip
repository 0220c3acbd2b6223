// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the integer VALU ops the median /
// matching kernels are made of, at 1, 2 and 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[8], b = seed ^ threadIdx.x, c = seed * 3u + 1u;
  for (int i = 0; i < 8; ++i) a[i] = seed + i * 977u + threadIdx.x;
  unsigned long long bal = 0;
  asm volatile("v_cmp_gt_u32 vcc, %0, %1" ::"v"(b), "v"(c) : "vcc");
  unsigned long long bal2 = 0x123456789abcdefULL + seed;
  double d[8] = {1, 2, 3, 4, 5, 6, 7, 8}, dd = 1.0000001;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#define STEP(i)                                                                                              \
  if (OP == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                   \
  if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));                              \
  if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 11" : "+v"(a[i]) : "v"(b));                          \
  if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );                      \
  if (OP == 19) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(bal2));           \
  if (OP == 20) a[i] = (b > a[i]) ? (a[i] ^ c) : a[i] + 1;                                                    \
  if (OP == 21) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));                      \
  if (OP == 22) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                      \
  if (OP == 23) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                      \
  if (OP == 24) asm volatile("v_lshrrev_b32 %0, 11, %0" : "+v"(a[i]) : );                                    \
  if (OP == 25) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 26) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
  if (OP == 27) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 28) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                     \
  if (OP == 29) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 30) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));                              \
  if (OP == 31) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));                               \
  if (OP == 32) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(a[i]) : "v"(b), "v"(c));        \
  if (OP == 33) asm volatile("v_xnor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                 \
  if (OP == 34) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                   \
  if (OP == 35) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]) : );                                            \
  if (OP == 36) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 37) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 38) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                     \
  if (OP == 39) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                     \
  if (OP == 40) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                     \
  if (OP == 41) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
  if (OP == 42) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
  if (OP == 43) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
  if (OP == 44) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                     \
  if (OP == 45) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]) : );                                     \
  if (OP == 46) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");                      \
  if (OP == 47) asm volatile("v_cmp_lt_i16_sdwa %0, sext(%1), %2 src0_sel:BYTE_0 src1_sel:DWORD" : "=s"(bal) : "v"(a[i]), "v"(b)); \
  if (OP == 48) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");           \
  if (OP == 49) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");                 \
  if (OP == 50) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]) : );                                     \
  if (OP == 51) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[i]) : );                                    \
  if (OP == 52) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 53) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 54) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                     \
  if (OP == 55) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));                                      \
  if (OP == 56) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b)); \
  if (OP == 57) asm volatile("v_and_b32 %0, 0x7ff07ff, %0" : "+v"(a[i]) : );                                 \
  if (OP == 58) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                     \
  if (OP == 59) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                  \
  if (OP == 60) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dd));                          \
  if (OP == 61) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 62) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
  if (OP == 63) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));                         \
  if (OP == 64) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                         \
  if (OP == 65) asm volatile("v_and_b32 %0, %0, %1\n v_bcnt_u32_b32 %0, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
  if (OP == 66) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));                         \
  if (OP == 67) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 68) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
  if (OP == 69) asm volatile("v_pk_lshrrev_b16 %0, 15, %0" : "+v"(a[i]) : );                                 \
  if (OP == 70) asm volatile("v_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 71) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                         \
  if (OP == 72) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 73) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");                      \
  if (OP == 74) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(bal) : "v"(a[i]), "v"(b));                     \
  if (OP == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                   \
  if (OP == 5) asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(bal) : "v"(a[i] & 63), "s"(bal2));             \
  if (OP == 6) asm volatile("v_cmp_ne_u32 %0, %1, %2" : "=s"(bal) : "v"(a[i]), "v"(b));                      \
  if (OP == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                   \
  if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));                               \
  if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                \
  if (OP == 10) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));                          \
  if (OP == 11) asm volatile("v_bfe_u32 %0, %0, %1, 11" : "+v"(a[i]) : "v"(b));                              \
  if (OP == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
  if (OP == 13) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dd));                                 \
  if (OP == 14) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dd));                             \
  if (OP == 15) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dd));                                 \
  if (OP == 16) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b)); \
  if (OP == 17) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));                           \
  if (OP == 18) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      REP8(STEP)
    }
  }
  uint32_t r = (uint32_t)bal;
  for (int i = 0; i < 8; ++i) r ^= (uint32_t)d[i];
  for (int i = 0; i < 8; ++i) r ^= a[i];
  if (r == 0xDEADBEEF) out[threadIdx.x] = r + c;
}

template <int OP>
void run(const char* name, uint32_t* out) {
  const int iters = 20000;
  for (int wps : {4}) {
    const int blocks = 256 * wps;  // 4 waves per block -> wps waves per SIMD when one block lands per CU slot
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, 7u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 32 * wps;
    printf("%-18s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cyc at 2.4 GHz)\n", name, wps, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  }
}

int main() {
  uint32_t* out;
  hipMalloc(&out, 4096);
  run<0>("v_and_b32", out);
  run<1>("v_bcnt_u32_b32", out);
  run<2>("v_alignbit_b32", out);
  run<3>("v_cndmask_b32", out);
  run<4>("v_xor_b32", out);
  run<5>("v_lshrrev_b64", out);
  run<6>("v_cmp_ne_u32 sgpr", out);
  run<7>("v_add_f32", out);
  run<8>("v_fma_f32", out);
  run<9>("v_mul_lo_u32", out);
  run<10>("v_mad_u32_u24", out);
  run<11>("v_bfe_u32", out);
  run<12>("v_add_u32", out);
  run<13>("v_add_f64", out);
  run<14>("v_fma_f64", out);
  run<15>("v_mul_f64", out);
  run<16>("v_mov_dpp wave_shr", out);
  run<17>("v_lshl_or_b32", out);
  run<18>("v_and_or_b32", out);
  run<19>("v_cndmask e64 sgpr", out);
  run<20>("c++ select", out);
  run<21>("v_bfi_b32", out);
  run<22>("v_xad_u32", out);
  run<23>("v_or3_b32", out);
  run<24>("v_lshrrev_b32", out);
  run<25>("v_min_u32", out);
  run<26>("v_pk_min_u16", out);
  run<27>("v_sub_u32", out);
  run<28>("v_max3_u32", out);
  run<29>("v_mul_f32", out);
  run<30>("v_mul_u32_u24", out);
  run<31>("v_sad_u8", out);
  run<32>("v_bitop3_b32", out);
  run<33>("v_xnor_b32", out);
  run<34>("v_or_b32", out);
  run<35>("v_not_b32", out);
  run<36>("v_min_f32", out);
  run<37>("v_max_f32", out);
  run<38>("v_med3_f32", out);
  run<39>("v_min3_f32", out);
  run<40>("v_max3_f32", out);
  run<41>("v_pk_min_f16", out);
  run<42>("v_pk_max_f16", out);
  run<43>("v_pk_add_u16", out);
  run<44>("v_perm_b32", out);
  run<45>("v_cvt_f32_ubyte0", out);
  run<46>("v_cmp_gt_u32 vcc", out);
  run<47>("v_cmp_lt_i16_sdwa", out);
  run<48>("v_addc_co_u32", out);
  run<49>("v_sub_co_u32", out);
  run<50>("v_lshlrev_b32", out);
  run<51>("v_ashrrev_i32", out);
  run<52>("v_max_u32", out);
  run<53>("v_min_i32", out);
  run<54>("v_add3_u32", out);
  run<55>("v_mov_b32", out);
  run<56>("v_mov_dpp row_shr", out);
  run<57>("v_and_b32 literal", out);
  run<58>("v_med3_i32", out);
  run<59>("v_dot4_u32_u8", out);
  run<60>("v_pk_fma_f32", out);
  run<61>("v_min_u16", out);
  run<62>("v_pk_max_i16", out);
  run<63>("v_mbcnt_lo", out);
  run<64>("v_cndmask vcc", out);
  run<65>("and+bcnt pair (x2)", out);
  run<66>("v_alignbyte_b32", out);
  run<67>("v_sub_u16", out);
  run<68>("v_pk_sub_i16", out);
  run<69>("v_pk_lshrrev_b16", out);
  run<70>("v_max_f16", out);
  run<71>("v_fmac_f32", out);
  run<72>("v_sub_f32", out);
  run<73>("v_cmp_gt_f32 vcc", out);
  run<74>("v_cmp_lt_f32 sgpr", out);
  return 0;
}
