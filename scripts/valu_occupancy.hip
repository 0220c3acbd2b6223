// VALU issue rate against the number of resident waves per SIMD, and a VGPR-bank probe (source registers in the same /
// different banks: no difference).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_occupancy scripts/valu_occupancy.hip
// Measured on MI355X (cycles per wave-instruction per SIMD at 2.4 GHz, loop overhead included; 1 / 2 / 3 / 4 / 5 / 6 / 8 waves):
//   v_and 8.5 / 4.3 / 3.2 / 3.15 / 2.8 / 2.8 / 2.5, v_bcnt 9.5 / 5.9 / 5.7 / 5.0 / 4.7 / 4.6 / 4.5, v_bitop3 9.4 / 4.7 / 4.1 / 3.3 / 3.0 / 2.95 / 2.8:
//   one wave issues a VALU instruction every ~8-9 cycles at best, a SIMD needs four or more resident waves to approach its rate.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
  unsigned r = 0;
  asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n"
               "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n"
               "v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n"
               "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n"
               :: "v"(threadIdx.x) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
  for (int it = 0; it < iters; ++it) {
    if (V == 0)  // dst/src0 bank b, src1 bank b+1 (different)
      asm volatile("v_and_b32 v20, v20, v29\n v_and_b32 v21, v21, v30\n v_and_b32 v22, v22, v31\n v_and_b32 v23, v23, v28\n"
                   "v_and_b32 v24, v24, v33\n v_and_b32 v25, v25, v34\n v_and_b32 v26, v26, v35\n v_and_b32 v27, v27, v32\n"
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27");
    if (V == 1)  // src0 and src1 in the SAME bank
      asm volatile("v_and_b32 v20, v20, v28\n v_and_b32 v21, v21, v29\n v_and_b32 v22, v22, v30\n v_and_b32 v23, v23, v31\n"
                   "v_and_b32 v24, v24, v32\n v_and_b32 v25, v25, v33\n v_and_b32 v26, v26, v34\n v_and_b32 v27, v27, v35\n"
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27");
    if (V == 2)  // bcnt different banks
      asm volatile("v_bcnt_u32_b32 v20, v29, v20\n v_bcnt_u32_b32 v21, v30, v21\n v_bcnt_u32_b32 v22, v31, v22\n v_bcnt_u32_b32 v23, v28, v23\n"
                   "v_bcnt_u32_b32 v24, v33, v24\n v_bcnt_u32_b32 v25, v34, v25\n v_bcnt_u32_b32 v26, v35, v26\n v_bcnt_u32_b32 v27, v32, v27\n"
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27");
    if (V == 3)  // bcnt same bank
      asm volatile("v_bcnt_u32_b32 v20, v28, v20\n v_bcnt_u32_b32 v21, v29, v21\n v_bcnt_u32_b32 v22, v30, v22\n v_bcnt_u32_b32 v23, v31, v23\n"
                   "v_bcnt_u32_b32 v24, v32, v24\n v_bcnt_u32_b32 v25, v33, v25\n v_bcnt_u32_b32 v26, v34, v26\n v_bcnt_u32_b32 v27, v35, v27\n"
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27");
    if (V == 4)  // bitop3 three different banks
      asm volatile("v_bitop3_b32 v20, v20, v29, v34 bitop3:0x90\n v_bitop3_b32 v21, v21, v30, v35 bitop3:0x90\n v_bitop3_b32 v22, v22, v31, v32 bitop3:0x90\n v_bitop3_b32 v23, v23, v28, v33 bitop3:0x90\n"
                   "v_bitop3_b32 v24, v24, v33, v30 bitop3:0x90\n v_bitop3_b32 v25, v25, v34, v31 bitop3:0x90\n v_bitop3_b32 v26, v26, v35, v28 bitop3:0x90\n v_bitop3_b32 v27, v27, v32, v29 bitop3:0x90\n"
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27");
    if (V == 5)  // bitop3 all three in the same bank
      asm volatile("v_bitop3_b32 v20, v20, v28, v32 bitop3:0x90\n v_bitop3_b32 v21, v21, v29, v33 bitop3:0x90\n v_bitop3_b32 v22, v22, v30, v34 bitop3:0x90\n v_bitop3_b32 v23, v23, v31, v35 bitop3:0x90\n"
                   "v_bitop3_b32 v24, v24, v32, v28 bitop3:0x90\n v_bitop3_b32 v25, v25, v33, v29 bitop3:0x90\n v_bitop3_b32 v26, v26, v34, v30 bitop3:0x90\n v_bitop3_b32 v27, v27, v35, v31 bitop3:0x90\n"
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27");
    if (V == 6)  // v_and with an SGPR-free constant source (inline constant): one VGPR read
      asm volatile("v_and_b32 v20, 63, v20\n v_and_b32 v21, 63, v21\n v_and_b32 v22, 63, v22\n v_and_b32 v23, 63, v23\n"
                   "v_and_b32 v24, 63, v24\n v_and_b32 v25, 63, v25\n v_and_b32 v26, 63, v26\n v_and_b32 v27, 63, v27\n"
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27");
    if (V == 7)  // v_mov only
      asm volatile("v_mov_b32 v20, v29\n v_mov_b32 v21, v30\n v_mov_b32 v22, v31\n v_mov_b32 v23, v28\n"
                   "v_mov_b32 v24, v33\n v_mov_b32 v25, v34\n v_mov_b32 v26, v35\n v_mov_b32 v27, v32\n"
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27");
  }
  asm volatile("v_xor_b32 %0, v20, v21\n v_xor_b32 %0, %0, v22\n v_xor_b32 %0, %0, v23\n v_xor_b32 %0, %0, v24\n v_xor_b32 %0, %0, v25\n v_xor_b32 %0, %0, v26\n v_xor_b32 %0, %0, v27" : "=v"(r));
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int V> void run(const char* name, unsigned* out, int wps) {
  const int iters = 20000, blocks = 256 * wps;  // 256 CUs x (wps waves per SIMD: a block of 256 threads = 4 waves = 1 per SIMD)
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ns_per = ms * 1e6 / ((double)iters * 8 * wps);
  printf("%-34s waves/SIMD %d: %.3f ms, %.2f cyc per wave-instr per SIMD at 2.4 GHz\n", name, wps, ms, ns_per * 2.4);
}
int main() {
  unsigned* out; hipMalloc(&out, 256 * 16 * 256 * 4);
  for (int wps : {3, 4, 5, 6, 8}) {
    run<0>("v_and  src banks differ", out, wps);
    run<1>("v_and  src banks equal", out, wps);
    run<2>("v_bcnt src banks differ", out, wps);
    run<3>("v_bcnt src banks equal", out, wps);
    run<4>("v_bitop3 three banks", out, wps);
    run<5>("v_bitop3 one bank", out, wps);
    run<6>("v_and inline const", out, wps);
    run<7>("v_mov", out, wps);
  }
  return 0;
}
