// Ad-hoc micro-benchmark: how long do N nearly-empty 256-thread workgroups take on MI355X?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS, int VG>
__global__ __launch_bounds__(256) void k(const int* n, int* out) {
  __shared__ double s[LDS > 0 ? LDS : 1];
  const int b = blockIdx.z;
  if ((int)blockIdx.x * 512 >= n[b]) return;
  double acc[VG];
  for (int i = 0; i < VG; ++i) acc[i] = threadIdx.x * 0.5 + i;
  for (int it = 0; it < 100; ++it)
    for (int i = 0; i < VG; ++i) acc[i] = acc[i] * 1.0001 + s[(threadIdx.x + i) % (LDS > 0 ? LDS : 1)];
  double t = 0;
  for (int i = 0; i < VG; ++i) t += acc[i];
  if (t == 12345.678) out[0] = 1;
}
template <int LDS, int VG>
void run(const char* name, dim3 grid, const int* n, int* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<LDS, VG>), grid, dim3(256), 0, 0, n, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep == 2) printf("%-28s grid %4d x %3d x %3d = %6d WGs: %.3f ms  (%.1f ns/WG)\n", name, grid.x, grid.y, grid.z, grid.x * grid.y * grid.z, ms, ms * 1e6 / (grid.x * grid.y * grid.z));
  }
}
int main() {
  int *n, *out; hipMalloc(&n, 64 * 4); hipMalloc(&out, 4);
  int h[64]; for (int i = 0; i < 64; ++i) h[i] = 0;   // every workgroup exits at once
  hipMemcpy(n, h, sizeof(h), hipMemcpyHostToDevice);
  run<0, 1>("no LDS, all exit", dim3(8, 16, 64), n, out);
  run<1600, 1>("12.8KB LDS, all exit", dim3(8, 16, 64), n, out);
  run<1600, 40>("12.8KB LDS+80 VGPR, all exit", dim3(8, 16, 64), n, out);
  run<1600, 40>("same, 2048 WGs", dim3(2, 16, 64), n, out);
  run<1600, 40>("same, 32768 WGs", dim3(32, 16, 64), n, out);
  for (int i = 0; i < 64; ++i) h[i] = 100000;       // nobody exits early
  hipMemcpy(n, h, sizeof(h), hipMemcpyHostToDevice);
  run<1600, 40>("12.8KB LDS+80 VGPR, work", dim3(8, 16, 64), n, out);
  return 0;
}
