// Shader clock of the chip WHILE another process loads it: a few single-wave workgroups spin for `ms` milliseconds and
// sample s_memtime (shader cycles) against s_memrealtime (100 MHz) every ~50 us.  Run it beside the bench:
//     python bench.py --steps 400 --no-cpu --no-h2d --no-sub &  ./scripts/clock_probe 3000
// prints the clock per probe wave (min / mean / max over its samples).  Build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kSamples = 4096;

__global__ __launch_bounds__(64) void probe(float* ghz, int* nsamp, unsigned long long run_ticks) {
  unsigned long long t0, r0, t, r, rstart;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
  rstart = r0;
  int n = 0;
  while (true) {
    do {  // ~50 us between samples (5000 ticks of the 100 MHz counter); s_sleep keeps the probe off the issue ports
      __builtin_amdgcn_s_sleep(64);
      asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r));
    } while (r - r0 < 5000ULL);
    if (threadIdx.x == 0 && n < kSamples) ghz[blockIdx.x * kSamples + n] = (float)((double)(t - t0) / ((double)(r - r0) * 10.0));
    ++n;
    t0 = t;
    r0 = r;
    if (r - rstart >= run_ticks) break;
  }
  if (threadIdx.x == 0) nsamp[blockIdx.x] = n < kSamples ? n : kSamples;
}

int main(int argc, char** argv) {
  const double ms = argc > 1 ? atof(argv[1]) : 2000.0;
  const int waves = 16;
  float* ghz;
  int* ns;
  hipMalloc(&ghz, sizeof(float) * waves * kSamples);
  hipMalloc(&ns, sizeof(int) * waves);
  hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, ghz, ns, (unsigned long long)(ms * 1e5));
  hipDeviceSynchronize();
  std::vector<float> h(waves * kSamples);
  std::vector<int> hn(waves);
  hipMemcpy(h.data(), ghz, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  hipMemcpy(hn.data(), ns, hn.size() * sizeof(int), hipMemcpyDeviceToHost);
  double gmin = 1e9, gmax = 0, gsum = 0;
  long gn = 0;
  for (int w = 0; w < waves; ++w) {
    double mn = 1e9, mx = 0, sm = 0;
    for (int i = 0; i < hn[w]; ++i) {
      const double v = h[w * kSamples + i];
      mn = std::min(mn, v);
      mx = std::max(mx, v);
      sm += v;
    }
    if (hn[w]) {
      printf("probe wave %2d: %4d samples, shader clock min %.3f mean %.3f max %.3f GHz\n", w, hn[w], mn, sm / hn[w], mx);
      gmin = std::min(gmin, mn);
      gmax = std::max(gmax, mx);
      gsum += sm;
      gn += hn[w];
    }
  }
  printf("all probes: min %.3f mean %.3f max %.3f GHz over %.0f ms\n", gmin, gn ? gsum / gn : 0.0, gmax, ms);
  return 0;
}
