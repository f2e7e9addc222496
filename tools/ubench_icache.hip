// ubench_icache.hip -- cost of straight-line (fully unrolled) code executed once per wave vs the same work in a loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int REP>
__global__ void fma_once(double *out, long long *cyc, int n, double a) {
  double x0 = out[0] + threadIdx.x, x1 = out[1], x2 = out[2], x3 = out[3];
  long long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int r = 0; r < REP; ++r) { x0 = __builtin_fma(x0, a, 1.0); x1 = __builtin_fma(x1, a, 2.0); x2 = __builtin_fma(x2, a, 3.0); x3 = __builtin_fma(x3, a, 4.0); }
  }
  long long w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x + 64] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = w0; cyc[2 * blockIdx.x + 1] = w1; }
}
template <typename K> void run(const char *name, K kern, int blocks, long long *c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  static long long h[2 * 4096];
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); kern(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, c, 16 * blocks, hipMemcpyDeviceToHost);
    double avg = 0, mx = 0; long long t0 = h[0];
    for (int b = 0; b < blocks; ++b) { t0 = t0 < h[2 * b] ? t0 : h[2 * b]; }
    long long t1 = 0;
    for (int b = 0; b < blocks; ++b) { double d = (h[2 * b + 1] - h[2 * b]) / 100.0; avg += d; mx = d > mx ? d : mx; t1 = t1 > h[2 * b + 1] ? t1 : h[2 * b + 1]; }
    printf("%-40s launch %d: event %.1f us, per-wave body avg %.2f max %.2f us, span %.1f us\n", name, rep, ms * 1e3, avg / blocks, mx, (t1 - t0) / 100.0);
  }
}
int main() {
  double *d; long long *c;
  hipMalloc(&d, 8 * (1 << 22)); hipMalloc(&c, 16 * 4096); hipMemset(d, 0, 8 * (1 << 22));
  // 4096 FMAs per wave either way (8 B each => 32 KB straight-line)
  run("straight-line 32 KB, 1 WG x 256", [&]() { fma_once<1024><<<1, 256>>>(d, c, 1, 0.999999); }, 1, c);
  run("loop 64 x 64 FMAs, 1 WG x 256", [&]() { fma_once<16><<<1, 256>>>(d, c, 64, 0.999999); }, 1, c);
  run("straight-line 32 KB, 403 WG x 256", [&]() { fma_once<1024><<<403, 256>>>(d, c, 1, 0.999999); }, 403, c);
  run("loop 64 x 64 FMAs, 403 WG x 256", [&]() { fma_once<16><<<403, 256>>>(d, c, 64, 0.999999); }, 403, c);
  run("straight-line 8 KB, 403 WG x 256", [&]() { fma_once<256><<<403, 256>>>(d, c, 1, 0.999999); }, 403, c);
  run("loop 16 x 64 FMAs, 403 WG x 256", [&]() { fma_once<16><<<403, 256>>>(d, c, 16, 0.999999); }, 403, c);
  return 0;
}
