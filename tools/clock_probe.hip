// clock_probe.hip -- effective shader clock seen by a single-workgroup latency-bound kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void chain(double *out, long long *cyc, int n, double a) {
  double x = out[0];
  long long t0 = __builtin_readcyclecounter();       // s_memtime: shader cycles
  long long w0 = wall_clock64();                      // constant 100 MHz
  for (int i = 0; i < n; ++i) x = __builtin_fma(x, a, 1.0);
  long long t1 = __builtin_readcyclecounter();
  long long w1 = wall_clock64();
  if (threadIdx.x == 0) { out[1] = x; cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
__global__ void busy(float *p, int n) { float x = p[threadIdx.x]; for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f; p[blockIdx.x * blockDim.x + threadIdx.x] = x; }
int main() {
  double *d; long long *c; float *f;
  hipMalloc(&d, 16); hipMalloc(&c, 16); hipMalloc(&f, 4 * 256 * 2048);
  hipMemset(d, 0, 16); hipMemset(f, 0, 4 * 256 * 2048);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    int n = 200000;
    if (rep == 3) busy<<<2048, 256>>>(f, 200000);     // load the chip next to the probe (other stream would be better)
    hipEventRecord(e0);
    chain<<<1, 64>>>(d, c, n, 0.999999);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    printf("rep %d: %d dependent f64 FMAs: %.1f us wall, %lld shader cycles (%.2f cyc/op), wall_clock64 %.1f us => shader clock %.0f MHz\n",
           rep, n, ms * 1e3, h[0], (double)h[0] / n, h[1] / 100.0, h[0] / (h[1] / 100.0));
  }
  return 0;
}
