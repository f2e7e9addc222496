// ubench.hip -- MI355X micro-benchmarks behind the BA kernel design (f64 VALU latency/issue, LDS f64 atomics).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench.hip -o /tmp/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
template <int CHAINS>
__global__ void fma_chain(double *out, long long *cyc, int n, double a) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = out[c] + threadIdx.x;
  __syncthreads();
  long long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, 1.0);
  }
  long long w1 = wall_clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x + 64] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = w1 - w0;
}
// MODE 0: every lane its own address (stride 8 B); 1: all lanes one address; 2: 13-way groups; 3: stride 36 doubles (bank pattern of
// the Schur window); 4: stride 37 doubles
template <int MODE>
__global__ void lds_atomic(double *out, long long *cyc, int n) {
  __shared__ double s[64 * 40 * 4];
  for (int i = threadIdx.x; i < 64 * 40 * 4; i += blockDim.x) s[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int idx;
  if (MODE == 0) idx = lane;
  else if (MODE == 1) idx = 0;
  else if (MODE == 2) idx = lane % 5;
  else if (MODE == 3) idx = (lane % 16) * 36;
  else idx = (lane % 16) * 37;
  double *p = s + w * 64 * 40 + idx;
  long long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int r = 0; r < 36; ++r) __hip_atomic_fetch_add(p + r * (MODE == 0 ? 64 : 1), 1.0 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  long long w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[threadIdx.x];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = w1 - w0;
}
template <typename K>
void run(const char *name, K kern, int blocks, int threads, double ops_per_thread_iter, int n, double *d, long long *c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    kern(blocks, threads, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    if (rep) printf("%-44s blocks %4d x %4d thr: %8.1f us, block0 %8.1f us => %.2f ns per wave-instr per wave\n", name, blocks, threads, ms * 1e3, h / 100.0,
                    (h * 10.0) / (n * ops_per_thread_iter));
  }
}
int main() {
  double *d; long long *c;
  hipMalloc(&d, 8 * (1 << 22)); hipMalloc(&c, 16); hipMemset(d, 0, 8 * (1 << 22));
  const int n = 2000;
  run("f64 fma, 1 dependent chain", [&](int b, int t, int n) { fma_chain<1><<<b, t>>>(d, c, n, 0.999999); }, 1, 64, REP * 1, n, d, c);
  run("f64 fma, 2 chains", [&](int b, int t, int n) { fma_chain<2><<<b, t>>>(d, c, n, 0.999999); }, 1, 64, REP * 2, n, d, c);
  run("f64 fma, 4 chains", [&](int b, int t, int n) { fma_chain<4><<<b, t>>>(d, c, n, 0.999999); }, 1, 64, REP * 4, n, d, c);
  run("f64 fma, 8 chains", [&](int b, int t, int n) { fma_chain<8><<<b, t>>>(d, c, n, 0.999999); }, 1, 64, REP * 8, n, d, c);
  run("f64 fma, 1 chain, 4 waves/CU (1/SIMD)", [&](int b, int t, int n) { fma_chain<1><<<b, t>>>(d, c, n, 0.999999); }, 1, 256, REP, n, d, c);
  run("f64 fma, 1 chain, 8 waves/CU (2/SIMD)", [&](int b, int t, int n) { fma_chain<1><<<b, t>>>(d, c, n, 0.999999); }, 1, 512, REP, n, d, c);
  run("f64 fma, 1 chain, 16 waves/CU (4/SIMD)", [&](int b, int t, int n) { fma_chain<1><<<b, t>>>(d, c, n, 0.999999); }, 1, 1024, REP, n, d, c);
  run("f64 fma, 8 chains, 16 waves/CU", [&](int b, int t, int n) { fma_chain<8><<<b, t>>>(d, c, n, 0.999999); }, 1, 1024, REP * 8, n, d, c);
  run("f64 fma, 8 chains, 16 waves/CU, 256 CUs", [&](int b, int t, int n) { fma_chain<8><<<b, t>>>(d, c, n, 0.999999); }, 256, 1024, REP * 8, n, d, c);
  const int m = 500;
  run("lds add f64, distinct addr, 1 wave", [&](int b, int t, int n) { lds_atomic<0><<<b, t>>>(d, c, n); }, 1, 64, 36, m, d, c);
  run("lds add f64, one addr (64-way), 1 wave", [&](int b, int t, int n) { lds_atomic<1><<<b, t>>>(d, c, n); }, 1, 64, 36, m, d, c);
  run("lds add f64, 5 addrs (13-way), 1 wave", [&](int b, int t, int n) { lds_atomic<2><<<b, t>>>(d, c, n); }, 1, 64, 36, m, d, c);
  run("lds add f64, 16 addrs stride 36, 1 wave", [&](int b, int t, int n) { lds_atomic<3><<<b, t>>>(d, c, n); }, 1, 64, 36, m, d, c);
  run("lds add f64, 16 addrs stride 37, 1 wave", [&](int b, int t, int n) { lds_atomic<4><<<b, t>>>(d, c, n); }, 1, 64, 36, m, d, c);
  run("lds add f64, distinct addr, 4 waves", [&](int b, int t, int n) { lds_atomic<0><<<b, t>>>(d, c, n); }, 1, 256, 36, m, d, c);
  run("lds add f64, 13-way, 4 waves", [&](int b, int t, int n) { lds_atomic<2><<<b, t>>>(d, c, n); }, 1, 256, 36, m, d, c);
  run("lds add f64, stride 36, 4 waves", [&](int b, int t, int n) { lds_atomic<3><<<b, t>>>(d, c, n); }, 1, 256, 36, m, d, c);
  run("lds add f64, stride 37, 4 waves", [&](int b, int t, int n) { lds_atomic<4><<<b, t>>>(d, c, n); }, 1, 256, 36, m, d, c);
  return 0;
}
