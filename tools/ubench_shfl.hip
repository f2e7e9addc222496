// ubench_shfl.hip -- cross-lane f64 movement cost on gfx950: ds_bpermute (shfl) vs DPP row shifts, 1 and 8 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ double dpp_row_shr1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true);   // row_shr:1, bound_ctrl -> 0 at row start
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(double *out, long long *cyc, int n) {
  double x[9];
  for (int c = 0; c < 9; ++c) x[c] = out[c] + threadIdx.x;
  __syncthreads();
  long long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        double up;
        if (MODE == 0) up = __shfl_up(x[c], 1 << r, 64);
        else up = dpp_row_shr1(x[c]);
        x[c] += up;
      }
  }
  long long w1 = wall_clock64();
  double s = 0;
  for (int c = 0; c < 9; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x + 64] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = w1 - w0;
}
template <typename K> void run(const char *name, K kern, int threads, int n, long long *c) {
  for (int rep = 0; rep < 2; ++rep) {
    kern(threads, n); hipDeviceSynchronize();
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    if (rep) printf("%-36s %4d thr: %8.1f us => %.2f ns per f64 shift+add per wave\n", name, threads, h / 100.0, (h * 10.0) / (n * 36.0));
  }
}
int main() {
  double *d; long long *c;
  hipMalloc(&d, 8 * (1 << 20)); hipMalloc(&c, 16); hipMemset(d, 0, 8 * (1 << 20));
  const int n = 500;
  for (int t : {64, 256, 512, 1024}) {
    run("shfl_up f64 (2x ds_bpermute) + add", [&](int th, int n) { k<0><<<1, th>>>(d, c, n); }, t, n, c);
    run("dpp row_shr:1 f64 (2x v_mov_dpp) + add", [&](int th, int n) { k<1><<<1, th>>>(d, c, n); }, t, n, c);
  }
  return 0;
}
