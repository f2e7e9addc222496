// common.h -- context, error plumbing and wave64 helpers shared by the gfx950 kernels.
// Target: MI355X (gfx950, CDNA4, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/scavislam_hip.h"

struct svs_ctx {
  int device = 0;
  int n_cu = 256;                 // compute units of the device (MI355X: 256)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
};

#define SVS_HIP(ctx, call)                                                              \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      char buf_[512];                                                                   \
      snprintf(buf_, sizeof buf_, "%s:%d %s -> %s", __FILE__, __LINE__, #call,          \
               hipGetErrorString(e_));                                                  \
      (ctx)->err = buf_;                                                                \
      return SVS_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define SVS_REQUIRE(ctx, cond)                                                          \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      char buf_[512];                                                                   \
      snprintf(buf_, sizeof buf_, "%s:%d requirement failed: %s", __FILE__, __LINE__,   \
               #cond);                                                                  \
      if (ctx) (ctx)->err = buf_;                                                       \
      return SVS_ERR_INVALID;                                                           \
    }                                                                                   \
  } while (0)

#define SVS_LAUNCH_CHECK(ctx) SVS_HIP(ctx, hipGetLastError())

__host__ __device__ static inline int div_up(int a, int b) { return (a + b - 1) / b; }

// ---- wave64 reductions (DPP/bpermute via __shfl; no LDS, no volatile warp idioms) -----------
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
