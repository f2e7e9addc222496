// image.hip -- context + preprocessing kernels (u8 pyramid step, f32 convert + Sobel).
// Replaces FrameGrabber::preprocessing (frame_grabber.cpp:285-336): cv::buildPyramid on u8 and
// the CPU path's convertTo(CV_32F,1/255.) + Sobel(ksize=1).  Integer path is bit-exact to the
// OpenCV 2.4.2 semantics restated in SURVEY.md A.2; HBM-bound stencils, LDS-tiled.
#include "common.h"

// ---------------------------------------------------------------------------------------------
extern "C" int svs_ctx_create(int device, void *hip_stream, svs_ctx **out) {
  if (!out) return SVS_ERR_INVALID;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SVS_ERR_NO_DEVICE;
  svs_ctx *c = new svs_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return SVS_ERR_NO_DEVICE; }
  if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SVS_ERR_HIP; }
    c->own_stream = true;
  }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { delete c; return SVS_ERR_HIP; }
  *out = c;
  return SVS_OK;
}
extern "C" int svs_ctx_destroy(svs_ctx *c) {
  if (!c) return SVS_OK;
  (void)hipStreamSynchronize(c->stream);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return SVS_OK;
}
extern "C" int svs_ctx_sync(svs_ctx *c) { SVS_REQUIRE(c, c); SVS_HIP(c, hipStreamSynchronize(c->stream)); return SVS_OK; }
extern "C" void *svs_ctx_stream(svs_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" const char *svs_last_error(svs_ctx *c) { return c ? c->err.c_str() : "null ctx"; }
extern "C" int svs_malloc(svs_ctx *c, size_t bytes, void **p) {
  SVS_REQUIRE(c, c && p);
  SVS_HIP(c, hipSetDevice(c->device));
  SVS_HIP(c, hipMalloc(p, bytes ? bytes : 1));
  return SVS_OK;
}
extern "C" int svs_free(svs_ctx *c, void *p) { SVS_REQUIRE(c, c); if (p) SVS_HIP(c, hipFree(p)); return SVS_OK; }
extern "C" int svs_memcpy_h2d(svs_ctx *c, void *d, const void *h, size_t n) {
  SVS_REQUIRE(c, c);
  SVS_HIP(c, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->stream));
  SVS_HIP(c, hipStreamSynchronize(c->stream));  // h may be pageable and reused by the caller
  return SVS_OK;
}
extern "C" int svs_memcpy_d2h(svs_ctx *c, void *h, const void *d, size_t n) {
  SVS_REQUIRE(c, c);
  SVS_HIP(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream));
  SVS_HIP(c, hipStreamSynchronize(c->stream));
  return SVS_OK;
}
extern "C" int svs_timer_start(svs_ctx *c) { SVS_REQUIRE(c, c); SVS_HIP(c, hipEventRecord(c->ev0, c->stream)); return SVS_OK; }
extern "C" int svs_timer_stop_ms(svs_ctx *c, float *ms) {
  SVS_REQUIRE(c, c && ms);
  SVS_HIP(c, hipEventRecord(c->ev1, c->stream));
  SVS_HIP(c, hipEventSynchronize(c->ev1));
  SVS_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return SVS_OK;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) { i = i < 0 ? -i : 2 * n - 2 - i; }
  return i;
}

// One pyrDown step.  Block = 32x8 output pixels; source footprint 67x19 staged in LDS, then the
// separable [1 4 6 4 1] passes run out of LDS (horizontal into u16, vertical into the output).
constexpr int PD_TX = 32, PD_TY = 8, PD_SW = 2 * PD_TX + 3, PD_SH = 2 * PD_TY + 3;
__global__ __launch_bounds__(256) void pyr_down_u8_kernel(const uint8_t *__restrict__ src, int w, int h, int ss,
                                                          size_t sb, uint8_t *__restrict__ dst, int dw, int dh,
                                                          int ds, size_t db) {
  __shared__ uint8_t s_src[PD_SH][PD_SW + 1];
  __shared__ uint16_t s_h[PD_SH][PD_TX];
  const int tid = threadIdx.y * PD_TX + threadIdx.x;
  const int ox0 = blockIdx.x * PD_TX, oy0 = blockIdx.y * PD_TY;
  src += (size_t)blockIdx.z * sb;
  dst += (size_t)blockIdx.z * db;
  const int sx0 = 2 * ox0 - 2, sy0 = 2 * oy0 - 2;
  for (int i = tid; i < PD_SH * PD_SW; i += 256) {
    int r = i / PD_SW, c = i - r * PD_SW;
    int y = reflect101(sy0 + r, h), x = reflect101(sx0 + c, w);
    s_src[r][c] = src[(size_t)y * ss + x];
  }
  __syncthreads();
  for (int i = tid; i < PD_SH * PD_TX; i += 256) {
    int r = i / PD_TX, c = i - r * PD_TX;
    const uint8_t *p = &s_src[r][2 * c];
    s_h[r][c] = (uint16_t)(p[0] + 4 * p[1] + 6 * p[2] + 4 * p[3] + p[4]);
  }
  __syncthreads();
  const int ox = ox0 + threadIdx.x, oy = oy0 + threadIdx.y;
  if (ox < dw && oy < dh) {
    int r = 2 * threadIdx.y, c = threadIdx.x;
    int acc = s_h[r][c] + 4 * s_h[r + 1][c] + 6 * s_h[r + 2][c] + 4 * s_h[r + 3][c] + s_h[r + 4][c];
    dst[(size_t)oy * ds + ox] = (uint8_t)((acc + 128) >> 8);
  }
}

extern "C" int svs_pyr_down_u8(svs_ctx *ctx, const uint8_t *d_src, int w, int h, int sstride, size_t s_bstride,
                               uint8_t *d_dst, int dstride, size_t d_bstride, int batch) {
  SVS_REQUIRE(ctx, ctx && d_src && d_dst && w >= 3 && h >= 3 && batch >= 1);
  int dw = (w + 1) / 2, dh = (h + 1) / 2;
  dim3 grid(div_up(dw, PD_TX), div_up(dh, PD_TY), batch), block(PD_TX, PD_TY);
  hipLaunchKernelGGL(pyr_down_u8_kernel, grid, block, 0, ctx->stream, d_src, w, h, sstride, s_bstride, d_dst, dw,
                     dh, dstride, d_bstride);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

// convertTo(CV_32F, 1/255.) then Sobel(ksize=1): dx = I(x+1)-I(x-1), dy = I(y+1)-I(y-1),
// BORDER_REFLECT_101 (=> 0 on the border).  One thread per 4 consecutive pixels: 5 u8 reads
// per row neighbourhood, 3 x float4 coalesced stores.
__global__ __launch_bounds__(256) void convert_sobel_kernel(const uint8_t *__restrict__ src, int w, int h, int ss,
                                                            size_t sb, float *__restrict__ img, float *__restrict__ dx,
                                                            float *__restrict__ dy, int fs, size_t fb) {
  const float sc = (float)(1. / 255.);
  int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  int y = blockIdx.y;
  if (x4 >= w) return;
  src += (size_t)blockIdx.z * sb;
  size_t fo = (size_t)blockIdx.z * fb + (size_t)y * fs;
  const uint8_t *row = src + (size_t)y * ss;
  const uint8_t *rowm = src + (size_t)reflect101(y - 1, h) * ss;
  const uint8_t *rowp = src + (size_t)reflect101(y + 1, h) * ss;
  float c[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) c[k] = (float)row[reflect101(x4 - 1 + k, w)] * sc;
  if (x4 + 3 < w && (fs & 3) == 0 && (fb & 3) == 0 && (((uintptr_t)img | (uintptr_t)dx | (uintptr_t)dy) & 15) == 0) {
    // full group of 4: three 16-byte stores per lane (1 KiB per wave-instruction)
    float4 vi = make_float4(c[1], c[2], c[3], c[4]);
    float4 vx = make_float4(c[2] - c[0], c[3] - c[1], c[4] - c[2], c[5] - c[3]);
    float4 vy;
    vy.x = (float)rowp[x4] * sc - (float)rowm[x4] * sc;
    vy.y = (float)rowp[x4 + 1] * sc - (float)rowm[x4 + 1] * sc;
    vy.z = (float)rowp[x4 + 2] * sc - (float)rowm[x4 + 2] * sc;
    vy.w = (float)rowp[x4 + 3] * sc - (float)rowm[x4 + 3] * sc;
    *reinterpret_cast<float4 *>(img + fo + x4) = vi;
    *reinterpret_cast<float4 *>(dx + fo + x4) = vx;
    *reinterpret_cast<float4 *>(dy + fo + x4) = vy;
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int x = x4 + k;
    if (x < w) {
      img[fo + x] = c[k + 1];
      dx[fo + x] = c[k + 2] - c[k];
      dy[fo + x] = (float)rowp[x] * sc - (float)rowm[x] * sc;
    }
  }
}

extern "C" int svs_convert_sobel_f32(svs_ctx *ctx, const uint8_t *d_src, int w, int h, int sstride, size_t s_bstride,
                                     float *d_img, float *d_dx, float *d_dy, int fstride, size_t f_bstride,
                                     int batch) {
  SVS_REQUIRE(ctx, ctx && d_src && d_img && d_dx && d_dy && w >= 2 && h >= 2 && batch >= 1);
  dim3 block(64), grid(div_up(div_up(w, 4), 64), h, batch);
  hipLaunchKernelGGL(convert_sobel_kernel, grid, block, 0, ctx->stream, d_src, w, h, sstride, s_bstride, d_img,
                     d_dx, d_dy, fstride, f_bstride);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
