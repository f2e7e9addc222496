/* svs_cuda_emul.h -- TEST INFRASTRUCTURE ONLY (oracle/_ref recipe).
 *
 * A small host emulation of the CUDA 4.x constructs that /root/reference/scavislam/gpu/dense_tracking.{cuh,cu}
 * use, so that g++ can compile those two reference files AS THEY LIE (the only reference sources without OpenCV /
 * Eigen / g2o / Sophus dependencies) into oracle/_ref/libsvs_ref_gpu.so.  Nothing here is derived from the reference:
 * it is the execution model (grid of blocks of threads, __syncthreads, __shared__, 2-D linear-filtered textures)
 * restated from the CUDA C Programming Guide.  The product never includes or links this.
 *
 * Execution model: blocks run one after the other; the threads of a block are ucontext fibers resumed round-robin in
 * ascending thread id.  __syncthreads() yields, so every round advances each live thread to its next barrier.
 * The reference's warpReduce() relies on pre-Volta implicit warp-synchronous execution (all 32 lanes execute each
 * statement together, `volatile` shared memory); the recipe makes that explicit by inserting svs_lockstep() -- the
 * same yield -- after each statement of warpReduce (what __syncwarp() does in today's CUDA).  Because lane t only writes
 * element t and reads element t+off, resuming lanes in ascending order reproduces "all lanes read, then all lanes
 * write" exactly.
 *
 * Texture unit: tex2D(x, y) with cudaFilterModeLinear, unnormalised coordinates, clamp addressing (CUDA C Programming
 * Guide, "Texture Fetching / Linear Filtering"): xB = x - 0.5, i = floor(xB), alpha = frac(xB), result = sum of the four
 * neighbours weighted (1-a)(1-b), a(1-b), (1-a)b, ab.  The hardware keeps alpha/beta in 9-bit fixed point (8 fractional
 * bits); that quantisation is device behaviour, selectable here with svs_emul::tex_frac_bits = 8 (default: exact weights,
 * -1).  The four products are added in the order of the reference's own software bilinear (maths_utils.cpp:46-65:
 * (x,y), (x,y+1), (x+1,y), (x+1,y+1)) -- the hardware's order is not documented. */
#ifndef SVS_CUDA_EMUL_H
#define SVS_CUDA_EMUL_H
#include <ucontext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <tuple>
#include <utility>
#include <vector>

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
struct uint3 { unsigned x, y, z; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

#define __host__
#define __device__
#define __global__
#define __shared__ static

static uint3 threadIdx, blockIdx;
static dim3 blockDim, gridDim;

static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float min(float a, float b) { return fminf(a, b); }

namespace svs_emul {
static int tex_frac_bits = -1;        /* -1: exact bilinear weights; 8: the texture unit's 1.8 fixed-point weights */
struct Fiber { ucontext_t ctx; char *stack; bool done; uint3 tid; };
static ucontext_t g_main;
static Fiber *g_cur = nullptr;
static std::function<void()> *g_body = nullptr;
static const size_t STACK_BYTES = 256 * 1024;
static void trampoline() {
  (*g_body)();
  g_cur->done = true;
  swapcontext(&g_cur->ctx, &g_main);
}
static inline void yield() { swapcontext(&g_cur->ctx, &g_main); }
static void run_grid(dim3 grid, dim3 block, std::function<void()> body) {
  const unsigned nthr = block.x * block.y * block.z;
  static std::vector<Fiber> fibers;
  while (fibers.size() < nthr) { Fiber f; f.stack = (char *)malloc(STACK_BYTES); f.done = true; fibers.push_back(f); }
  g_body = &body;
  gridDim = grid; blockDim = block;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
        unsigned t = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
              Fiber &f = fibers[t];
              getcontext(&f.ctx);
              f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK_BYTES; f.ctx.uc_link = nullptr;
              f.tid.x = tx; f.tid.y = ty; f.tid.z = tz; f.done = false;
              makecontext(&f.ctx, trampoline, 0);
            }
        for (bool live = true; live;) {
          live = false;
          for (unsigned i = 0; i < nthr; ++i) {
            Fiber &f = fibers[i];
            if (f.done) continue;
            g_cur = &f; threadIdx = f.tid;
            swapcontext(&g_main, &f.ctx);
            live = live || !f.done;
          }
        }
      }
  g_body = nullptr;
}
template <class... A> struct Bound { dim3 grid, block; std::tuple<A &&...> args; };
struct Cfg {
  dim3 grid, block;
  template <class... A> Bound<A...> operator()(A &&...a) const { return Bound<A...>{grid, block, std::forward_as_tuple(std::forward<A>(a)...)}; }
};
template <class K, class Tup, size_t... I> static void call(K k, Tup &t, std::index_sequence<I...>) { k(std::get<I>(t)...); }
}  // namespace svs_emul

/* `kernel<<<grid, block>>>(args...)` is rewritten by the recipe (sed, no file written) to
   `kernel >> svs_cfg(grid, block)(args...)` */
static inline svs_emul::Cfg svs_cfg(dim3 grid, dim3 block) { return svs_emul::Cfg{grid, block}; }
template <class K, class... A> static void operator>>(K kernel, svs_emul::Bound<A...> b) {
  svs_emul::run_grid(b.grid, b.block, [&]() { svs_emul::call(kernel, b.args, std::index_sequence_for<A...>{}); });
}
static inline void __syncthreads() { svs_emul::yield(); }
static inline void svs_lockstep() { svs_emul::yield(); }

/* ---- runtime API subset ---- */
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
enum cudaTextureReadMode { cudaReadModeElementType, cudaReadModeNormalizedFloat };
enum cudaTextureFilterMode { cudaFilterModePoint, cudaFilterModeLinear };
enum cudaChannelFormatKind { cudaChannelFormatKindSigned, cudaChannelFormatKindUnsigned, cudaChannelFormatKindFloat };
struct cudaChannelFormatDesc { int x, y, z, w; cudaChannelFormatKind f; };
static inline cudaChannelFormatDesc cudaCreateChannelDesc(int x, int y, int z, int w, cudaChannelFormatKind f) { cudaChannelFormatDesc d = {x, y, z, w, f}; return d; }
template <class T> static inline int cudaMalloc(T **p, size_t n) { *p = (T *)calloc(1, n); return *p ? 0 : 2; }
static inline int cudaFree(void *p) { free(p); return 0; }
static inline int cudaMemcpy(void *d, const void *s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return 0; }
static inline int cudaThreadSynchronize() { return 0; }

template <class T, int DIM, cudaTextureReadMode MODE> struct texture {
  cudaTextureFilterMode filterMode = cudaFilterModePoint;
  const T *ptr = nullptr; int width = 0, height = 0; size_t pitch = 0;
};
template <class T, int DIM, cudaTextureReadMode MODE>
static inline int cudaBindTexture2D(size_t *, texture<T, DIM, MODE> &t, const void *p, const cudaChannelFormatDesc &, size_t w, size_t h, size_t pitch_bytes) {
  t.ptr = (const T *)p; t.width = (int)w; t.height = (int)h; t.pitch = pitch_bytes; return 0;
}
template <int DIM, cudaTextureReadMode MODE> static inline float tex2D(const texture<float, DIM, MODE> &t, float x, float y) {
  auto at = [&](int i, int j) {
    i = i < 0 ? 0 : (i >= t.width ? t.width - 1 : i);      /* cudaAddressModeClamp (the default) */
    j = j < 0 ? 0 : (j >= t.height ? t.height - 1 : j);
    return *(const float *)((const char *)t.ptr + (size_t)j * t.pitch + (size_t)i * sizeof(float));
  };
  if (t.filterMode == cudaFilterModePoint) return at((int)floorf(x), (int)floorf(y));
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fi = floorf(xb), fj = floorf(yb);
  float a = xb - fi, b = yb - fj;
  if (svs_emul::tex_frac_bits >= 0) {
    const float q = (float)(1 << svs_emul::tex_frac_bits);
    a = floorf(a * q + 0.5f) / q; b = floorf(b * q + 0.5f) / q;
  }
  const int i = (int)fi, j = (int)fj;
  const float wx0 = 1 - a, wx1 = a, wy0 = 1 - b, wy1 = b;
  return (wx0 * wy0) * at(i, j) + (wx0 * wy1) * at(i, j + 1) + (wx1 * wy0) * at(i + 1, j) + (wx1 * wy1) * at(i + 1, j + 1);
}
#endif
