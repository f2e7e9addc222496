// dense_full.hip -- full-resolution dense SE3 tracking for gfx950: the CUDA build's tracker.
// Replaces DenseTracker::denseTrackingGpu / computeDensePointCloudGpu (dense_tracking.cpp:60-215), the GpuTracker class
// with its four kernels (gpu/dense_tracking.cuh:281-342, gpu/dense_tracking.cu:82-569) and the CUDA branch of
// FrameGrabber::preprocessing (frame_grabber.cpp:291-313, filters :102-115).
//
// Arithmetic contract (oracle/vision.c, pinned bit for bit against the reference-compiled kernels in oracle/_ref):
// every per-pixel quantity -- transformed point, projection, in-frame test, texture position RN(uv + 0.5f) - 0.5f,
// bilinear taps, residual, the six Jacobian entries incl. the reference's `1./p.z` double promotions -- is the f32 value
// the reference's code produces (no contraction; svs_dense_pixel_terms_full exposes them for the bit-exact test).
// Only the accumulator differs: the reference adds 27 f32 numbers through an 8x8 shared-memory tree, per-block results on
// the host; here they are f64 sums (more accurate than the reference, deterministic, fixed order).
//
// MI355X-first design:
//  * ONE launch per frame batch runs the whole coarse-to-fine damped LM (3 levels x <= 15 accepted steps): the reference
//    pays a launch + device synchronisation + D2H copy per pass, and two passes per trial.
//  * ONE fused sweep per trial: chi2 at the trial pose and H,b at the same pose come from the same per-pixel terms, so the
//    reference's chi2(T_new) pass and the jacobianReduction(T) pass that follows an accepted step are one read of the
//    32 B/px (cloud 16, previous intensity 4, current image + two gradient images 12); after a rejected step the old H,b
//    are re-damped without touching memory.
//  * a stream's pixels are shared by NWG workgroups (NWG * streams ~ the chip's resident workgroups); partial sums meet at
//    the stream's leader workgroup through write-through (sc1) 8-byte words + one arrival counter, the leader adds them in
//    a fixed order, takes the LM step on one lane and publishes the next pose the same way: no grid-wide barrier, no fence
//    that would flush an XCD's L2 (MI355X_MICROARCH.md, "inter-workgroup visibility").  With >= 1 stream per resident
//    workgroup slot NWG is 1 and nothing leaves the workgroup.
//  * divisions: one IEEE reciprocal per denominator + a 4-instruction correctly rounded quotient (Markstein) instead of
//    twelve ~10-instruction IEEE divisions; the f64-promoted entries J0, J1 reduce to f32 quotients exactly (double
//    rounding through f64 is innocuous for a quotient of two f32 numbers, and a 2^-52 perturbation cannot reach an f32
//    rounding boundary, which a quotient of two 24-bit numbers misses by >= 2^-49 relative).
#include "common.h"
#include <algorithm>

namespace {

constexpr int NS = 28;                 // 21 H (packed upper by column) + 6 b + chi2; n_valid rides along as the 29th value
constexpr int FULL_THREADS = 256;
constexpr int FULL_WAVES = FULL_THREADS / 64;
#ifndef SVS_FULL_MINW
#define SVS_FULL_MINW 4
#endif
constexpr int FULL_MINW = SVS_FULL_MINW;   // waves per SIMD the tracker's register allocation must allow (= workgroups per CU): the sweep is
                                           // gather-latency bound, the one-lane LM step may spill

struct FullLevel {
  const float4 *cloud; const float *prev, *cur, *dx, *dy;
  int w, h, s4, fs;
  float f, cx, cy;
};
struct M34 { float m[12]; };           // GpuMatrix34: column-major 3x4 (gpu/dense_tracking.cuh:219-237)
struct M44 { float m[16]; };

// ---- exact f32 quotients sharing a denominator ------------------------------------------------------------------
// r = RN(1 / z) (IEEE division); q = RN(a r); e = a - q z exactly (FMA); RN(q + e r) = RN(a / z) (Markstein: r correctly
// rounded, q faithful); q's sign bit is OR-ed in so that a = +-0 keeps the IEEE sign of the zero quotient (for a != 0 the two
// signs agree anyway).  4 instructions instead of ~10.  Valid while nothing over- or underflows on the way: the callers use
// it only for 1e-9 < |z| < 1e9 (then every numerator of an in-frame pixel is far inside the f32 range) and take the IEEE
// path otherwise.
struct Recip { float z, r; };
__device__ __forceinline__ Recip recip(float z) { Recip R; R.z = z; R.r = 1.0f / z; return R; }
template <bool FAST> __device__ __forceinline__ float quot(float a, const Recip &R) {
  if (!FAST) return a / R.z;
  const float q = a * R.r;
  const float e = __builtin_fmaf(-q, R.z, a);
  const float q2 = __builtin_fmaf(e, R.r, q);
  return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, q) & 0x80000000u) | __builtin_bit_cast(unsigned, q2));
}
__device__ __forceinline__ bool z_is_tame(float z) { const float az = fabsf(z); return az > 1e-9f && az < 1e9f; }

// texture fetch of the reference: tex2D(tex, uv.x + 0.5f, uv.y + 0.5f), linear filter, clamp addressing
// (gpu/dense_tracking.cu:206-215).  (xt, yt) = uv + 0.5f as the reference forms it; the unit samples at (xt, yt) - 0.5.
struct TexPos { int i0, i1, j0, j1; float w00, w01, w10, w11; };
__device__ __forceinline__ TexPos tex_pos(float xt, float yt, int w, int h) {
  TexPos P;
  const float xb = xt - 0.5f, yb = yt - 0.5f;
  const float fi = floorf(xb), fj = floorf(yb);
  const float sx = xb - fi, sy = yb - fj;
  const float wx0 = 1 - sx, wx1 = sx, wy0 = 1 - sy, wy1 = sy;
  const int i = (int)fi, j = (int)fj;
  P.i0 = min(max(i, 0), w - 1); P.i1 = min(max(i + 1, 0), w - 1);
  P.j0 = min(max(j, 0), h - 1); P.j1 = min(max(j + 1, 0), h - 1);
  P.w00 = wx0 * wy0; P.w01 = wx0 * wy1; P.w10 = wx1 * wy0; P.w11 = wx1 * wy1;     // (x,y) (x,y+1) (x+1,y) (x+1,y+1)
  return P;
}
struct Taps { float v00, v01, v10, v11; };
// uniform base (SGPR pair) + 32-bit unsigned byte offset per lane: one address register per tap instead of a 64-bit add each
template <class T> __device__ __forceinline__ T ld_at(const T *base, unsigned byte_off) { return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off); }
struct TapOffs { unsigned o00, o01, o10, o11; };
__device__ __forceinline__ TapOffs tap_offsets(const TexPos &P, int stride) {
  TapOffs o;
  const unsigned r0 = (unsigned)(P.j0 * stride), r1 = (unsigned)(P.j1 * stride);
  o.o00 = (r0 + (unsigned)P.i0) * 4u; o.o01 = (r1 + (unsigned)P.i0) * 4u; o.o10 = (r0 + (unsigned)P.i1) * 4u; o.o11 = (r1 + (unsigned)P.i1) * 4u;
  return o;
}
__device__ __forceinline__ Taps tex_load(const float *__restrict__ m, const TapOffs &o) {
  Taps t;
  t.v00 = ld_at(m, o.o00); t.v01 = ld_at(m, o.o01); t.v10 = ld_at(m, o.o10); t.v11 = ld_at(m, o.o11);
  return t;
}
__device__ __forceinline__ float tex_mix(const TexPos &P, const Taps &t) { return P.w00 * t.v00 + P.w01 * t.v01 + P.w10 * t.v10 + P.w11 * t.v11; }

// One thread of jacobianReduction_kernel / chi2_kernel up to its reduction (gpu/dense_tracking.cu:193-221, :395-411), in two
// stages so that a sweep can issue the taps of the next pixel before it does the arithmetic of this one.
// FUSE: dx / dy images are not read; their taps are formed from the current image as the reference's derivative filter
// defines them (I(x+1) - I(x-1), I(y+1) - I(y-1), BORDER_REPLICATE: frame_grabber.cpp:102-115) -- bit-identical values.
template <bool JAC, bool FUSE> struct PixelA {
  bool ok;
  float x, y, z, ip;
  Recip Rz;
  TexPos P;
  Taps c, gx, gy;                     // raw taps: current image, dx, dy  (FUSE: gx / gy hold the tap DIFFERENCES' operands)
  Taps gxm, gym;                      // FUSE only: the "minus" neighbours (gx / gy hold the "plus" ones)
};
// stage A: transform, project, in-frame test, tap loads issued (not waited for)
template <bool JAC, bool FUSE>
__device__ __forceinline__ void pixel_stage_a(const FullLevel &L, const M34 &T, const float4 p, const float ip, PixelA<JAC, FUSE> &A) {
  const float x = p.x * T.m[0] + p.y * T.m[3] + p.z * T.m[6] + p.w * T.m[9];       // matTimesVec / dotStride3
  const float y = p.x * T.m[1] + p.y * T.m[4] + p.z * T.m[7] + p.w * T.m[10];
  const float z = p.x * T.m[2] + p.y * T.m[5] + p.z * T.m[8] + p.w * T.m[11];
  A.x = x; A.y = y; A.z = z; A.ip = ip;
  A.Rz = recip(z);
  float uu, vv;
  if (__builtin_expect(z_is_tame(z), 1)) { uu = quot<true>(L.f * x, A.Rz) + L.cx; vv = quot<true>(L.f * y, A.Rz) + L.cy; }     // cameraProject
  else { uu = L.f * x / z + L.cx; vv = L.f * y / z + L.cy; }
  A.ok = (p.w > 0) && (uu >= 1.f && vv >= 1.f && uu <= (float)(L.w - 2) && vv <= (float)(L.h - 2));
  // a pixel that does not contribute becomes a harmless one: tap position (1, 1), point (0, 0, 1); stage B zeroes its residual
  // and gradients, so its Jacobian is +-0 and nothing has to be masked term by term
  if (!A.ok) { uu = 1.f; vv = 1.f; A.x = 0.f; A.y = 0.f; A.z = 1.f; A.Rz.z = 1.f; A.Rz.r = 1.f; }
  A.P = tex_pos(uu + 0.5f, vv + 0.5f, L.w, L.h);
  const TapOffs O = tap_offsets(A.P, L.fs);
  A.c = tex_load(L.cur, O);
  if (JAC) {
    if (FUSE) {
      // columns i0-1 .. i1+1 and rows j0-1 .. j1+1 of the current image, clamped (REPLICATE)
      const TexPos &P = A.P;
      const int xm0 = max(P.i0 - 1, 0), xp0 = min(P.i0 + 1, L.w - 1), xm1 = max(P.i1 - 1, 0), xp1 = min(P.i1 + 1, L.w - 1);
      const int ym0 = max(P.j0 - 1, 0), yp0 = min(P.j0 + 1, L.h - 1), ym1 = max(P.j1 - 1, 0), yp1 = min(P.j1 + 1, L.h - 1);
      const float *r0 = L.cur + (size_t)P.j0 * L.fs, *r1 = L.cur + (size_t)P.j1 * L.fs;
      A.gx.v00 = r0[xp0]; A.gxm.v00 = r0[xm0]; A.gx.v01 = r1[xp0]; A.gxm.v01 = r1[xm0];
      A.gx.v10 = r0[xp1]; A.gxm.v10 = r0[xm1]; A.gx.v11 = r1[xp1]; A.gxm.v11 = r1[xm1];
      const float *c0m = L.cur + (size_t)ym0 * L.fs, *c0p = L.cur + (size_t)yp0 * L.fs, *c1m = L.cur + (size_t)ym1 * L.fs, *c1p = L.cur + (size_t)yp1 * L.fs;
      A.gy.v00 = c0p[P.i0]; A.gym.v00 = c0m[P.i0]; A.gy.v01 = c1p[P.i0]; A.gym.v01 = c1m[P.i0];
      A.gy.v10 = c0p[P.i1]; A.gym.v10 = c0m[P.i1]; A.gy.v11 = c1p[P.i1]; A.gym.v11 = c1m[P.i1];
    } else {
      A.gx = tex_load(L.dx, O);
      A.gy = tex_load(L.dy, O);
    }
  }
}
// frameJacobian (gpu/dense_tracking.cu:65-80); the f64-promoted J0, J1 are exactly the f32 quotients (file header)
template <bool FAST>
__device__ __forceinline__ void frame_jacobian(float x, float y, float z, const Recip &Rz, float f, float gx, float gy, float *J) {
  const float zsq = z * z;
  const Recip Rq = recip(zsq);
  gx *= f; gy *= f;
  J[0] = -quot<FAST>(gx, Rz);                                   // (float)(-dx * (1. / p.z))
  J[1] = -quot<FAST>(gy, Rz);                                   // (float)(-dy * 1. / p.z)
  J[2] = quot<FAST>(gx * x, Rq) + quot<FAST>(gy * y, Rq);
  J[3] = quot<FAST>(gx * (x * y), Rq) + gy * (1.f + quot<FAST>(y * y, Rq));
  J[4] = -gx * (1.f + quot<FAST>(x * x, Rq)) - quot<FAST>(gy * (x * y), Rq);
  J[5] = quot<FAST>(gx * y, Rz) - quot<FAST>(gy * x, Rz);
}
// stage B: residual and Jacobian from the landed taps.  Returns whether the pixel contributes.
template <bool JAC, bool FUSE>
__device__ __forceinline__ bool pixel_stage_b(const FullLevel &L, const PixelA<JAC, FUSE> &A, float &res, float *J) {
  const float ic = tex_mix(A.P, A.c);
  res = A.ok ? A.ip - ic : 0.f;
  if (JAC) {
    float gx, gy;
    if (FUSE) {
      Taps dx, dy;
      dx.v00 = A.gx.v00 - A.gxm.v00; dx.v01 = A.gx.v01 - A.gxm.v01; dx.v10 = A.gx.v10 - A.gxm.v10; dx.v11 = A.gx.v11 - A.gxm.v11;
      dy.v00 = A.gy.v00 - A.gym.v00; dy.v01 = A.gy.v01 - A.gym.v01; dy.v10 = A.gy.v10 - A.gym.v10; dy.v11 = A.gy.v11 - A.gym.v11;
      gx = 0.5f * tex_mix(A.P, dx);
      gy = 0.5f * tex_mix(A.P, dy);
    } else {
      gx = 0.5f * tex_mix(A.P, A.gx);
      gy = 0.5f * tex_mix(A.P, A.gy);
    }
    if (!A.ok) { gx = 0.f; gy = 0.f; }
    if (__builtin_expect(z_is_tame(A.z), 1)) frame_jacobian<true>(A.x, A.y, A.z, A.Rz, L.f, gx, gy, J);
    else frame_jacobian<false>(A.x, A.y, A.z, A.Rz, L.f, gx, gy, J);
  }
  return A.ok;
}
template <bool JAC, bool FUSE>
__device__ __forceinline__ bool full_pixel(const FullLevel &L, const M34 &T, const float4 p, const float ip, float &res, float *J) {
  PixelA<JAC, FUSE> A;
  pixel_stage_a<JAC, FUSE>(L, T, p, ip, A);
  return pixel_stage_b<JAC, FUSE>(L, A, res, J);
}

struct AccF {
  double v[NS];
  int n;
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < NS; ++i) v[i] = 0;
    n = 0;
  }
  // chi2 += (double)(res * res) as the restatement does; H and b take the exact products of the f32 terms (one f64 FMA each
  // instead of an f32 product, a conversion and an f64 add: the sweep is issue-bound on its f64 instructions)
  // (a pixel that does not contribute arrives with res = 0 and J = +-0, see pixel_stage_a / _b)
  template <bool JAC> __device__ __forceinline__ void add(bool ok, float res, const float *J) {
    const float r = res;
    v[27] += (double)(r * r);
    n += ok ? 1 : 0;
    if (JAC) {
      double Jd[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) Jd[i] = (double)J[i];
      int k = 0;
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int q = 0; q <= c; ++q) { v[k] = __builtin_fma(Jd[c], Jd[q], v[k]); ++k; }
      const double rd = (double)r;
#pragma unroll
      for (int i = 0; i < 6; ++i) v[21 + i] = __builtin_fma(Jd[i], rd, v[21 + i]);
    }
  }
};

// workgroup reduction of the 29 values (recursive halving inside a wave: 32 f64 exchanges instead of 29 x 6 butterflies);
// result in s_out[0..NS], valid for all threads after the trailing barrier
__device__ __forceinline__ void full_block_reduce(const AccF &a, double (*s_part)[NS + 1], double *s_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double x[32];
#pragma unroll
  for (int i = 0; i < NS; ++i) x[i] = a.v[i];
  x[NS] = (double)a.n;
#pragma unroll
  for (int i = NS + 1; i < 32; ++i) x[i] = 0.0;
#pragma unroll
  for (int step = 0; step < 5; ++step) {
    const int half = 16 >> step, bit = 1 << step;
    const bool up = (lane & bit) != 0;
#pragma unroll
    for (int k = 0; k < half; ++k) {
      const double send = up ? x[k] : x[k + half];
      const double keep = up ? x[k + half] : x[k];
      x[k] = keep + __shfl_xor(send, bit, 64);
    }
  }
  x[0] += __shfl_xor(x[0], 32, 64);
  const int idx = ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
  if (lane < 32 && idx <= NS) s_part[wave][idx] = x[0];
  __syncthreads();
  if (threadIdx.x <= NS) {
    double s = 0;
#pragma unroll
    for (int w = 0; w < FULL_WAVES; ++w) s += s_part[w][threadIdx.x];
    s_out[threadIdx.x] = s;
  }
  __syncthreads();
}

// one sweep of this workgroup's share of a level: pixel groups of FULL_THREADS consecutive pixels (row-major over the
// w x h image, so a wave reads 1 KiB of consecutive float4 cloud entries), groups dealt round-robin to the stream's
// workgroups.  The T-independent loads of the next pixel (cloud entry, previous intensity: the HBM stream) are issued before
// this pixel's arithmetic.  Measured (profiles/r2_notes.md): the sweep is VALU-bound (75 % VALU-busy at 4 waves per SIMD), a
// deeper software pipeline (taps of pixel k+1 in flight during the Jacobian of pixel k) costs 28 registers and buys nothing.
template <bool JAC, bool FUSE>
__device__ __forceinline__ void full_sweep(const FullLevel &L, const M34 &T, int wg, int nwg, AccF &a) {
  const int n = L.w * L.h;
  const int step = FULL_THREADS * nwg;
  const int su = step % L.w, sv = step / L.w;
  int i = wg * FULL_THREADS + (int)threadIdx.x;
  int u = i % L.w, v = i / L.w;
  const float4 none = make_float4(0.f, 0.f, 1.f, -1.f);
  bool in = i < n;
  float4 p = in ? ld_at(L.cloud, (unsigned)(v * L.s4 + u) * 16u) : none;
  float ip = in ? ld_at(L.prev, (unsigned)(v * L.fs + u) * 4u) : 0.f;
  for (; i < n; i += step) {
    const float4 pc = p;
    const float ipc = ip;
    u += su; v += sv;
    if (u >= L.w) { u -= L.w; ++v; }
    in = i + step < n;
    p = in ? ld_at(L.cloud, (unsigned)(v * L.s4 + u) * 16u) : none;
    ip = in ? ld_at(L.prev, (unsigned)(v * L.fs + u) * 4u) : 0.f;
    float res = 0.f, J[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool ok = full_pixel<JAC, FUSE>(L, T, pc, ipc, res, J);
    a.template add<JAC>(ok, res, J);
  }
}

// ---- 6x6 solve + SE3 exp (as in dense.hip; the LM step runs on one lane) -----------------------------------------
__device__ void f_solve6(const double *A, const double *b, double *x) {
  double M[6][7];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) M[i][j] = A[i * 6 + j];
    M[i][6] = b[i];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int p = k;
    double best = fabs(M[k][k]);
#pragma unroll
    for (int i = k + 1; i < 6; ++i) { const double v = fabs(M[i][k]); if (v > best) { best = v; p = i; } }
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const bool sw = p == i;
#pragma unroll
      for (int j = k; j < 7; ++j) { const double a = M[k][j], c = M[i][j]; M[k][j] = sw ? c : a; M[i][j] = sw ? a : c; }
    }
    const double piv = M[k][k];
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const double f = M[i][k] / piv;
#pragma unroll
      for (int j = k; j < 7; ++j) M[i][j] -= f * M[k][j];
    }
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s_ = M[i][6];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) s_ -= M[i][j] * x[j];
    x[i] = s_ / M[i][i];
  }
}
__device__ void f_se3_exp_mul(const double *x, const double *T, double *Tn) {   // Tn = exp(x) * T
  const double *w = x + 3;
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
  double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, W2[9], R[9], V[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
  double a, b;
  if (th < 1e-10) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; } else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
  for (int i = 0; i < 9; ++i) R[i] = a * W[i] + b * W2[i];
  R[0] += 1; R[4] += 1; R[8] += 1;
  if (th < 1e-10) { for (int i = 0; i < 9; ++i) V[i] = R[i]; }
  else {
    double c = (1.0 - cos(th)) / th2, d = (th - sin(th)) / (th2 * th);
    for (int i = 0; i < 9; ++i) V[i] = c * W[i] + d * W2[i];
    V[0] += 1; V[4] += 1; V[8] += 1;
  }
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = V[3 * i] * x[0] + V[3 * i + 1] * x[1] + V[3 * i + 2] * x[2];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) Tn[4 * i + j] = R[3 * i] * T[j] + R[3 * i + 1] * T[4 + j] + R[3 * i + 2] * T[8 + j];
    Tn[4 * i + 3] += t[i];
  }
}

// ---- cross-workgroup words: write-through stores / L1-bypassing loads, agent scope --------------------------------
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct TrackFullArgs {
  FullLevel lv[3];
  size_t cloud_b[3], f_b[3];
  double *T_jac;                       // optional [batch][3][12]
  svs_dense_lm_record *rec; int rec_cap;   // optional [batch][rec_cap]
  int32_t *n_rec;                      // optional [batch]
  // cross-workgroup hand-off area (nwg > 1): per stream [nwg][32] partial sums, 16 broadcast words, arrival counter, epoch
  double *part; double *bcast; unsigned *count; unsigned *epoch;
  int nwg;
};

enum { PH_INIT = 0, PH_TRIAL = 1 };
constexpr long SPIN_LIMIT = 1l << 22;          // x s_sleep(2): seconds -- a sibling workgroup that never became resident

// The whole DenseTracker::denseTrackingGpu loop of one stream (dense_tracking.cpp:60-193).
template <bool FUSE, bool MULTI>
__global__ __launch_bounds__(FULL_THREADS, FULL_MINW) void dense_track_full_kernel(TrackFullArgs A, double *__restrict__ T_io, int32_t *__restrict__ passes_out) {
  __shared__ double s_part[FULL_WAVES][NS + 1];
  __shared__ double s_out[NS + 1];
  __shared__ double s_sum[NS + 1];
  __shared__ double s_T[12], s_Teval[12], s_H[21], s_b[6], s_Tj[3][12];
  __shared__ int s_ctl[2];             // level (or -1 when finished), failure flag
  __shared__ double s_grp[8][32];      // leader: group sums of the other workgroups' partials
  const int nwg = MULTI ? A.nwg : 1;
  const int slot = MULTI ? blockIdx.x / nwg : blockIdx.x, wg = MULTI ? blockIdx.x % nwg : 0;
  const int tid = threadIdx.x;
  const bool leader = wg == 0;
  double *part = MULTI ? A.part + (size_t)slot * nwg * 32 : nullptr;
  double *bcast = MULTI ? A.bcast + (size_t)slot * 16 : nullptr;
  unsigned *count = MULTI ? A.count + slot : nullptr, *epoch = MULTI ? A.epoch + slot : nullptr;
  // LM state of the leader (used by lane 0 of workgroup 0 only; kept in LDS so that nothing of it is live across the sweeps)
  __shared__ struct { double mu, nu; float chi2; int trial, iter, phase, n_rec; } s_lm;
  if (tid == 0) { s_lm.mu = 0.01f; s_lm.nu = 2; s_lm.chi2 = 0.f; s_lm.trial = 0; s_lm.iter = 0; s_lm.phase = PH_INIT; s_lm.n_rec = 0; }   // `double mu = 0.01f` (dense_tracking.cpp:103)
  if (tid < 12) { const double t = T_io[(size_t)slot * 12 + tid]; s_T[tid] = t; s_Teval[tid] = t; }
  if (tid < 36) s_Tj[tid / 12][tid % 12] = T_io[(size_t)slot * 12 + tid % 12];
  if (tid == 0) { s_ctl[0] = 2; s_ctl[1] = 0; }
  __syncthreads();
  int passes = 0;
  for (;;) {
    const int level = s_ctl[0];
    if (level < 0) break;
    FullLevel L = A.lv[level];
    L.cloud += slot * A.cloud_b[level];
    L.prev += slot * A.f_b[level]; L.cur += slot * A.f_b[level];
    if (!FUSE) { L.dx += slot * A.f_b[level]; L.dy += slot * A.f_b[level]; }
    M34 T;                              // GpuMatrix34::set: f64 -> f32, column-major (dense_tracking.cpp:80-82)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r) T.m[3 * c + r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)s_Teval[4 * r + c])));   // uniform: keep in SGPRs
    AccF a;
    a.zero();
    full_sweep<true, FUSE>(L, T, wg, nwg, a);
    full_block_reduce(a, s_part, s_out);
    ++passes;
    bool failed = false;
    if (MULTI && !leader) {
      if (tid <= NS) st_agent(part + wg * 32 + tid, s_out[tid]);
      drain_stores();
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spin = 0;
        while (__hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)passes && ++spin < SPIN_LIMIT) __builtin_amdgcn_s_sleep(2);
        if (spin >= SPIN_LIMIT) s_ctl[1] = 1;
      }
      __syncthreads();
      if (s_ctl[1]) break;                                     // the leader never answered: give up (passes_out is the leader's)
      if (tid < 12) s_Teval[tid] = ld_agent(bcast + tid);
      if (tid == 12) { const double c = ld_agent(bcast + 12); s_ctl[0] = (int)c; }
      __syncthreads();
      continue;
    }
    // ---- leader: gather the stream's sums in a fixed order ----
    if (MULTI) {
      if (tid == 0) {
        const unsigned want = (unsigned)(nwg - 1) * (unsigned)passes;
        long spin = 0;
        while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spin < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
        if (spin >= SPIN_LIMIT) s_ctl[1] = 1;
      }
      __syncthreads();
      failed = s_ctl[1] != 0;
      // 8 lanes per value, each adds every 8th workgroup's partial (8 loads in flight), then the 8 group sums in order
      const int val = tid & 31, grp = tid >> 5;
      double acc = 0;
      if (!failed && val <= NS) {
        for (int w0 = 1 + grp; w0 < nwg; w0 += 64) {
          double t8[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { const int w = w0 + 8 * k; t8[k] = w < nwg ? ld_agent(part + w * 32 + val) : 0.0; }
#pragma unroll
          for (int k = 0; k < 8; ++k) acc += t8[k];
        }
      }
      __syncthreads();
      s_grp[grp][val] = acc;
      __syncthreads();
      if (tid <= NS) {
        double s = s_out[tid];
#pragma unroll
        for (int g = 0; g < 8; ++g) s += s_grp[g][tid];
        s_sum[tid] = s;
      }
    } else if (tid <= NS) s_sum[tid] = s_out[tid];
    __syncthreads();
    // ---- leader, lane 0: one step of the reference's LM state machine ----
    if (tid == 0) {
      float chi2 = s_lm.chi2;
      double mu = s_lm.mu, nu = s_lm.nu;
      int trial = s_lm.trial, iter = s_lm.iter, phase = s_lm.phase, n_rec = s_lm.n_rec;
      const float chi2_e = (float)s_sum[27];                      // GpuTracker::chi2 returns float
      int lvl = level;
      bool level_done = false;
      if (failed) { lvl = -1; }
      else if (phase == PH_INIT) {                                 // float chi2 = gpu_tracker_.chi2(...) (:84-92)
        chi2 = chi2_e;
        for (int k = 0; k < 21; ++k) s_H[k] = s_sum[k];
        for (int k = 0; k < 6; ++k) s_b[k] = s_sum[21 + k];
        mu = 0.01f; nu = 2; trial = 0; iter = 0;
        if (A.rec && n_rec < A.rec_cap) A.rec[(size_t)slot * A.rec_cap + n_rec] = svs_dense_lm_record{lvl, 2, chi2, chi2};
        ++n_rec;
        phase = PH_TRIAL;
      } else {
        const float new_chi2 = chi2_e;
        const double rho = chi2 - new_chi2;                       // float - float, widened (:142)
        if (A.rec && n_rec < A.rec_cap) A.rec[(size_t)slot * A.rec_cap + n_rec] = svs_dense_lm_record{lvl, rho > 0 ? 1 : 0, chi2, new_chi2};
        ++n_rec;
        bool stop = false;
        if (rho > 0) {
          double mx = 0;
          for (int k = 0; k < 6; ++k) mx = fmax(mx, fabs(s_b[k]));
          stop = mx <= 1e-10;                                     // norm_max(b) <= EPS, b of the pass the step came from (:148)
          for (int k = 0; k < 12; ++k) s_T[k] = s_Teval[k];
          chi2 = new_chi2;
          for (int k = 0; k < 21; ++k) s_H[k] = s_sum[k];         // = the next iteration's jacobianReduction at the accepted pose
          for (int k = 0; k < 6; ++k) s_b[k] = s_sum[21 + k];
          const double t = 2 * rho - 1;
          mu *= fmax(1. / 3., 1 - t * t * t);
          nu = 2.;
          trial = 0;
          ++iter;                                                 // leaves the do-while; next i of the for loop
        } else {
          mu *= nu;
          nu *= 2.;
          ++trial;
          if (trial == 2) stop = true;
        }
        level_done = stop || iter >= 15;
      }
      if (lvl >= 0 && level_done) {
        --lvl;
        phase = PH_INIT;
        for (int k = 0; k < 12; ++k) s_Teval[k] = s_T[k];         // next level starts with chi2 + H,b at the current pose
      } else if (lvl >= 0) {
        // jacobianReduction at s_T gave (s_H, s_b): damped solve and trial pose (:115-128)
        for (int k = 0; k < 12; ++k) s_Tj[lvl][k] = s_T[k];
        double H[36], nb[6], x[6], Tn[12];
        int k = 0;
        for (int c = 0; c < 6; ++c) for (int r = 0; r <= c; ++r) { H[6 * r + c] = s_H[k]; H[6 * c + r] = s_H[k]; ++k; }
        for (int q = 0; q < 6; ++q) H[7 * q] += mu * H[7 * q];
        for (int q = 0; q < 6; ++q) nb[q] = -s_b[q];
        f_solve6(H, nb, x);
        f_se3_exp_mul(x, s_T, Tn);
        for (int q = 0; q < 12; ++q) s_Teval[q] = Tn[q];
      }
      s_ctl[0] = lvl;
      s_lm.chi2 = chi2; s_lm.mu = mu; s_lm.nu = nu; s_lm.trial = trial; s_lm.iter = iter; s_lm.phase = phase; s_lm.n_rec = n_rec;
    }
    __syncthreads();
    if (MULTI) {
      if (tid < 12) st_agent(bcast + tid, s_Teval[tid]);
      if (tid == 12) st_agent(bcast + 12, (double)s_ctl[0]);
      drain_stores();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(epoch, (unsigned)passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!leader) return;
  if (tid < 12) T_io[(size_t)slot * 12 + tid] = s_T[tid];
  if (tid == 0) {
    if (passes_out) passes_out[slot] = s_ctl[1] ? -1 : passes;
    if (A.n_rec) A.n_rec[slot] = s_lm.n_rec;
  }
  if (A.T_jac && tid < 36) A.T_jac[(size_t)slot * 36 + tid] = s_Tj[tid / 12][tid % 12];
}

// ---- single passes behind the GpuTracker call surface -------------------------------------------------------------
template <bool JAC>
__global__ __launch_bounds__(FULL_THREADS) void dense_pass_full_kernel(FullLevel L, M34 T, double *__restrict__ partials) {
  __shared__ double s_part[FULL_WAVES][NS + 1];
  __shared__ double s_out[NS + 1];
  AccF a;
  a.zero();
  full_sweep<JAC, false>(L, T, blockIdx.x, gridDim.x, a);
  full_block_reduce(a, s_part, s_out);
  if (threadIdx.x <= NS) partials[(size_t)blockIdx.x * (NS + 1) + threadIdx.x] = s_out[threadIdx.x];
}
__global__ void dense_finalize_full_kernel(const double *__restrict__ partials, int nblocks, svs_dense_sums *__restrict__ out) {
  const int t = threadIdx.x;
  if (t > NS) return;
  double s = 0;
  for (int b = 0; b < nblocks; ++b) s += partials[(size_t)b * (NS + 1) + t];
  if (t < 21) out->H[t] = s;
  else if (t < 27) out->b[t - 21] = s;
  else if (t == 27) out->chi2 = s;
  else out->n_valid = (long long)s;
}

// residualImage_kernel (gpu/dense_tracking.cu:495-541)
__global__ __launch_bounds__(256) void residual_image_full_kernel(FullLevel L, M34 T, float *__restrict__ rimg) {
  const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (u >= L.w || v >= L.h) return;
  const float4 p = L.cloud[(size_t)v * L.s4 + u];
  float4 o = make_float4(0.f, 1.f, 0.f, 1.f);
  if (p.w > 0) {
    float res, J[6];
    o = make_float4(1.f, 0.f, 0.f, 1.f);
    if (full_pixel<false, false>(L, T, p, L.prev[(size_t)v * L.fs + u], res, J)) {
      float g = 1 - 50.f * res * res;
      if (g < 0.f) g = 0.f;
      o = make_float4(g, g, g, 1.f);
    }
  }
  reinterpret_cast<float4 *>(rimg)[(size_t)v * L.s4 + u] = o;
}

// parity probe: the per-pixel terms of jacobianReduction_kernel before its reduction: out[v][u] = {J0..J5, res, valid}
template <bool FUSE>
__global__ __launch_bounds__(256) void pixel_terms_full_kernel(FullLevel L, M34 T, float *__restrict__ out) {
  const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (u >= L.w || v >= L.h) return;
  float res = 0.f, J[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool ok = full_pixel<true, FUSE>(L, T, L.cloud[(size_t)v * L.s4 + u], L.prev[(size_t)v * L.fs + u], res, J);
  float *o = out + 8 * ((size_t)v * L.w + u);
#pragma unroll
  for (int i = 0; i < 6; ++i) o[i] = ok ? J[i] : 0.f;
  o[6] = ok ? res : 0.f;
  o[7] = ok ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void pointcloud_full_kernel(M44 TQ, const float *__restrict__ disp, int w, int h, int si, int so,
                                                              int factor, float *__restrict__ cloud) {
  const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (u >= w || v >= h) return;
  const float d = disp[(size_t)v * si + u * factor] * factor;     // row not scaled: .cu:97-98 quirk kept
  float4 o;
  if (d <= 0) o = make_float4(0.f, 0.f, 0.f, -1.f);
  else {
    const float q[4] = {(float)u, (float)v, d, 1.f};
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = q[0] * TQ.m[i] + q[1] * TQ.m[4 + i] + q[2] * TQ.m[8 + i] + q[3] * TQ.m[12 + i];
    o = make_float4(r[0] / r[3], r[1] / r[3], r[2] / r[3], 1.f);
  }
  reinterpret_cast<float4 *>(cloud)[(size_t)v * so + u] = o;
}

// computeDensePointCloudGpu with the pose on the device: TQ = [T^-1; 0 0 0 1] * Q in double, every product sum in ascending k (what
// Eigen's 4x4 product does and the oracle pins), narrowed to float (GpuMatrix4::set), then the kernel above per stream
__global__ __launch_bounds__(256) void pointcloud_full_pose_kernel(const double *__restrict__ Tarr, svs_cam cam, const float *__restrict__ disp, int w, int h, int si,
                                                                   size_t disp_b, int so, size_t cloud_b, int factor, float *__restrict__ cloud) {
  const int slot = blockIdx.z;
  const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (u >= w || v >= h) return;
  double T[12], Ti[16];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = Tarr[(size_t)slot * 12 + k];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Ti[4 * r + c] = T[4 * c + r];
#pragma unroll
  for (int r = 0; r < 3; ++r) Ti[4 * r + 3] = -(Ti[4 * r] * T[3] + Ti[4 * r + 1] * T[7] + Ti[4 * r + 2] * T[11]);
  Ti[12] = 0; Ti[13] = 0; Ti[14] = 0; Ti[15] = 1;
  const double Q[16] = {1, 0, 0, -cam.cx, 0, 1, 0, -cam.cy, 0, 0, 0, cam.f, 0, 0, 1.0 / cam.b, 0};      // stereo_camera.cpp:24-34
  float TQ[16];                                                                                          // column-major, like GpuMatrix4
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double s = Ti[4 * r] * Q[c];
#pragma unroll
      for (int k = 1; k < 4; ++k) s += Ti[4 * r + k] * Q[4 * k + c];
      TQ[4 * c + r] = (float)s;
    }
  const float d = disp[slot * disp_b + (size_t)v * si + u * factor] * factor;     // row not scaled: .cu:97-98 quirk kept
  float4 o;
  if (d <= 0) o = make_float4(0.f, 0.f, 0.f, -1.f);
  else {
    const float q[4] = {(float)u, (float)v, d, 1.f};
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = q[0] * TQ[i] + q[1] * TQ[4 + i] + q[2] * TQ[8 + i] + q[3] * TQ[12 + i];
    o = make_float4(r[0] / r[3], r[1] / r[3], r[2] / r[3], 1.f);
  }
  reinterpret_cast<float4 *>(cloud)[slot * cloud_b + (size_t)v * so + u] = o;
}

// ---- CUDA-build preprocessing (frame_grabber.cpp:291-313): f32 level 0, f32 pyrDown, REPLICATE derivatives ---------
__global__ __launch_bounds__(256) void convert_f32_kernel(const uint8_t *__restrict__ src, int w, int h, int ss, size_t s_b,
                                                          float *__restrict__ dst, int ds, size_t d_b) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (x >= w || y >= h) return;
  dst[b * d_b + (size_t)y * ds + x] = (float)src[b * s_b + (size_t)y * ss + x] * (float)(1. / 255.);
}
__device__ __forceinline__ int refl101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}
// cv::gpu::pyrDown on f32 (ASSUMED OpenCV 2.4.2 gpu semantics, see oracle/vision.c): columns first, then rows, taps in
// ascending order, REFLECT_101.  One lane per output pixel: 25 taps from L1/L2 (the source row set of neighbouring lanes
// overlaps 5/2-fold); the level images are 1.2 MB + 0.3 MB per frame, this is not where the time goes.
__global__ __launch_bounds__(256) void pyr_down_f32_kernel(const float *__restrict__ src, int w, int h, int ss, size_t s_b,
                                                           float *__restrict__ dst, int ow, int oh, int ds, size_t d_b) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (x >= ow || y >= oh) return;
  const float k0 = 0.0625f, k1 = 0.25f, k2 = 0.375f;
  const float *s = src + b * s_b;
  int xs[5], ys[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) { xs[t] = refl101(2 * x - 2 + t, w); ys[t] = refl101(2 * y - 2 + t, h); }
  float col[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    float a = k0 * s[(size_t)ys[0] * ss + xs[t]];
    a = a + k1 * s[(size_t)ys[1] * ss + xs[t]];
    a = a + k2 * s[(size_t)ys[2] * ss + xs[t]];
    a = a + k1 * s[(size_t)ys[3] * ss + xs[t]];
    a = a + k0 * s[(size_t)ys[4] * ss + xs[t]];
    col[t] = a;
  }
  float a = k0 * col[0];
  a = a + k1 * col[1]; a = a + k2 * col[2]; a = a + k1 * col[3]; a = a + k0 * col[4];
  dst[b * d_b + (size_t)y * ds + x] = a;
}
__global__ __launch_bounds__(256) void deriv_replicate_kernel(const float *__restrict__ img, int w, int h, int s, size_t i_b,
                                                              float *__restrict__ dx, float *__restrict__ dy) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (x >= w || y >= h) return;
  const float *im = img + b * i_b;
  const int xm = max(x - 1, 0), xp = min(x + 1, w - 1), ym = max(y - 1, 0), yp = min(y + 1, h - 1);
  dx[b * i_b + (size_t)y * s + x] = im[(size_t)y * s + xp] - im[(size_t)y * s + xm];
  dy[b * i_b + (size_t)y * s + x] = im[(size_t)yp * s + x] - im[(size_t)ym * s + x];
}

static FullLevel make_level(const float *cloud4, int w, int h, int s4, const float *prev, const float *cur, const float *dx, const float *dy,
                            int fs, float f, float cx, float cy) {
  FullLevel L;
  L.cloud = reinterpret_cast<const float4 *>(cloud4); L.prev = prev; L.cur = cur; L.dx = dx; L.dy = dy;
  L.w = w; L.h = h; L.s4 = s4; L.fs = fs; L.f = f; L.cx = cx; L.cy = cy;
  return L;
}

}  // namespace

extern "C" int svs_preprocess_gpu_sem(svs_ctx *ctx, const uint8_t *d_src, int w, int h, int sstride, size_t s_bstride,
                                      float *const *d_img, float *const *d_dx, float *const *d_dy, const int32_t *fstride,
                                      const size_t *f_bstride, int levels, int batch) {
  SVS_REQUIRE(ctx, ctx && d_src && d_img && d_dx && d_dy && fstride && f_bstride && w > 0 && h > 0 && levels >= 1 && levels <= 3 && batch >= 1);
  SVS_DEVICE(ctx);
  int lw = w, lh = h;
  for (int l = 0; l < levels; ++l) {
    SVS_REQUIRE(ctx, d_img[l] && d_dx[l] && d_dy[l] && fstride[l] >= lw);
    const dim3 grid(div_up(lw, 64), div_up(lh, 4), batch);
    if (l == 0)
      hipLaunchKernelGGL(convert_f32_kernel, grid, dim3(256), 0, ctx->stream, d_src, w, h, sstride, s_bstride, d_img[0], fstride[0], f_bstride[0]);
    else {
      const int pw = l == 1 ? w : (w + 1) / 2, ph = l == 1 ? h : (h + 1) / 2;
      hipLaunchKernelGGL(pyr_down_f32_kernel, grid, dim3(256), 0, ctx->stream, (const float *)d_img[l - 1], pw, ph, fstride[l - 1], f_bstride[l - 1],
                         d_img[l], lw, lh, fstride[l], f_bstride[l]);
    }
    SVS_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(deriv_replicate_kernel, grid, dim3(256), 0, ctx->stream, (const float *)d_img[l], lw, lh, fstride[l], f_bstride[l], d_dx[l], d_dy[l]);
    SVS_LAUNCH_CHECK(ctx);
    lw = (lw + 1) / 2; lh = (lh + 1) / 2;
  }
  return SVS_OK;
}

extern "C" int svs_dense_residual_image_full(svs_ctx *ctx, const float *d_cloud4, int w, int h, int stride_f4, const float *d_prev,
                                             const float *d_cur, int stride_f, float f, float cx, float cy, const float *h_T,
                                             float *d_res_img4) {
  SVS_REQUIRE(ctx, ctx && d_cloud4 && d_prev && d_cur && h_T && d_res_img4 && w > 0 && h > 0);
  SVS_DEVICE(ctx);
  M34 T;
  for (int i = 0; i < 12; ++i) T.m[i] = h_T[i];
  hipLaunchKernelGGL(residual_image_full_kernel, dim3(div_up(w, 64), div_up(h, 4)), dim3(256), 0, ctx->stream,
                     make_level(d_cloud4, w, h, stride_f4, d_prev, d_cur, nullptr, nullptr, stride_f, f, cx, cy), T, d_res_img4);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

extern "C" int svs_dense_pixel_terms_full(svs_ctx *ctx, const float *d_cloud4, int w, int h, int stride_f4, const float *d_prev,
                                          const float *d_cur, const float *d_dx, const float *d_dy, int stride_f, float f, float cx,
                                          float cy, const float *h_T, float *d_terms8) {
  SVS_REQUIRE(ctx, ctx && d_cloud4 && d_prev && d_cur && h_T && d_terms8 && w > 0 && h > 0 && (!d_dx == !d_dy));
  SVS_DEVICE(ctx);
  M34 T;
  for (int i = 0; i < 12; ++i) T.m[i] = h_T[i];
  const FullLevel L = make_level(d_cloud4, w, h, stride_f4, d_prev, d_cur, d_dx, d_dy, stride_f, f, cx, cy);
  if (d_dx) hipLaunchKernelGGL(pixel_terms_full_kernel<false>, dim3(div_up(w, 64), div_up(h, 4)), dim3(256), 0, ctx->stream, L, T, d_terms8);
  else hipLaunchKernelGGL(pixel_terms_full_kernel<true>, dim3(div_up(w, 64), div_up(h, 4)), dim3(256), 0, ctx->stream, L, T, d_terms8);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

extern "C" int svs_dense_pass_full(svs_ctx *ctx, const float *d_cloud4, int w, int h, int stride_f4, const float *d_prev,
                                   const float *d_cur, const float *d_dx, const float *d_dy, int stride_f, float f,
                                   float cx, float cy, const float *h_T, int do_jac, svs_dense_sums *d_out) {
  SVS_REQUIRE(ctx, ctx && d_cloud4 && d_prev && d_cur && h_T && d_out && w > 0 && h > 0);
  SVS_REQUIRE(ctx, !do_jac || (d_dx && d_dy));
  SVS_DEVICE(ctx);
  M34 T;
  for (int i = 0; i < 12; ++i) T.m[i] = h_T[i];
  const int nblocks = std::min(div_up(w * h, FULL_THREADS), 4 * ctx->n_cu);
  void *part = nullptr;
  const int rc = svs_ctx_scratch(ctx, (size_t)nblocks * (NS + 1) * sizeof(double), &part);
  if (rc) return rc;
  const FullLevel L = make_level(d_cloud4, w, h, stride_f4, d_prev, d_cur, d_dx, d_dy, stride_f, f, cx, cy);
  if (do_jac) hipLaunchKernelGGL(dense_pass_full_kernel<true>, dim3(nblocks), dim3(FULL_THREADS), 0, ctx->stream, L, T, (double *)part);
  else hipLaunchKernelGGL(dense_pass_full_kernel<false>, dim3(nblocks), dim3(FULL_THREADS), 0, ctx->stream, L, T, (double *)part);
  SVS_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(dense_finalize_full_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double *)part, nblocks, d_out);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

extern "C" int svs_pointcloud_full(svs_ctx *ctx, const float *h_TQ, const float *d_disp, int w, int h, int stride_in,
                                   int stride_out, int factor, float *d_cloud4) {
  SVS_REQUIRE(ctx, ctx && h_TQ && d_disp && d_cloud4 && w > 0 && h > 0 && factor >= 1);
  SVS_DEVICE(ctx);
  M44 TQ;
  for (int i = 0; i < 16; ++i) TQ.m[i] = h_TQ[i];
  hipLaunchKernelGGL(pointcloud_full_kernel, dim3(div_up(w, 64), div_up(h, 4)), dim3(256), 0, ctx->stream, TQ, d_disp, w, h,
                     stride_in, stride_out, factor, d_cloud4);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

extern "C" int svs_pointcloud_full_pose(svs_ctx *ctx, const double *d_T, const svs_cam *cam, const float *d_disp, int disp_stride, size_t disp_bstride, int w, int h,
                                        int stride_out, size_t cloud_bstride, int factor, float *d_cloud4, int batch) {
  SVS_REQUIRE(ctx, ctx && d_T && cam && d_disp && d_cloud4 && w > 0 && h > 0 && factor >= 1 && batch >= 1 && stride_out >= w);
  SVS_DEVICE(ctx);
  hipLaunchKernelGGL(pointcloud_full_pose_kernel, dim3(div_up(w, 64), div_up(h, 4), batch), dim3(256), 0, ctx->stream, d_T, *cam, d_disp, w, h, disp_stride,
                     disp_bstride, stride_out, cloud_bstride, factor, d_cloud4);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

extern "C" int svs_dense_track_full(svs_ctx *ctx, const svs_dense_track_full_args *a, double *d_T_io, int32_t *d_passes_out, int batch) {
  SVS_REQUIRE(ctx, ctx && a && d_T_io && batch >= 1);
  SVS_DEVICE(ctx);
  TrackFullArgs A;
  const bool fuse = a->d_dx[0] == nullptr;
  for (int l = 0; l < 3; ++l) {
    SVS_REQUIRE(ctx, a->d_cloud4[l] && a->d_prev[l] && a->d_cur[l] && a->w[l] >= 4 && a->h[l] >= 4);
    SVS_REQUIRE(ctx, fuse ? (!a->d_dx[l] && !a->d_dy[l]) : (a->d_dx[l] && a->d_dy[l]));
    SVS_REQUIRE(ctx, a->stride_f4[l] >= a->w[l] && a->stride_f[l] >= a->w[l]);
    A.lv[l] = make_level(a->d_cloud4[l], a->w[l], a->h[l], a->stride_f4[l], a->d_prev[l], a->d_cur[l], a->d_dx[l], a->d_dy[l], a->stride_f[l],
                         (float)a->f[l], (float)a->cx[l], (float)a->cy[l]);      // GpuIntrinsics::set (gpu/dense_tracking.cuh:30-38)
    A.cloud_b[l] = a->cloud_bstride[l]; A.f_b[l] = a->f_bstride[l];
  }
  A.T_jac = a->d_T_jac_out;
  A.rec = a->d_record_out; A.rec_cap = a->d_record_out ? a->record_cap : 0; A.n_rec = a->d_n_record_out;
  SVS_REQUIRE(ctx, !a->d_record_out || a->record_cap > 0);
  // workgroups per stream: fill the resident slots (FULL_MINW workgroups of 256 threads per CU).  Workgroup ids are stream-major and
  // dispatch is in id order, so on an oversubscribed device the resident set is a prefix of whole streams (+ one partial one that
  // completes as soon as an earlier stream retires): co-residency is wanted for speed, not needed for progress.
  const int slots = FULL_MINW * ctx->n_cu;
  int nwg = ctx->full_nwg ? ctx->full_nwg : std::max(1, std::min(slots / batch, 64));
  nwg = std::min(nwg, std::max(1, div_up(a->w[0] * a->h[0], FULL_THREADS)));
  A.nwg = nwg;
  A.part = nullptr; A.bcast = nullptr; A.count = nullptr; A.epoch = nullptr;
  if (nwg > 1) {
    void *scr = nullptr;
    const size_t n_part = (size_t)batch * nwg * 32, n_bc = (size_t)batch * 16;
    const int rc = svs_ctx_scratch(ctx, (n_part + n_bc + (size_t)batch) * sizeof(double), &scr);
    if (rc) return rc;
    A.part = static_cast<double *>(scr);
    A.bcast = A.part + n_part;
    A.count = reinterpret_cast<unsigned *>(A.bcast + n_bc);
    A.epoch = A.count + batch;
    SVS_HIP(ctx, hipMemsetAsync(A.count, 0, sizeof(double) * (size_t)batch, ctx->stream));
    SvsSpinScope gate(ctx, nwg * batch);      // workgroups of a stream wait for their leader: one such launch on the device at a time (common.h); small launches: priority lane
    if (gate.rc) return gate.rc;
    if (fuse) hipLaunchKernelGGL((dense_track_full_kernel<true, true>), dim3(nwg * batch), dim3(FULL_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out);
    else hipLaunchKernelGGL((dense_track_full_kernel<false, true>), dim3(nwg * batch), dim3(FULL_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out);
    SVS_LAUNCH_CHECK(ctx);
    if (int grc = gate.leave()) return grc;
  } else {
    if (fuse) hipLaunchKernelGGL((dense_track_full_kernel<true, false>), dim3(batch), dim3(FULL_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out);
    else hipLaunchKernelGGL((dense_track_full_kernel<false, false>), dim3(batch), dim3(FULL_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out);
  }
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
