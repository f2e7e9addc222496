// dense.hip -- dense (direct) SE3 tracking for gfx950.
// Replaces DenseTracker::denseTrackingCpu / computeDensePointCloudCpu
// (dense_tracking.cpp:222-423; the parity target, SURVEY.md section 0 last row).  The CUDA build's
// full-resolution tracker (GpuTracker, denseTrackingGpu) lives in dense_full.hip.
//
// MI355X-first design:
//  * per-sample work is a gather (cloud, 4-tap bilinear of cur/dx/dy) + 27 accumulators; the
//    reduction is wave64 shuffles -> LDS across waves -> one partial per workgroup, summed by a
//    fixed-order finalize kernel (deterministic, no float atomics, no host-side summation of
//    block partials as in the reference's .cu:343-355);
//  * svs_dense_track_cpu_sem keeps the WHOLE coarse-to-fine LM loop on the device: one 1024-lane
//    workgroup per camera stream iterates chi2 / H,b / 6x6 solve / SE3 exp / accept-reject with
//    LDS broadcasts, so a frame costs one launch instead of ~90 launch+sync+D2H round trips.
#include "common.h"
#include "seqsum.h"
#include <algorithm>

#ifndef SVS_TRK_LAZY
#define SVS_TRK_LAZY 1      // (0: build experiments only -- the tracker kernel of rounds 1-4, accept test on the f64 sums alone)
#endif

namespace {

constexpr int NSUM = 28;   // 21 H + 6 b + chi2 ; n_valid kept separately

struct Acc {
  double v[NSUM];
  long long n;
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < NSUM; ++i) v[i] = 0;
    n = 0;
  }
};

// (P: `const float *`, or the same pointer with its address space spelled out -- see LevelArgsG)
template <class P>
__device__ __forceinline__ float interp32f(P m, int stride, float u, float v) {
  float x = floorf(u), y = floorf(v);
  float sx = u - x, sy = v - y;
  float wx0 = 1 - sx, wx1 = sx, wy0 = 1 - sy, wy1 = sy;
  int xi = (int)x, yi = (int)y;
  const P p = m + (unsigned)(yi * stride + xi);      // (in-frame samples: non-negative, below 2^31)
  float v00 = p[0], v10 = p[1], v01 = p[stride], v11 = p[stride + 1];
  return (wx0 * wy0) * v00 + (wx0 * wy1) * v01 + (wx1 * wy0) * v10 + (wx1 * wy1) * v11;
}

struct LevelArgs {
  const float *cloud; const uint8_t *prev; const float *cur, *dx, *dy;
  int pstride, fstride;
  svs_cam cam;
  const uint8_t *cur8;   // optional: current u8 level image; f32 image and Sobel taps are then formed on the fly
  int c8stride;
};

// The same with the address space of every pointer spelled out (global memory).  The tracker's sweep is a real call (track_pass_call): its operands come out of
// LDS, and a pointer that has been through memory is a flat pointer to the compiler -- every access a flat_load that also probes the LDS aperture and counts on
// both wait counters.
#define SVS_AS1 __attribute__((address_space(1)))
typedef float svs_f4 __attribute__((ext_vector_type(4)));
struct LevelArgsG {
  const SVS_AS1 float *cloud; const SVS_AS1 uint8_t *prev; const SVS_AS1 float *cur, *dx, *dy;
  int pstride, fstride;
  svs_cam cam;
  const SVS_AS1 uint8_t *cur8;
  int c8stride;
};
__device__ __forceinline__ float4 load_f4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 load_f4(const SVS_AS1 float *p) { const svs_f4 v = *(const SVS_AS1 svs_f4 *)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) { uint32_t w; __builtin_memcpy(&w, p, 4); return w; }
__device__ __forceinline__ uint32_t load_u32_unaligned(const SVS_AS1 uint8_t *p) {
  typedef uint32_t u32_unaligned __attribute__((aligned(1)));
  return *(const SVS_AS1 u32_unaligned *)p;
}

// Bilinear taps of I, dx, dy straight from the u8 level image.  (float)u8 * float(1/255.) and the
// centred differences are exactly the values frame_grabber.cpp:315-333 would have materialised, so
// the results are bit-identical to reading the f32 pyramids -- at 1/8 of the HBM traffic (the f32
// path touches ~half of three 4 B/px images per pass for a 1/16 sampling grid).  In-frame samples
// (border 2) never need the REFLECT_101 border rule.
template <class P>
__device__ __forceinline__ void taps_u8(P img, int stride, float u, float v, float &ic, float &gx, float &gy) {
  const float sc = (float)(1. / 255.);
  const float x = floorf(u), y = floorf(v);
  const float sx = u - x, sy = v - y;
  const float wx0 = 1 - sx, wx1 = sx, wy0 = 1 - sy, wy1 = sy;
  const int xi = (int)x, yi = (int)y;
  float f[4][4];
  // 32-bit offsets from the stream's (scalar) image pointer: the tap loads take the base from scalar registers and the sweep spends no 64-bit multiply-adds
  // (quarter rate) on addresses.  In-frame samples (border 2) keep yi - 1 >= 1 and xi - 1 >= 1: the offsets are non-negative and far below 2^31
  const unsigned o0 = (unsigned)((yi - 1) * stride + (xi - 1));
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t w4 = load_u32_unaligned(img + (o0 + (unsigned)(r * stride)));
#pragma unroll
    for (int c = 0; c < 4; ++c) f[r][c] = (float)((w4 >> (8 * c)) & 0xff) * sc;
  }
  const float w00 = wx0 * wy0, w01 = wx0 * wy1, w10 = wx1 * wy0, w11 = wx1 * wy1;   // v00=(x,y) v01=(x,y+1) v10=(x+1,y) v11
  ic = w00 * f[1][1] + w01 * f[2][1] + w10 * f[1][2] + w11 * f[2][2];
  gx = w00 * (f[1][2] - f[1][0]) + w01 * (f[2][2] - f[2][0]) + w10 * (f[1][3] - f[1][1]) + w11 * (f[2][3] - f[2][1]);
  gy = w00 * (f[2][1] - f[0][1]) + w01 * (f[3][1] - f[1][1]) + w10 * (f[2][2] - f[0][2]) + w11 * (f[3][2] - f[1][2]);
}

// loop body of dense_tracking.cpp:229-261 (chi2) / :278-331 (H, b), one sample.
// Branch-free (invalid samples are predicated to zero contributions and a safe address) so that the
// unrolled caller can keep the cloud load and the 12 bilinear taps of several samples in flight:
// the pass is bound by gather latency, not by bytes.
// the two T-independent loads of a sample (stored 3D point, previous-frame intensity): issued one trip ahead by track_pass
struct SampleIn { float4 c4; uint8_t prev; };
template <class LA>
__device__ __forceinline__ SampleIn sample_load(const LA &L, int u, int v, int cw, bool in_range) {
  SampleIn s;
  s.c4 = in_range ? load_f4(L.cloud + 4u * (unsigned)(v * cw + u)) : make_float4(0.f, 0.f, 1.f, -1.f);
  s.prev = L.prev[(unsigned)(((in_range ? v : 0) * 4) * L.pstride + (in_range ? u : 0) * 4)];
  return s;
}
// x / z and y / z share their denominator, and the Jacobian needs 1 / z again: one refined reciprocal serves all three.
// r = RN(1 / b) after two Newton steps on v_rcp_f64 (what the compiler's own division expansion does per quotient), then
// q = a r, e = a - q b (exact in an FMA), q + e r is the correctly rounded a / b by Markstein's theorem whenever r is the
// correctly rounded reciprocal -- 3 f64 instructions per quotient instead of 11.  The pass is f64-issue bound
// (~200 VALU instructions per sample, profiles/r1_notes.md), so this is ~10 % of its time.  Non-finite / zero b give
// NaN or inf exactly where the IEEE quotient would be inf or NaN: those samples fail the |uv| < 1e9 gate either way.
__device__ __forceinline__ double rcp_rn(double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(e, r, r);
  e = __builtin_fma(-b, r, 1.0);
  return __builtin_fma(e, r, r);
}
__device__ __forceinline__ double div_rn(double a, double b, double r) {
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-q, b, a), r, q);
}

// TM (term mode): what happens to the sample's float term res * res of the reference's `chi2 += res * res` -- 0: nothing; 1: stored at t_out (if not null);
// 2: stored past the caches (sibling workgroups of the stream will read it: MULTI)
template <bool JAC, bool U8SRC = false, int TM = 0, class LA = LevelArgs>
__device__ __forceinline__ void sample_cpu_sem(const LA &L, const double *T, const SampleIn &in, bool in_range, Acc &a,
                                               const float *ip_lut = nullptr, float *term_out = nullptr) {      // term_out (TM != 0): where the caller wants the sample's term
  const float4 c4 = in.c4;
  bool ok = in_range && (c4.w > 0);
  const double xp0 = c4.x, xp1 = c4.y, xp2 = c4.z;
  const double x = T[0] * xp0 + T[1] * xp1 + T[2] * xp2 + T[3];
  const double y = T[4] * xp0 + T[5] * xp1 + T[6] * xp2 + T[7];
  const double z = T[8] * xp0 + T[9] * xp1 + T[10] * xp2 + T[11];
  const double rz = rcp_rn(z);
  float uvx = (float)(L.cam.f * div_rn(x, z, rz) + L.cam.cx);
  float uvy = (float)(L.cam.f * div_rn(y, z, rz) + L.cam.cy);
  ok = ok && (fabsf(uvx) < 1e9f && fabsf(uvy) < 1e9f);
  const int ui = ok ? (int)uvx : 0, vi = ok ? (int)uvy : 0;
  ok = ok && (ui >= 2 && vi >= 2 && ui < L.cam.w - 2 && vi < L.cam.h - 2);
  if (!ok) { uvx = 2.f; uvy = 2.f; }                       // safe tap position, contribution masked below
  // (float)((1./255.) * u8): a double product rounded once (dense_tracking.cpp:290-293), NOT the f32 product convertTo
  // uses for the current image -- the tracker kernel keeps the 256 possible values in LDS (3 f64-rate instructions less)
  const float ip = ip_lut ? ip_lut[in.prev] : (float)((1. / 255.) * in.prev);
  float ic, g8x = 0.f, g8y = 0.f;
  if (U8SRC) taps_u8(L.cur8, L.c8stride, uvx, uvy, ic, g8x, g8y);
  else ic = interp32f(L.cur, L.fstride, uvx, uvy);
  float res = ip - ic;
  // dense_tracking.cpp:299-300 `if (res > 0.1) res = 0.1; if (res < -0.1) res = -0.1;` on a float res: the comparisons are made in double against the double 0.1,
  // which lies between 0.1f and the float below it, so `res > 0.1` is `res >= 0.1f` and the assignment stores 0.1f -- min / max with +-0.1f, bit for bit
  // (res is never NaN: both images are finite), two instructions instead of two conversions and two f64 compares
  res = fminf(fmaxf(res, -0.1f), 0.1f);
  if (!ok) res = 0.f;
  a.v[27] += (double)(res * res);
  // the term of the reference's `float chi2 += res*res` (0 where the reference skips the sample: x + 0 is exact)
  if constexpr (TM != 0) *term_out = res * res;
  a.n += ok ? 1 : 0;
  if (JAC) {
    const float gx = ok ? (float)(0.5 * (U8SRC ? g8x : interp32f(L.dx, L.fstride, uvx, uvy))) : 0.f;
    const float gy = ok ? (float)(0.5 * (U8SRC ? g8y : interp32f(L.dy, L.fstride, uvx, uvy))) : 0.f;
    const double zs = ok ? z : 1.0, xs = ok ? x : 0.0, ys = ok ? y : 0.0;
    // transformations.h:117-139 frame_jac_xyz2uv.  One reciprocal instead of eight f64 divisions and
    // fused multiply-adds in the 27 accumulations: the pass is f64-issue bound on its CU, and H/b only
    // need 1e-9 relative agreement with the serial oracle (uv above stays division-exact because the
    // in-frame test and the tap addresses must match bit for bit).
    {
      // (every fused multiply-add written out: left to the compiler's contraction the two instantiations of this function -- f32 pyramids / u8 source -- fused
      // different pairs, and their sums, which the tests hold bit-identical, drifted apart with every unrelated change to the kernel)
      const double f = L.cam.f, iz = ok ? rz : 1.0, iz2 = iz * iz, fx = f * iz, xz = xs * iz2 * f, yz = ys * iz2 * f;
      const double r0[6] = {-fx, 0, xz, xz * ys, -__builtin_fma(xz, xs, f), ys * fx};
      const double r1[6] = {0, -fx, yz, __builtin_fma(yz, ys, f), -(yz * xs), -(xs * fx)};
      double J[6];
      J[0] = gx * r0[0];
      J[1] = gy * r1[1];
#pragma unroll
      for (int k = 2; k < 6; ++k) J[k] = __builtin_fma(gx, r0[k], gy * r1[k]);
      int k = 0;
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int r = 0; r <= c; ++r) { a.v[k] = __builtin_fma(J[r], J[c], a.v[k]); ++k; }
#pragma unroll
      for (int i = 0; i < 6; ++i) a.v[21 + i] = __builtin_fma(J[i], (double)res, a.v[21 + i]);
    }
  }
}

// block reduction; result valid in threads [0, NSUM] (index = value id).  Inside a wave the 29 sums are reduced by
// recursive halving: in step s a lane keeps one half of its partial vector and hands the other half to its partner
// (lane ^ 2^s), so 16+8+4+2+1(+1) = 32 f64 exchanges replace 29 x 6 butterflies -- the pass is short, this tail is not.
template <int NWAVES>
__device__ __forceinline__ void block_reduce(Acc &a, double (*s_part)[NSUM + 1], double *out_vals /* NSUM+1 in LDS */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double x[32];
#pragma unroll
  for (int i = 0; i < NSUM; ++i) x[i] = a.v[i];
  x[NSUM] = (double)a.n;
#pragma unroll
  for (int i = NSUM + 1; i < 32; ++i) x[i] = 0.0;
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
  // lanes 0..31 now hold the wave sum of value id = bit-reversed lane index (5 bits)
  const int idx = ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
  if (lane < 32 && idx <= NSUM) s_part[wave][idx] = x[0];
  __syncthreads();
  if (threadIdx.x <= NSUM) {
    double s = 0;
    for (int w = 0; w < NWAVES; ++w) s += s_part[w][threadIdx.x];
    out_vals[threadIdx.x] = s;
  }
  __syncthreads();
}

// ---- multi-workgroup single pass (GpuTracker-style API) --------------------------------------
template <bool JAC>
__global__ __launch_bounds__(256) void dense_pass_cpu_sem_kernel(LevelArgs L, size_t cloud_b, size_t prev_b, size_t f_b,
                                                                 const double *__restrict__ Tarr, double *__restrict__ partials) {
  __shared__ double s_part[4][NSUM + 1];
  __shared__ double s_out[NSUM + 1];
  const int slot = blockIdx.y;
  L.cloud += slot * cloud_b; L.prev += slot * prev_b; L.cur += slot * f_b; L.dx += slot * f_b; L.dy += slot * f_b;
  double T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = Tarr[(size_t)slot * 12 + i];
  const int cw = L.cam.w / 4, ch = L.cam.h / 4, n = cw * ch;
  Acc a;
  a.zero();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) sample_cpu_sem<JAC>(L, T, sample_load(L, i % cw, i / cw, cw, true), true, a);
  block_reduce<4>(a, s_part, s_out);
  if (threadIdx.x <= NSUM) partials[((size_t)slot * gridDim.x + blockIdx.x) * (NSUM + 1) + threadIdx.x] = s_out[threadIdx.x];
}

__global__ void dense_finalize_kernel(const double *__restrict__ partials, int nblocks, svs_dense_sums *__restrict__ out) {
  const int slot = blockIdx.x, t = threadIdx.x;
  if (t > NSUM) return;
  double s = 0;
  for (int b = 0; b < nblocks; ++b) s += partials[((size_t)slot * nblocks + b) * (NSUM + 1) + t];
  svs_dense_sums *o = &out[slot];
  if (t < 21) o->H[t] = s;
  else if (t < 27) o->b[t - 21] = s;
  else if (t == 27) o->chi2 = (double)(float)s;     // reference keeps chi2 in a float
  else o->n_valid = (long long)s;
}

// ---- device-resident LM (whole denseTrackingCpu) ---------------------------------------------
// 6x6 solve by Gaussian elimination with partial pivoting (stands in for Eigen's ldlt(), as in the oracle).  Fully unrolled
// with compile-time indices: the row exchange is a chain of predicated swaps, so the augmented matrix lives in registers
// (a dynamically indexed copy would sit in scratch memory, and this runs on one lane while the workgroup waits).
__device__ void d_solve6(const double *A, const double *b, double *x) {
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
__device__ void d_se3_exp_mul(const double *x, const double *T, double *Tn) {   // Tn = exp(x) * T
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

__device__ __forceinline__ double mo2_bcast(double v, int src_lane) {      // src_lane is a compile-time constant after unrolling
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
// A x = b for a symmetric positive definite 6x6 system on seven lanes of a wave: lane c < 6 holds column c of A in col[0..6), lane 6 holds b.
// Gauss-Jordan, the multipliers of a step broadcast from the pivot column's lane; no pivoting (the reference's ldlt() pivots, which an SPD matrix
// does not need).  Every lane returns x in x[0..6).
__device__ __forceinline__ void wave_solve6(const double (&col)[6], double (&x)[6]) {
  double a[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) a[r] = col[r];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double ip = 1.0 / mo2_bcast(a[k], k);
    const double ak = a[k] * ip;               // row k of this lane's column, scaled
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if (r == k) continue;
      const double m = mo2_bcast(a[r], k);     // element (r, k) of the pivot column
      a[r] -= m * ak;
    }
    a[k] = ak;
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) x[r] = mo2_bcast(a[r], 6);
}
// exp(x) * T with everything in registers (d_se3_exp_mul above is an out-of-line call on arrays in scratch memory)
__device__ __forceinline__ void mo2_exp_mul(const double (&x)[6], const double (&T)[12], double (&Tn)[12]) {
  const double w0 = x[3], w1 = x[4], w2 = x[5];
  const double th2 = w0 * w0 + w1 * w1 + w2 * w2, th = sqrt(th2);
  // a = sin(th) / th, b = c = (1 - cos th) / th^2, d = (th - sin th) / th^3.  An LM step rotates by far less than half a radian: there the three
  // are even power series in th^2 (9 terms: remainder < 1e-22), no square root, no division, no libm call on the chain of the trial
  double a, b, c, d;
  if (th2 < 0.25) {
    // Horner in th^2 with the coefficients 1 / (k + 1)! written out (a table would live in scratch memory)
#define SVS_IF(k) (1.0 / k)
    const double t = th2;
    a = 1.0 - t * (SVS_IF(6.) - t * (SVS_IF(120.) - t * (SVS_IF(5040.) - t * (SVS_IF(362880.) - t * (SVS_IF(39916800.) - t * (SVS_IF(6227020800.) - t * (SVS_IF(1307674368000.) -
        t * (SVS_IF(355687428096000.) - t * SVS_IF(121645100408832000.)))))))));
    b = SVS_IF(2.) - t * (SVS_IF(24.) - t * (SVS_IF(720.) - t * (SVS_IF(40320.) - t * (SVS_IF(3628800.) - t * (SVS_IF(479001600.) - t * (SVS_IF(87178291200.) -
        t * (SVS_IF(20922789888000.) - t * (SVS_IF(6402373705728000.) - t * SVS_IF(2432902008176640000.)))))))));
    d = SVS_IF(6.) - t * (SVS_IF(120.) - t * (SVS_IF(5040.) - t * (SVS_IF(362880.) - t * (SVS_IF(39916800.) - t * (SVS_IF(6227020800.) - t * (SVS_IF(1307674368000.) -
        t * (SVS_IF(355687428096000.) - t * (SVS_IF(121645100408832000.) - t * SVS_IF(51090942171709440000.)))))))));
#undef SVS_IF
    c = b;
  } else {
    double sn, cs;
    sincos(th, &sn, &cs);
    const double ith2 = 1.0 / th2;
    a = sn / th; b = (1.0 - cs) * ith2; c = b; d = (th - sn) * ith2 / th;
  }
  // W = hat(w), W2 = W * W
  const double W[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
  double W2[9], R[9], V[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) { R[i] = a * W[i] + b * W2[i]; V[i] = c * W[i] + d * W2[i]; }
  R[0] += 1; R[4] += 1; R[8] += 1;
  V[0] += 1; V[4] += 1; V[8] += 1;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double t = V[3 * i] * x[0] + V[3 * i + 1] * x[1] + V[3 * i + 2] * x[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) Tn[4 * i + j] = R[3 * i] * T[j] + R[3 * i + 1] * T[4 + j] + R[3 * i + 2] * T[8 + j];
    Tn[4 * i + 3] += t;
  }
}


struct TrackArgs {
  LevelArgs lv[3];
  size_t cloud_b[3], prev_b[3], f_b[3], c8_b[3];
  double *T_jac;     // optional [batch][3][12]: pose of the last H,b pass of each level (what residual_img[level] shows)
  svs_dense_lm_record *rec; int rec_cap; int32_t *n_rec;      // optional accept / reject record of the loop
};

#ifndef SVS_TRK_THREADS
#define SVS_TRK_THREADS 512
#endif
#ifndef SVS_TRK_UNROLL
#define SVS_TRK_UNROLL 1
#endif
#ifndef SVS_TRK_MINW_BIG
#define SVS_TRK_MINW_BIG 4      // waves per SIMD the big-batch instantiations are built for (4: 128 VGPRs, two 512-lane workgroups per CU)
#endif
#ifndef SVS_TRK_T_SCALAR
#define SVS_TRK_T_SCALAR 1
#endif
constexpr int TRK_THREADS = SVS_TRK_THREADS;
constexpr int TRK_MINW_BIG = SVS_TRK_MINW_BIG;
constexpr int TRK_UNROLL = SVS_TRK_UNROLL;

template <bool JAC, bool U8SRC, int TM = 0, class LA = LevelArgs>
__device__ __forceinline__ void track_pass(const LA &L, const double *T_in, double (*s_part)[NSUM + 1], double *s_out, const float *ip_lut,
                                           int first, int nwg, float *t_buf = nullptr) {      // first = wg * TRK_THREADS + tid; nwg workgroups share the sweep; t_buf: the pass's terms (TM), or null
  const int cw = L.cam.w / 4, ch = L.cam.h / 4, n = cw * ch;
  // the pose of a sweep is the same in every lane: kept in scalar registers (24 vector registers less over the whole sweep -- the kernel is built for 128)
  double T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i)
    T[i] = SVS_TRK_T_SCALAR ? __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(T_in[i])), __builtin_amdgcn_readfirstlane(__double2loint(T_in[i]))) : T_in[i];
  Acc a;
  a.zero();
  // TRK_UNROLL samples per lane per trip (independent gather chains in flight); the next trip's stored points and
  // previous-frame intensities (independent of T) are loaded before this trip's arithmetic, so only the bilinear
  // taps -- whose addresses depend on the projection -- are exposed.  The grid position of the prefetched sample is
  // carried incrementally (one division per pass instead of one per sample: the pass is VALU-issue bound).
  const int STEP = TRK_UNROLL * TRK_THREADS * nwg;
  const int su = STEP % cw, sv = STEP / cw;
  SampleIn nxt[TRK_UNROLL];
  int pu[TRK_UNROLL], pv[TRK_UNROLL];
#pragma unroll
  for (int q = 0; q < TRK_UNROLL; ++q) {
    const int j = first + q * TRK_THREADS * nwg;
    pu[q] = j % cw; pv[q] = j / cw;
    const bool in = pv[q] < ch;
    nxt[q] = sample_load(L, in ? pu[q] : 0, in ? pv[q] : 0, cw, in);
  }
  // TM: the term of a sample is stored one trip LATE, in front of the next trip's loads.  vmcnt counts loads and stores in one queue: a store issued at the end of a
  // trip would be the youngest entry when the next trip waits for its prefetched operands at the top of the loop -- a full write round trip exposed per trip
  // (measured: + 18 % on the sweep); issued here it retires under the projection arithmetic and the tap loads that follow.
  // (the buffer's address space spelled out: the pointer comes out of a two-entry array indexed by the LM loop, and a flat store would tie the loop's LDS waits to memory)
  typedef __attribute__((address_space(1))) float *gptr_t;
  static_assert(TM == 0 || TRK_UNROLL == 1, "the deferred term store keeps one term per lane");
  float pend_t = 0.f;
  gptr_t pend_p = (gptr_t)t_buf + first - STEP;
  bool pend = false;
  auto flush = [&]() {
    if constexpr (TM == 1) { if (pend) *pend_p = pend_t; }
    if constexpr (TM == 2) { if (pend) __hip_atomic_store(pend_p, pend_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  };
  for (int i = first; i < n; i += STEP) {
    if constexpr (TM != 0) { flush(); pend_p += STEP; pend = t_buf != nullptr; }
    SampleIn cur[TRK_UNROLL];
#pragma unroll
    for (int q = 0; q < TRK_UNROLL; ++q) {
      cur[q] = nxt[q];
      pu[q] += su; pv[q] += sv;
      if (pu[q] >= cw) { pu[q] -= cw; ++pv[q]; }
      const bool in = pv[q] < ch;
      nxt[q] = sample_load(L, in ? pu[q] : 0, in ? pv[q] : 0, cw, in);
    }
#pragma unroll
    for (int q = 0; q < TRK_UNROLL; ++q) {
      if constexpr (TM != 0) sample_cpu_sem<JAC, U8SRC, TM, LA>(L, T, cur[q], true, a, ip_lut, &pend_t);      // (i < n: in range)
      else sample_cpu_sem<JAC, U8SRC, 0, LA>(L, T, cur[q], i + q * TRK_THREADS * nwg < n, a, ip_lut);
    }
  }
  if constexpr (TM != 0) flush();
  block_reduce<TRK_THREADS / 64>(a, s_part, s_out);
}

// ---- the sweep as a CALL ---------------------------------------------------------------------------------------------------------------------------------
// The tracker kernel is built for 128 vector registers (two 512-lane workgroups per CU) and the sweep alone fills them: 28 f64 accumulators, the sample in
// flight, the prefetched next one.  Inlined into the LM loop it shared its register allocation with everything that loop keeps alive -- per-stream pointers, the
// records, the accept test -- and paid with spills inside the loop over the samples and reloads in every pass.  As a function of its own the sweep gets the whole
// budget: the caller leaves its operands in LDS (uniform: read back into scalar registers), keeps its own state across the call, and nothing of the LM loop is
// live inside.  Results in g_s_out as before.
struct PassArgs { LevelArgs L; double T[12]; float *t_buf; int wg, nwg; };
__shared__ PassArgs g_pa;
__shared__ double g_s_part[TRK_THREADS / 64][NSUM + 1];
__shared__ double g_s_out[NSUM + 1];
__shared__ float g_iplut[256];
__device__ __forceinline__ int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni_f64(double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); }
template <class T>
__device__ __forceinline__ T *uni_ptr(T *p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
template <bool U8SRC, int TM>
__device__ __noinline__ void track_pass_call() {
  LevelArgsG L;
  L.cloud = (const SVS_AS1 float *)uni_ptr(g_pa.L.cloud); L.prev = (const SVS_AS1 uint8_t *)uni_ptr(g_pa.L.prev);
  L.cur = (const SVS_AS1 float *)uni_ptr(g_pa.L.cur); L.dx = (const SVS_AS1 float *)uni_ptr(g_pa.L.dx); L.dy = (const SVS_AS1 float *)uni_ptr(g_pa.L.dy);
  L.cur8 = (const SVS_AS1 uint8_t *)uni_ptr(g_pa.L.cur8);
  L.pstride = uni_i32(g_pa.L.pstride); L.fstride = uni_i32(g_pa.L.fstride); L.c8stride = uni_i32(g_pa.L.c8stride);
  L.cam.f = uni_f64(g_pa.L.cam.f); L.cam.cx = uni_f64(g_pa.L.cam.cx); L.cam.cy = uni_f64(g_pa.L.cam.cy); L.cam.b = 0;
  L.cam.w = uni_i32(g_pa.L.cam.w); L.cam.h = uni_i32(g_pa.L.cam.h);
  const int nwg = uni_i32(g_pa.nwg), first = uni_i32(g_pa.wg) * TRK_THREADS + (int)threadIdx.x;
  track_pass<true, U8SRC, TM, LevelArgsG>(L, g_pa.T, g_s_part, g_s_out, g_iplut, first, nwg, uni_ptr(g_pa.t_buf));
}

// "trk_seq_chi2": the reference's `float chi2`, summed the way the reference sums it -- one f32 accumulator, samples in row-major order (dense_tracking.cpp:229-262,
// 341-367).  Near convergence chi2 - new_chi2 is below the rounding noise of these 19 200-term sums, so the accept test of the last LM steps of a level is decided by
// the summation order; with this option the loop takes exactly the reference's decisions (the default compares the f64 sums, narrowed).  Wave 0 walks the terms the
// pass left in t_buf: 64 loads at a time, then 64 dependent adds on values broadcast from the lanes in order.  Slow (~0.1 ms per sum) -- parity runs only.
// a pointer into GLOBAL memory, told to the compiler: exact_seq_sum_f32 is a real call (it must not take part in the register allocation of the tracker's sweeps), and
// behind a call boundary a plain `const float *` is a flat pointer -- every access a flat_load that also probes the LDS aperture
typedef const __attribute__((address_space(1))) float *svs_gptr_f32;
typedef const __attribute__((address_space(1))) svs_f4 *svs_gptr_f4;
template <bool COH, typename P>      // COH: the terms were stored past the caches by sibling workgroups (MULTI) and are read the same way
__device__ __forceinline__ float seq_term_load(P p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
template <bool COH = false, typename P = const float *>
__device__ __forceinline__ float seq_sum_f32(P t, int n) {
  float acc = 0.f;
  const int lane = threadIdx.x & 63;
  for (int base = 0; base < n; base += 64) {
    const float v = base + lane < n ? seq_term_load<COH, P>(t + base + lane) : 0.f;      // (written by this workgroup before a barrier: workgroup-scope visibility is enough)
#pragma unroll
    for (int l = 0; l < 64; ++l) acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
  }
  return acc;
}
// every lane of the workgroup gets the sequential sum (wave 0 computes it between two barriers)
__device__ __forceinline__ float seq_chi2_f32(const float *t, int n) {
  __shared__ float s_seq;
  __syncthreads();
  if (threadIdx.x < 64) { const float v = seq_sum_f32(t, n); if (threadIdx.x == 0) s_seq = v; }
  __syncthreads();
  return s_seq;
}

// ---- the same bits without the chain (seqsum.h): the default accept test ------------------------------------------------------------------------------
// The tracker decides on its f64 sums wherever their difference is outside the rigorous error bound of the reference's float sums; inside the bound (most trials of
// level 0: the bound is 1.1e-3 of chi2 at 19 200 samples, the last LM steps of a level change chi2 by less) it needs the reference's float sums themselves.
// exact_seq_sum_f32 forms one from the terms a pass left in its buffer, all lanes of the workgroup at work: lane k owns the k-th run of S consecutive terms
// (f64 sum + count -> workgroup scan -> is my run certainly inside one binade? -> its integer map), a wave-level segmented scan composes the maps of neighbouring
// safe runs, and wave 0 walks the result: one table look-up per group of safe runs, a readlane chain over the ~10 runs that may straddle a power of two (staged in
// LDS by their lanes).  ~400 dependent float adds instead of 19 200.  Any failed check -> the plain chain over all terms (never seen; counted by the tests' hook).
constexpr int SEQ_Q = TRK_THREADS >= 512 ? 10 : 13;  // a lane's run of terms lives in SEQ_Q float4 registers: runs of up to 40 terms, i.e. up to 20 480 terms per sum (a 640 x 512 image)
constexpr int SEQ_RUN_MAX = 4 * SEQ_Q;
constexpr int SEQ_STAGE_FLOATS = 48 * SEQ_RUN_MAX;   // the runs the walker adds the slow way (typically ~10)
struct SeqShared {
  SvsSeqMap pref[TRK_THREADS];                       // composition of the safe runs from the start of my group (wave-local) through me
  unsigned char eb[TRK_THREADS];                     // biased exponent of my run's binade (0: not safe)
  short slot[TRK_THREADS];                           // my place in `stage` (unsafe runs), -1: none (the walker reads global memory)
  unsigned long long unsafe[TRK_THREADS / 64];
  double wave_P[TRK_THREADS / 64];
  int wave_c[TRK_THREADS / 64], wave_u[TRK_THREADS / 64];
  float stage[SEQ_STAGE_FLOATS];
  float result;
  int fell_back;
};
// the terms [j0, j0 + 4 nq) of a pass as nq float4 (16-byte aligned: the buffers are, and runs are multiples of four terms long); reads up to SEQ_RUN_MAX terms
// past the last one the caller will use (the buffers are padded by that much)
template <bool COH>
__device__ __forceinline__ void seq_load_run(svs_gptr_f32 p, int nq, svs_f4 (&v)[SEQ_Q]) {
  if constexpr (COH) {
    // past the caches, as the sibling workgroups stored them (sc1 = agent scope on gfx950, what an agent-scope atomic load compiles to); all ten requests in
    // flight before the one wait -- ten atomic dword loads in a row would be forty round trips
    static_assert(SEQ_Q == 10 || SEQ_Q == 13, "the blocks below issue ten (+ three) loads");
    asm volatile("global_load_dwordx4 %0, %10, off sc1\n\t"
                 "global_load_dwordx4 %1, %10, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %10, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %10, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %10, off offset:64 sc1\n\t"
                 "global_load_dwordx4 %5, %10, off offset:80 sc1\n\t"
                 "global_load_dwordx4 %6, %10, off offset:96 sc1\n\t"
                 "global_load_dwordx4 %7, %10, off offset:112 sc1\n\t"
                 "global_load_dwordx4 %8, %10, off offset:128 sc1\n\t"
                 "global_load_dwordx4 %9, %10, off offset:144 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9])
                 : "v"(p)
                 : "memory");
    if constexpr (SEQ_Q == 13)
      asm volatile("global_load_dwordx4 %0, %3, off offset:160 sc1\n\t"
                   "global_load_dwordx4 %1, %3, off offset:176 sc1\n\t"
                   "global_load_dwordx4 %2, %3, off offset:192 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(v[SEQ_Q - 3]), "=&v"(v[SEQ_Q - 2]), "=&v"(v[SEQ_Q - 1])
                   : "v"(p)
                   : "memory");
  } else {
#pragma unroll
    for (int q = 0; q < SEQ_Q; ++q) v[q] = q < nq ? *reinterpret_cast<svs_gptr_f4>(p + 4 * q) : svs_f4{0.f, 0.f, 0.f, 0.f};
  }
}
__shared__ SeqShared g_seq_sh;      // file scope: the routine below is a real call, and a reference handed through a call would be a flat pointer too
template <bool COH>
__device__ __noinline__ float exact_seq_sum_f32(const float *t_flat, int n) {
  SeqShared &sh = g_seq_sh;
  svs_gptr_f32 t = (svs_gptr_f32)t_flat;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = max(4, (div_up(n, TRK_THREADS) + 3) & ~3);      // terms per lane: a multiple of four
  if (S > SEQ_RUN_MAX) {                                         // images beyond 640 x 512: the chain (never with the reference's cameras)
    __syncthreads();
    if (wave == 0) { const float v = seq_sum_f32<COH, svs_gptr_f32>(t, n); if (lane == 0) { sh.result = v; sh.fell_back = 1; } }
    __syncthreads();
    return sh.result;
  }
  const int nq = S / 4;
  const int j0 = min(n, tid * S), j1 = min(n, j0 + S);
  const bool empty = j0 >= j1;
  svs_f4 v[SEQ_Q];
  seq_load_run<COH>(t + (empty ? 0 : j0), nq, v);
  double ps = 0;
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < SEQ_Q; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (!(q < nq && j0 + 4 * q + c < j1)) v[q][c] = 0.f;      // past my run / past the last term: zeros add nothing anywhere below
      ps += (double)v[q][c];
      cnt += v[q][c] != 0.f ? 1 : 0;
    }
  double incl = ps;
  int cincl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(incl, d, 64);
    const int oc = __shfl_up(cincl, d, 64);
    if (lane >= d) { incl += o; cincl += oc; }
  }
  __syncthreads();                                   // (sh may still be read by the walker of the previous call)
  if (lane == 63) { sh.wave_P[wave] = incl; sh.wave_c[wave] = cincl; }
  __syncthreads();
  double base = 0;
  int cbase = 0;
  for (int w = 0; w < wave; ++w) { base += sh.wave_P[w]; cbase += sh.wave_c[w]; }
  // exclusive prefix as a sum of the earlier runs (never a difference: the bound in svs_seq_safe covers f64 sums of non-negative terms only)
  const double excl = __shfl_up(incl, 1, 64);
  const int cexcl = __shfl_up(cincl, 1, 64);
  const double P_s = base + (lane ? excl : 0.0), P_e = P_s + ps;
  const int c_s = cbase + (lane ? cexcl : 0), c_e = c_s + cnt;
  int eb = 0;
  bool safe = empty || svs_seq_safe(P_s, P_e, c_s, c_e, &eb);
  sh.eb[tid] = (safe && !empty) ? (unsigned char)eb : (unsigned char)0;
  __syncthreads();
  // two neighbouring safe runs share their binade (P_s of one is P_e of the other); nothing below relies on that proof: a change of binade ends the group
  if (safe && !empty && tid > 0) { const int pe = sh.eb[tid - 1]; if (pe != 0 && pe != eb) safe = false; }
  const unsigned long long um = __ballot(!safe);
  if (lane == 0) { sh.unsafe[wave] = um; sh.wave_u[wave] = __popcll(um); }
  __syncthreads();
  int ubase = 0;
  for (int w = 0; w < wave; ++w) ubase += sh.wave_u[w];
  const int uidx = ubase + __popcll(um & ((1ull << lane) - 1ull));
  const int slot = (!safe && (uidx + 1) * S <= SEQ_STAGE_FLOATS) ? uidx : -1;
  sh.slot[tid] = (short)slot;
  if (!safe) sh.eb[tid] = 0;
  SvsSeqMap m{0, 0};
  if (safe) {
#pragma unroll
    for (int q = 0; q < SEQ_Q; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) svs_seq_add_term_fast(m, v[q][c], eb);      // (zeros past the run add nothing)
  } else if (slot >= 0) {
#pragma unroll
    for (int q = 0; q < SEQ_Q; ++q)
      if (q < nq) *reinterpret_cast<svs_f4 *>(&sh.stage[slot * S + 4 * q]) = v[q];
  }
  bool flag = !safe || lane == 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o0 = __shfl_up(m.d0, d, 64), o1 = __shfl_up(m.dd, d, 64), of = __shfl_up((int)flag, d, 64);
    if (lane >= d && !flag) { m = svs_seq_compose(SvsSeqMap{o0, o1}, m); flag = of != 0; }
  }
  sh.pref[tid] = m;
  __syncthreads();
  if (wave == 0) {
    float acc = 0.f;
    bool ok = true;
    const int nseg = div_up(n, S);
    for (int chunk = 0; chunk * 64 < nseg && ok; ++chunk) {
      const int limit = min(64, nseg - chunk * 64);
      const unsigned long long mask = sh.unsafe[chunk];
      // this chunk's 64 runs, one per lane, in registers: the walk below is a dependent chain, and a readlane is a tenth of an LDS round trip
      const SvsSeqMap my = sh.pref[chunk * 64 + lane];
      const int my_eb = sh.eb[chunk * 64 + lane], my_sl = sh.slot[chunk * 64 + lane];
      // the terms of an unsafe run, one per lane (staged in LDS by the run's own lane, else from global memory); the NEXT unsafe run's terms are requested
      // before this one's chain of adds starts, so that only the first LDS round trip of a chunk is exposed
      auto run_terms = [&](int idx) {
        const int seg = chunk * 64 + idx, sj0 = seg * S, len = min(S, n - sj0), sl = __builtin_amdgcn_readlane(my_sl, idx);      // len <= SEQ_RUN_MAX <= 64
        // (two loads and a select of VALUES: a select of the two addresses would make a flat pointer out of an LDS and a global one)
        float x = sh.stage[(sl >= 0 && lane < len) ? sl * S + lane : 0];
        if (sl < 0) x = seq_term_load<COH, svs_gptr_f32>(t + sj0 + (lane < len ? lane : 0));
        return lane < len ? x : 0.f;
      };
      unsigned long long m = limit < 64 ? mask & ((1ull << limit) - 1ull) : mask;
      float x_next = m ? run_terms((int)__builtin_ctzll(m)) : 0.f;
      int pos = 0;
      while (ok) {
        const int nxt = m ? (int)__builtin_ctzll(m) : limit;
        if (nxt > pos) {
          const SvsSeqMap g{__builtin_amdgcn_readlane(my.d0, nxt - 1), __builtin_amdgcn_readlane(my.dd, nxt - 1)};
          ok = svs_seq_apply(&acc, g, __builtin_amdgcn_readlane(my_eb, nxt - 1));
        }
        if (nxt >= limit || !ok) break;
        m &= m - 1ull;
        const float x = x_next;
        if (m) x_next = run_terms((int)__builtin_ctzll(m));
#pragma unroll
        for (int l = 0; l < SEQ_RUN_MAX; ++l) acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));      // (+ 0 past the run: exact)
        pos = nxt + 1;
      }
    }
    if (!ok) acc = seq_sum_f32<COH, svs_gptr_f32>(t, n);
    if (lane == 0) { sh.result = acc; if (!ok) sh.fell_back = 1; }
  }
  __syncthreads();
  return sh.result;
}

// The exact float sums of a near-tie when several workgroups share a stream (latency mode, continuation launch).  Every workgroup must take the same decision, the
// sums must be formed ONCE (round 5 formed them in all eight workgroups), and nobody may start the next sweep -- which overwrites one of the two term buffers -- before
// the readers are done.  Workgroup 0 forms the trial pass's sum; when the accepted pass's sum is not known either (the level's first near-tie, or the
// first after an accepted step that needed no sums) workgroup 1 forms it AT THE SAME TIME on its own CU.  Each publishes ONE 8-byte word (count of such events << 32 | the
// float's bits: value and flag in one atomic store, no fence, no second store) in the stream's hand-over block (bc[13]: trial sum, bc[15]: accepted sum), and every
// workgroup waits for the word(s) it did not form itself.  n_dec / n_old: how many trial / accepted sums this stream has needed so far, this one included (all workgroups count alike).
struct NearSums { float seq_old, seq_new; bool failed; };
template <bool COH>
__device__ __forceinline__ NearSums shared_near_sums(const float *t_old, const float *t_new, int n, int wg, int nwg, bool need_old, float seq_old_in, unsigned n_dec, unsigned n_old,
                                                     double *bc, unsigned *fail_word, unsigned &n_exact) {
  __shared__ unsigned long long s_words[2];
  __shared__ int s_nf;
  NearSums R{seq_old_in, 0.f, false};
  if (nwg <= 1) {                                    // one workgroup: both sums here, nothing to hand over
    if (need_old) { R.seq_old = exact_seq_sum_f32<COH>(t_old, n); ++n_exact; }
    R.seq_new = exact_seq_sum_f32<COH>(t_new, n); ++n_exact;
    return R;
  }
  unsigned long long *w_new = reinterpret_cast<unsigned long long *>(bc + 13), *w_old = reinterpret_cast<unsigned long long *>(bc + 15);
  if (wg == 0) {
    R.seq_new = exact_seq_sum_f32<COH>(t_new, n); ++n_exact;
    if (threadIdx.x == 0) __hip_atomic_store(w_new, ((unsigned long long)n_dec << 32) | (unsigned long long)__float_as_uint(R.seq_new), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (wg == 1 && need_old) {
    R.seq_old = exact_seq_sum_f32<COH>(t_old, n); ++n_exact;
    if (threadIdx.x == 0) __hip_atomic_store(w_old, ((unsigned long long)n_old << 32) | (unsigned long long)__float_as_uint(R.seq_old), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {
    int nf = 0;
    auto wait_word = [&](const unsigned long long *w, unsigned want) {
      unsigned long long v = 0;
      long spin = 0;
      for (; spin < (1l << 24); ++spin) { v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((unsigned)(v >> 32) >= want) break; __builtin_amdgcn_s_sleep(1); }
      if (spin >= (1l << 24)) nf = 1;
      return v;
    };
    s_words[0] = wg == 0 ? 0ull : wait_word(w_new, n_dec);
    s_words[1] = (need_old && wg != 1) ? wait_word(w_old, n_old) : 0ull;
    if (nf) __hip_atomic_store(fail_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_nf = nf;
  }
  __syncthreads();
  if (wg != 0) R.seq_new = __uint_as_float((unsigned)s_words[0]);
  if (need_old && wg != 1) R.seq_old = __uint_as_float((unsigned)s_words[1]);
  R.failed = s_nf != 0;
  __syncthreads();                                   // (s_words is rewritten at the next near-tie)
  return R;
}

// MULTI: a few streams only (latency mode) -- gridDim.x workgroups share every sweep of one stream.  Each leaves its 29 partial
// sums in global memory, a device-scope counter barrier (all workgroups of the launch are resident: NW * batch <= #CUs) lets
// every workgroup add the partials up in the same fixed order, and each then runs the identical LM bookkeeping redundantly,
// so no pose has to be broadcast.  Partials are double-buffered by pass parity: a workgroup can only overwrite a buffer after
// everyone has passed the barrier of the pass in between, i.e. after everyone has read it.
// The cross-workgroup exchange below (partials, arrival counter, hand-over of the coarse level's pose) uses RELAXED agent-scope atomics plus a raw
// `s_waitcnt vmcnt(0)` instead of release / acquire fences: on gfx950 (and gfx942) vmcnt counts stores as well as loads and sc1 atomics write through /
// bypass the non-coherent per-XCD L2, so "all my stores have left" + "flag store" is a release in practice.  This is OUTSIDE the HIP memory model and wrong on
// any target that counts stores separately (vscnt: gfx10 and later) -- hence the hard stop.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "dense.hip: the relaxed-atomic + s_waitcnt vmcnt(0) hand-off is only valid on gfx950 / gfx942"
#endif
struct TrackMulti { double *part; unsigned *bar; int fail_off; double *bcast;
                    const int *map; const unsigned char *nwg_of;
                    float *terms; size_t terms_b; unsigned *seq_stats; int terms_only; int *work16; };      // work16 (optional, flat kernel): [batch] the stream's passes counted as 16 / 4 / 1 per sweep of level 0 / 1 / 2      // the float terms of the accepted and of the trial pass, [batch][2][terms_b] (null: accept test on the f64 sums alone); [0] exact sums formed, [1] fallbacks to the chain      // BAL only: workgroup -> (stream << 4 | part), workgroups per stream      // bcast [batch][16]: the pose after the coarsest level + a ready word (zeroed before the launch)      // [batch][2][nwg][32]; [batch] arrival counters + [batch] failure flags at bar + fail_off (zeroed before the launch)
// MINW = minimum waves per SIMD the register allocation must allow: 2 (<= 256 VGPRs; the kernel takes 147: one 8-wave
// workgroup per CU) when there is at most one stream per CU, 4 (<= 128 VGPRs, a few spills, two workgroups per CU) for
// bigger batches, where the second resident workgroup hides the first one's dependent chains: 0.55 -> 0.45 ms per 256 streams.
// BAL ("trk_balance", big batches): MULTI with a per-stream number of workgroups.  All streams of a big batch are resident at once (two workgroups per
// CU) and the launch lasts as long as the stream with the most LM passes (bench batch: 5.6 .. 17.8 level-0 sweeps per stream, mean 8.8).  The streams that needed
// the most sweeps in the LAST frame get 2 .. 4 workgroups each (trk_assign_kernel), placed first in the grid so that they are resident together; the
// single-workgroup streams follow in order of decreasing work, the shortest ones start in the slots the first finishers leave.
template <bool U8SRC, bool MULTI, int MINW, bool SEQ = false, bool BAL = false>      // SEQ: "trk_seq_chi2" (its own instantiation: the hot ones keep their register budget)
__global__ __launch_bounds__(TRK_THREADS, MINW) void dense_track_cpu_sem_kernel(TrackArgs A, double *__restrict__ T_io, int *__restrict__ passes_out, TrackMulti G) {
  static_assert(!BAL || !SEQ, "BAL: grid order (and, with MULTI, workgroups per stream) from the assignment tables");
  int bal_entry = 0;
  if constexpr (BAL) {
    bal_entry = G.map[blockIdx.x];
    if (bal_entry < 0 || (!MULTI && (bal_entry & 15) != 0)) return;      // idle workgroup / a sibling of a table made for the split variant: the order-only variant runs part 0 alone
  }
  double (&s_out)[NSUM + 1] = g_s_out;
  float (&s_iplut)[256] = g_iplut;
  __shared__ double s_T[12], s_Tn[12], s_x[6], s_H[27], s_Tj[3][12];
  __shared__ bool s_failed;
  if (threadIdx.x == 0) { s_failed = false; if (!SEQ && SVS_TRK_LAZY) g_seq_sh.fell_back = 0; }
#ifdef SVS_SCRATCH_PROBE                 // (build experiment: does the SIZE of a kernel's scratch allocation cost anything when nothing touches it?)
  if (passes_out == reinterpret_cast<int *>(1)) { volatile int big[SVS_SCRATCH_PROBE]; for (int i = 0; i < SVS_SCRATCH_PROBE; ++i) big[i] = i; T_io[0] = big[threadIdx.x % SVS_SCRATCH_PROBE]; }
#endif
  constexpr int TM = SEQ ? 1 : (SVS_TRK_LAZY ? (MULTI ? 2 : 1) : 0);      // the terms of a pass: plain stores, or past the caches when sibling workgroups read them
  unsigned n_exact = 0, n_decisions = 0, n_old_sums = 0;      // exact float sums formed by this workgroup; near-ties of this stream so far / those of them that needed the accepted pass's sum as well (shared_near_sums)
  const int slot = BAL ? (bal_entry >> 4) : (MULTI ? blockIdx.y : blockIdx.x), wg = BAL ? (bal_entry & 15) : (MULTI ? blockIdx.x : 0);
  const int nwg = !MULTI ? 1 : (BAL ? (int)G.nwg_of[slot] : (int)gridDim.x);
  const int first = wg * TRK_THREADS + threadIdx.x;
  int sweep = 0;
  auto all_workgroups = [&]() {      // s_out[0..NSUM] <- sum over the workgroups of this stream
    if (!MULTI) return;
    if (BAL && nwg == 1) return;
    if constexpr (!BAL) {
      // Round 6: every 8-byte word carries its own flag (4 bytes of data + the sweep number: what NCCL calls the LL protocol).  A workgroup publishes its 29 sums as 58
      // such words and reads the others' the moment they carry this sweep's number: ONE store-to-load hop between workgroups -- no drain of the store queue, no arrival
      // counter (a device-scope read-modify-write), no second wait.  Rounds 2-5: stores, s_waitcnt vmcnt(0), barrier, counter += 1, poll the counter, barrier, loads --
      // 4-5 us of a 10 us pass in latency mode.  Two parities of buffers: a word is overwritten two sweeps later, which its writer can only reach after every sibling has
      // consumed this sweep (it needs their next sweep's words to get there).  The buffers are zeroed before the launch (sweep numbers start at 1).
      const unsigned ep = (unsigned)(++sweep);
      unsigned long long *buf = reinterpret_cast<unsigned long long *>(G.part) + ((size_t)slot * 2 + (ep & 1u)) * (size_t)nwg * 64;
      bool timed_out = false;
      if (threadIdx.x < 2 * (NSUM + 1)) {
        const int k = threadIdx.x;
        const double mine = s_out[k >> 1];
        const unsigned my_half = (k & 1) ? (unsigned)__double2hiint(mine) : (unsigned)__double2loint(mine);
        __hip_atomic_store(buf + (size_t)wg * 64 + k, ((unsigned long long)ep << 32) | my_half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double acc = 0;
        long budget = 1l << 24;
        for (int w = 0; w < nwg; ++w) {                      // workgroup order: the same sum in every workgroup
          unsigned h = my_half;
          if (w != wg) {
            unsigned long long v = __hip_atomic_load(buf + (size_t)w * 64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((unsigned)(v >> 32) != ep && budget > 0) { __builtin_amdgcn_s_sleep(1); --budget; v = __hip_atomic_load(buf + (size_t)w * 64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if ((unsigned)(v >> 32) != ep) timed_out = true;
            h = (unsigned)v;
          }
          const unsigned other = (unsigned)__shfl_xor((int)h, 1, 64);      // even lane: the low half is mine, the high half my neighbour's
          acc += __hiloint2double((int)((k & 1) ? h : other), (int)((k & 1) ? other : h));
        }
        if (!(k & 1)) s_out[k >> 1] = acc;                     // (this wave published from s_out before any lane of it writes here)
      }
      // a sibling never published (not resident: the device is shared with other work): raise the stream's failure flag -- the pose is left as it came in, passes_out reports -1
      if (__syncthreads_or(timed_out ? 1 : 0)) {
        if (threadIdx.x == 0) { __hip_atomic_store(G.bar + G.fail_off + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_failed = true; }
        __syncthreads();
      }
      return;
    }
    // write-through 8-byte words + one arrival counter (relaxed, agent scope): no fence that would write an XCD's whole L2 back
    double *buf = G.part + ((size_t)slot * 2 + (sweep & 1)) * (BAL ? 4 : nwg) * 32;
    if (threadIdx.x <= NSUM) __hip_atomic_store(buf + wg * 32 + threadIdx.x, s_out[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++sweep;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(G.bar + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)sweep * (unsigned)nwg;
      long spin = 0;
      for (; spin < (1l << 24) && __hip_atomic_load(G.bar + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin) __builtin_amdgcn_s_sleep(1);
      // a sibling workgroup never arrived (they are not all resident: the device is shared with other work).  Summing what is
      // there would let the replicated LM bookkeeping diverge silently: raise the stream's failure flag instead -- every
      // workgroup sees it at its next barrier at the latest, the pose is left as it came in and passes_out reports -1.
      if (spin >= (1l << 24)) __hip_atomic_store(G.bar + G.fail_off + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_failed = __hip_atomic_load(G.bar + G.fail_off + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    }
    __syncthreads();
    if (threadIdx.x <= NSUM) {
      double acc = 0;
      for (int w = 0; w < nwg; ++w) acc += __hip_atomic_load(buf + w * 32 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_out[threadIdx.x] = acc;
    }
    __syncthreads();
  };
  if (threadIdx.x < 12) s_T[threadIdx.x] = T_io[(size_t)slot * 12 + threadIdx.x];
  for (int i = threadIdx.x; i < 256; i += TRK_THREADS) s_iplut[i] = (float)((1. / 255.) * i);
  __syncthreads();
  int passes = 0, n_rec = 0;
  bool failed = false;
  // One fused pass per LM iteration.  The reference runs, per iteration, an H,b pass at T and a
  // chi2 pass at T_new, and after an accepted step starts the next iteration with an H,b pass at
  // that same T_new.  Evaluating chi2 AND H,b together at T_new therefore serves both (identical
  // sums, one gather sweep instead of two).  A rejected step makes the reference repeat the same
  // undamped solve at the unchanged T (mu is never applied, dense_tracking.cpp:332), reject again and
  // stop (trial == 2, :379-384): that is "stop on the first rejection" here.
  for (int level = 2; level >= 0; --level) {
    // MULTI: the coarsest level (1/16 of the samples: less than one per lane and workgroup) is run by workgroup 0 ALONE -- a sweep of it is
    // shorter than the cross-workgroup exchange that sharing it would cost -- and the others pick the pose up when it is done
    const bool solo = MULTI && level == 2 && !(BAL && nwg == 1);
    if (solo && wg != 0) {
      if (threadIdx.x == 0) {
        long spin = 0;
        for (; spin < (1l << 24) && __hip_atomic_load(reinterpret_cast<unsigned *>(G.bcast + (size_t)slot * 16 + 12), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u; ++spin)
          __builtin_amdgcn_s_sleep(2);
        if (spin >= (1l << 24)) s_failed = true;
      }
      __syncthreads();
      if (s_failed) { failed = true; break; }
      if (threadIdx.x < 12) s_T[threadIdx.x] = __hip_atomic_load(G.bcast + (size_t)slot * 16 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      continue;
    }
    const int lnwg = solo ? 1 : nwg;
    LevelArgs L = A.lv[level];
    L.cloud += slot * A.cloud_b[level]; L.prev += slot * A.prev_b[level];
    if (U8SRC) L.cur8 += slot * A.c8_b[level];
    else { L.cur += slot * A.f_b[level]; L.dx += slot * A.f_b[level]; L.dy += slot * A.f_b[level]; }
    const int n_lvl = (L.cam.w / 4) * (L.cam.h / 4);
    // the operands of this level's sweeps (track_pass_call): the level, which workgroups share it; pose and term buffer follow before each sweep
    __syncthreads();
    if (MINW != 2 && threadIdx.x == 0) { g_pa.L = L; g_pa.wg = solo ? 0 : wg; g_pa.nwg = lnwg; }
    auto sweep_at = [&](const double *s_pose, float *t_buf) {      // s_pose: 12 doubles in LDS
      if constexpr (MINW == 2) {
        // one workgroup per CU (latency mode, small batches): 256 registers to live in -- the sweep inline, as it always was; its passes are a few
        // microseconds long and a call (operands through LDS, callee-saved registers to scratch and back) would cost as much again
        track_pass<true, U8SRC, TM>(L, s_pose, g_s_part, g_s_out, g_iplut, solo ? (int)threadIdx.x : first, lnwg, t_buf);
      } else {
        if (threadIdx.x < 12) g_pa.T[threadIdx.x] = s_pose[threadIdx.x];
        if (threadIdx.x == 12) g_pa.t_buf = t_buf;
        __syncthreads();
        track_pass_call<U8SRC, TM>();
      }
    };
    // SEQ: one term buffer, every sum by the chain ("trk_seq_chi2"; never MULTI: G.part carries the buffer and fail_off its stream stride).
    // Default: two buffers -- the terms of the accepted pass (tb[cur]) and of the trial -- for the sums the accept test cannot decide in f64 (seqsum.h)
    float *tb[2];
    tb[0] = SEQ ? reinterpret_cast<float *>(G.part) + (size_t)slot * G.fail_off : (G.terms ? G.terms + (size_t)slot * 2 * G.terms_b : nullptr);
    tb[1] = SEQ ? tb[0] : (G.terms ? tb[0] + G.terms_b : nullptr);
    int cur = 0;
    sweep_at(s_T, tb[cur]);        // chi2 (dense_tracking.cpp:229-261) + H,b of iteration 0
    if (!solo) all_workgroups();
    if (MULTI && s_failed) { failed = true; break; }
    ++passes;
    float chi2;
    double S_old = s_out[27];                  // the f64 sum of the same float terms: exact to 1e-13
    int nv_old = (int)s_out[NSUM];             // samples that contributed a term (adding the zero of a skipped sample is exact)
    float seq_old = 0.f;
    bool seq_old_ok = false;
    if constexpr (SEQ) chi2 = seq_chi2_f32(tb[0], n_lvl); else chi2 = (float)S_old;
    if (threadIdx.x < 27) s_H[threadIdx.x] = s_out[threadIdx.x];
    if (A.rec && wg == 0 && threadIdx.x == 0 && n_rec < A.rec_cap) A.rec[(size_t)slot * A.rec_cap + n_rec] = svs_dense_lm_record{level, 2, chi2, chi2};
    ++n_rec;
    __syncthreads();
    bool stop = false;
    for (int it = 0; it < 15 && !stop; ++it) {
      if (threadIdx.x >= 64 && threadIdx.x < 76) s_Tj[level][threadIdx.x - 64] = s_T[threadIdx.x - 64];   // the reference's H,b pass of this iteration ran at s_T
      if (threadIdx.x < 64) {                        // H.ldlt().solve(-Jres): undamped (dense_tracking.cpp:332) -- on seven lanes of wave 0
        const int lane = threadIdx.x;
        double col[6], x[6], Tc[12], Tn[12];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const int lo = r < lane ? r : lane, hi = r < lane ? lane : r;      // packed upper by column: (lo, hi) at hi (hi + 1) / 2 + lo
          col[r] = lane < 6 ? s_H[hi * (hi + 1) / 2 + lo] : (lane == 6 ? -s_H[21 + r] : (r == 0 ? 1.0 : 0.0));
        }
        wave_solve6(col, x);
#pragma unroll
        for (int i = 0; i < 12; ++i) Tc[i] = s_T[i];
        mo2_exp_mul(x, Tc, Tn);
        if (lane < 12) {
          double v = Tn[0];
#pragma unroll
          for (int i = 1; i < 12; ++i) v = lane == i ? Tn[i] : v;
          s_Tn[lane] = v;
        }
        if (lane < 6) {
          double v = x[0];
#pragma unroll
          for (int i = 1; i < 6; ++i) v = lane == i ? x[i] : v;
          s_x[lane] = v;
        }
      }
      __syncthreads();
      sweep_at(s_Tn, tb[cur ^ 1]);      // new_chi2 (:335-367) + H,b for the next iteration
      if (!solo) all_workgroups();
      if (MULTI && s_failed) { failed = true; break; }
      ++passes;
      float new_chi2;
      bool accept;
      const double S_new = s_out[27];
      const int nv_new = (int)s_out[NSUM];
      if constexpr (SEQ) {
        new_chi2 = seq_chi2_f32(tb[0], n_lvl);
        accept = (double)chi2 - (double)new_chi2 > 0;
      } else {
        // the reference compares two float sums of n terms each: |sum_float - sum_exact| <= ((1 + 2^-24)^(n - 1) - 1) sum_exact.  Outside that band the f64 sums
        // decide as the float sums would; inside it the float sums are formed, bit for bit (exact_seq_sum_f32)
        const double gam = 1.0001 * (double)max(nv_old, nv_new) * 5.9604644775390625e-08 + 1e-12;
        const bool near = SVS_TRK_LAZY && tb[0] && !G.terms_only && !(fabs(S_old - S_new) > gam * (S_old + S_new));      // (terms_only: kernel A/B -- the stores without the sums)
        if (near) {
          float seq_new = 0.f;
          {
            const bool shared = MULTI && !solo;      // (the coarsest level of a shared stream is run by its leader alone: its near-ties are not counted, the siblings never see them)
            if (shared) { ++n_decisions; if (!seq_old_ok) ++n_old_sums; }
            const NearSums ns = shared_near_sums<MULTI>(tb[cur], tb[cur ^ 1], n_lvl, shared ? wg : 0, shared ? nwg : 1, !seq_old_ok, seq_old, n_decisions, n_old_sums,
                                                        MULTI ? G.bcast + (size_t)slot * 16 : nullptr, MULTI ? G.bar + G.fail_off + slot : nullptr, n_exact);
            if (ns.failed) { s_failed = true; failed = true; break; }
            seq_old = ns.seq_old; seq_new = ns.seq_new;
          }
          seq_old_ok = true;
          chi2 = seq_old; new_chi2 = seq_new;
          accept = (double)chi2 - (double)new_chi2 > 0;
          if (accept) seq_old = seq_new;
        } else {
          new_chi2 = (float)S_new;
          accept = S_old > S_new;
          if (accept) seq_old_ok = false;
        }
        if (accept) { S_old = S_new; nv_old = nv_new; cur ^= 1; }
      }
      if (A.rec && wg == 0 && threadIdx.x == 0 && n_rec < A.rec_cap) A.rec[(size_t)slot * A.rec_cap + n_rec] = svs_dense_lm_record{level, accept ? 1 : 0, chi2, new_chi2};
      ++n_rec;
      if (accept) {
        chi2 = new_chi2;
        double mx = -1;
        for (int q = 0; q < 6; ++q) mx = fmax(mx, fabs(s_x[q]));
        stop = mx <= 1e-10;
        if (threadIdx.x < 12) s_T[threadIdx.x] = s_Tn[threadIdx.x];
        if (threadIdx.x < 27) s_H[threadIdx.x] = s_out[threadIdx.x];
        __syncthreads();
      } else {
        stop = true;
      }
    }
    if (failed) break;
    if (solo) {                                      // workgroup 0 hands the pose of the coarsest level to its siblings
      __syncthreads();
      if (threadIdx.x < 12) __hip_atomic_store(G.bcast + (size_t)slot * 16 + threadIdx.x, s_T[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned *>(G.bcast + (size_t)slot * 16 + 12), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (wg != 0) return;
  if (failed) { if (threadIdx.x == 0 && passes_out) passes_out[slot] = -1; return; }      // pose left as it came in
  if (threadIdx.x < 12) T_io[(size_t)slot * 12 + threadIdx.x] = s_T[threadIdx.x];
  if (threadIdx.x == 0 && passes_out) passes_out[slot] = passes;
  if (threadIdx.x == 0 && A.n_rec) A.n_rec[slot] = n_rec;
  if (A.T_jac && threadIdx.x < 36) A.T_jac[(size_t)slot * 36 + threadIdx.x] = s_Tj[threadIdx.x / 12][threadIdx.x % 12];
  if (threadIdx.x == 0 && G.seq_stats && n_exact) {
    atomicAdd(G.seq_stats, n_exact);
    if (SVS_TRK_LAZY && g_seq_sh.fell_back) atomicAdd(G.seq_stats + 1, 1u);
  }
}

// ---- big batches, round 6: the LM loop as a flat state machine around ONE inlined sweep ------------------------------------------------------------------------
// dense_track_cpu_sem_kernel<., false, 4> calls its sweep (track_pass_call): the sweep fills the 128-register budget by itself, so as a callee it saves and restores every
// callee-saved register it touches -- 29 dwords per lane and call, 59 KB per workgroup each way, 18 calls per frame: 0.55 GB written (and read back) per 512-stream batch for
// nothing (round 5's rocprof: WRITE_SIZE 996 MB against 0.34 GB of terms).  Inlining it into the level / iteration loops was worse: everything those loops keep alive (sums,
// flags, buffers: uniform values the compiler cannot prove uniform) shared the sweep's allocation and spilled inside the loop over the samples.
// Here the LM state of the stream lives in LDS (TrkState: one lane writes it, everybody reads what it needs), the kernel is ONE loop { sweep; step }, and nothing of the step
// is alive while the sweep runs: no call on the hot path, no callee-saved traffic, no spills in the sweep.  Same sweep, same sums, same solve, same decisions as the kernel
// above: bit-identical results (tests/test_gpu_frontend.py::test_flat_tracker_kernel_is_bit_identical).
struct TrkState {
  double S_old; float chi2, seq_old;
  int work16;                                                        // passes so far, weighted 16 / 4 / 1 by level (what trk_assign_kernel orders the next frame's grid by)
  int level, it, cur, passes, n_rec, nv_old, seq_old_ok, phase;      // phase 0: the sweep just run was the level's first (chi2 + H,b at the accepted pose); 1: a trial
  unsigned n_exact;
};
__shared__ TrkState g_ts;
template <bool U8SRC, int TM>
__device__ __forceinline__ void track_pass_from_lds() {      // the body of track_pass_call, inlined: operands out of g_pa into scalar registers
  LevelArgsG L;
  L.cloud = (const SVS_AS1 float *)uni_ptr(g_pa.L.cloud); L.prev = (const SVS_AS1 uint8_t *)uni_ptr(g_pa.L.prev);
  L.cur = (const SVS_AS1 float *)uni_ptr(g_pa.L.cur); L.dx = (const SVS_AS1 float *)uni_ptr(g_pa.L.dx); L.dy = (const SVS_AS1 float *)uni_ptr(g_pa.L.dy);
  L.cur8 = (const SVS_AS1 uint8_t *)uni_ptr(g_pa.L.cur8);
  L.pstride = uni_i32(g_pa.L.pstride); L.fstride = uni_i32(g_pa.L.fstride); L.c8stride = uni_i32(g_pa.L.c8stride);
  L.cam.f = uni_f64(g_pa.L.cam.f); L.cam.cx = uni_f64(g_pa.L.cam.cx); L.cam.cy = uni_f64(g_pa.L.cam.cy); L.cam.b = 0;
  L.cam.w = uni_i32(g_pa.L.cam.w); L.cam.h = uni_i32(g_pa.L.cam.h);
  const int nwg = uni_i32(g_pa.nwg), first = uni_i32(g_pa.wg) * TRK_THREADS + (int)threadIdx.x;
  track_pass<true, U8SRC, TM, LevelArgsG>(L, g_pa.T, g_s_part, g_s_out, g_iplut, first, nwg, uni_ptr(g_pa.t_buf));
}
// the solve of an iteration, on wave 0 (seven lanes): H.ldlt().solve(-Jres), undamped (dense_tracking.cpp:332) + exp(x) * T.  A function of its own, called by ONE
// wave: its 70-odd live doubles are not part of the kernel's allocation, and whatever it saves on entry it saves for 64 lanes, not 512.
__shared__ double g_sT[12], g_sTn[12], g_sx[6], g_sH[27], g_sTj[3][12];
__device__ __noinline__ void track_solve_call() {
  const int lane = threadIdx.x;
  double col[6], x[6], Tc[12], Tn[12];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int lo = r < lane ? r : lane, hi = r < lane ? lane : r;      // packed upper by column: (lo, hi) at hi (hi + 1) / 2 + lo
    col[r] = lane < 6 ? g_sH[hi * (hi + 1) / 2 + lo] : (lane == 6 ? -g_sH[21 + r] : (r == 0 ? 1.0 : 0.0));
  }
  wave_solve6(col, x);
#pragma unroll
  for (int i = 0; i < 12; ++i) Tc[i] = g_sT[i];
  mo2_exp_mul(x, Tc, Tn);
  if (lane < 12) {
    double v = Tn[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) v = lane == i ? Tn[i] : v;
    g_sTn[lane] = v;
  }
  if (lane < 6) {
    double v = x[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) v = lane == i ? x[i] : v;
    g_sx[lane] = v;
  }
}
// ---- the tail of a big batch: the CONTINUATION launch ("trk_split", VERDICT round 5 item 2 iii) ------------------------------------------------------------------
// All streams of a big batch are resident at once (two workgroups per CU) and the launch lasts as long as its longest stream: 16.75 level-0 sweeps in the bench's batch
// against a mean of 8.7 -- and the last sweeps of a stream are the ones whose accept test needs the exact float sums, one after the other on the critical path.  Predicting
// the long streams from the last frame does not work (correlation -0.18 in the bench).  So the first launch (CONT = 0) runs every stream with one workgroup as before, but a
// stream that has taken K trials on level 0 and is still not done PARKS: its LM state -- the accepted pose, H,b of the accepted pass, the sums the accept test carries,
// the term buffers (they are in global memory already) -- goes to global memory, its index to a list, and the workgroup exits.  The second launch (CONT = 1) takes the list:
// every parked stream is resumed by NW workgroups (NW = what fits the device for the number of parked streams, decided on the device) that share each sweep exactly like
// the latency mode does (partial sums through write-through words + one arrival counter, the leader forms the exact float sums and publishes the decision).  A trial of a
// parked stream then takes ~30 us instead of ~170.  K is a constant, so which sweeps run in which form is a function of the data alone: deterministic results; the
// partial sums of a shared sweep are added in another order than the one-workgroup sweep's, so H,b differ from the unsplit run in their last bits (pose 1e-12), while the
// accept decisions are the reference's either way (exact float sums of the terms).
struct TrkPark { TrkState ts; double T[12], H[27], Tj[36]; };
struct TrackCont { TrkPark *park; int *list; unsigned *count; int K; int slots; };      // K < 0: no parking; slots: resident workgroup slots of the device for the second launch
constexpr int CONT_MAX_NWG = 8;
template <bool U8SRC, bool BAL, int CONT>      // CONT 0: first launch (BAL: grid order from the table); 1: the continuation (grid = CONT_MAX_NWG x batch, blockIdx.y = index into the list)
__global__ __launch_bounds__(TRK_THREADS, TRK_MINW_BIG) void dense_track_batch_kernel(TrackArgs A, double *__restrict__ T_io, int *__restrict__ passes_out, TrackMulti G, TrackCont C) {
  constexpr bool MULTI = CONT == 1;
  constexpr int TM = SVS_TRK_LAZY ? (MULTI ? 2 : 1) : 0;
  int slot = blockIdx.x, wg = 0, nwg = 1;
  if constexpr (BAL && !MULTI) {
    const int e = G.map[blockIdx.x];
    if (e < 0 || (e & 15) != 0) return;      // idle workgroup / a sibling entry of a table made for the split variant
    slot = e >> 4;
  }
  if constexpr (MULTI) {
    const int n_parked = (int)*C.count;
    if ((int)blockIdx.y >= n_parked) return;
    // workgroups per parked stream: as many as are resident together (all of a stream's workgroups wait for each other; siblings are neighbours in the grid)
    nwg = C.slots / n_parked;
    nwg = nwg >= 8 ? 8 : (nwg >= 4 ? 4 : (nwg >= 2 ? 2 : 1));
    wg = blockIdx.x;
    if (wg >= nwg) return;
    slot = C.list[blockIdx.y];
  }
  const int tid = threadIdx.x;
  __shared__ bool s_failed;
  float *const tb0 = G.terms ? G.terms + (size_t)slot * 2 * G.terms_b : nullptr;
  // level set-up (one lane): the operands of the level's sweeps
  auto set_level = [&](int level) {
    LevelArgs L = A.lv[level];
    L.cloud += slot * A.cloud_b[level]; L.prev += slot * A.prev_b[level];
    if (U8SRC) L.cur8 += slot * A.c8_b[level];
    else { L.cur += slot * A.f_b[level]; L.dx += slot * A.f_b[level]; L.dy += slot * A.f_b[level]; }
    g_pa.L = L; g_pa.wg = wg; g_pa.nwg = nwg;
  };
  auto enter_level = [&](int level) {          // ... the first sweep of the level at the accepted pose into term buffer 0
    set_level(level);
    g_pa.t_buf = tb0;
    g_ts.level = level; g_ts.phase = 0; g_ts.cur = 0; g_ts.it = 0;
  };
  for (int i = tid; i < 256; i += TRK_THREADS) g_iplut[i] = (float)((1. / 255.) * i);
  if (tid == 0) { g_seq_sh.fell_back = 0; s_failed = false; }
  int sweep = 0;                                   // MULTI: sweeps shared so far (parity of the partials' buffer, target of the arrival counter)
  unsigned n_decisions = 0, n_old_sums = 0;          // near-ties of this stream so far / those that needed the accepted pass's sum too (shared_near_sums)
  if constexpr (!MULTI) {
    if (tid < 12) { const double v = T_io[(size_t)slot * 12 + tid]; g_sT[tid] = v; g_pa.T[tid] = v; }
    if (tid == 0) {
      g_ts.passes = 0; g_ts.n_rec = 0; g_ts.n_exact = 0; g_ts.work16 = 0;
      enter_level(2);
    }
    __syncthreads();
  } else {
    // resume: the state as the first launch left it (behind a trial step that did not end the level); every workgroup of the stream loads it and goes on with the solve
    const TrkPark &P = C.park[slot];
    if (tid == 0) { g_ts = P.ts; if (wg != 0) g_ts.n_exact = 0; }
    if (tid < 12) g_sT[tid] = P.T[tid];
    if (tid >= 64 && tid < 64 + 27) g_sH[tid - 64] = P.H[tid - 64];
    if (tid >= 128 && tid < 128 + 36) g_sTj[(tid - 128) / 12][(tid - 128) % 12] = P.Tj[tid - 128];
    __syncthreads();
    if (tid == 0) set_level(g_ts.level);
    __syncthreads();
  }
  // MULTI: g_s_out <- the sums of all workgroups of the stream, added in workgroup order (dense_track_cpu_sem_kernel::all_workgroups: relaxed agent-scope words + vmcnt(0)
  // + one arrival counter, valid on gfx950 / gfx942 only -- see the #error above)
  auto all_workgroups = [&]() {
    if (nwg == 1) return;
    double *buf = G.part + ((size_t)slot * 2 + (sweep & 1)) * CONT_MAX_NWG * 32;
    if (tid <= NSUM) __hip_atomic_store(buf + wg * 32 + tid, g_s_out[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++sweep;
    if (tid == 0) {
      __hip_atomic_fetch_add(G.bar + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)sweep * (unsigned)nwg;
      long spin = 0;
      for (; spin < (1l << 24) && __hip_atomic_load(G.bar + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin) __builtin_amdgcn_s_sleep(1);
      if (spin >= (1l << 24)) __hip_atomic_store(G.bar + G.fail_off + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_failed = __hip_atomic_load(G.bar + G.fail_off + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    }
    __syncthreads();
    if (tid <= NSUM) {
      double acc = 0;
      for (int w = 0; w < nwg; ++w) acc += __hip_atomic_load(buf + w * 32 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      g_s_out[tid] = acc;
    }
    __syncthreads();
  };
  bool failed = false, parked = false;
  bool resume = MULTI;                             // the continuation enters the loop behind a step: solve first
#pragma nounroll
  for (;;) {
    bool level_done = false;
    int level;
    if (!resume) {
      track_pass_from_lds<U8SRC, TM>();            // (ends in a barrier: g_s_out is complete)
      if constexpr (MULTI) { all_workgroups(); if (s_failed) { failed = true; break; } }
      // ---- the LM step: every lane reads the state, one lane writes it back
      level = g_ts.level;
      const int phase = g_ts.phase, cur = g_ts.cur, n_rec = g_ts.n_rec;
      const int n_lvl = (A.lv[level].cam.w / 4) * (A.lv[level].cam.h / 4);
      if (phase == 0) {                              // chi2 (dense_tracking.cpp:229-261) + H,b of iteration 0
        const double S = g_s_out[27];
        const float chi2 = (float)S;
        __syncthreads();                             // (everybody has read g_s_out / the state)
        if (tid < 27) g_sH[tid] = g_s_out[tid];
        if (tid == 0) {
          g_ts.S_old = S; g_ts.nv_old = (int)g_s_out[NSUM]; g_ts.chi2 = chi2; g_ts.seq_old_ok = 0; g_ts.seq_old = 0.f;
          ++g_ts.passes; ++g_ts.n_rec; g_ts.work16 += 16 >> (2 * level);
          if (A.rec && wg == 0 && n_rec < A.rec_cap) A.rec[(size_t)slot * A.rec_cap + n_rec] = svs_dense_lm_record{level, 2, chi2, chi2};
        }
      } else {                                       // new_chi2 (:335-367) + H,b for the next iteration
        const double S_old = g_ts.S_old, S_new = g_s_out[27];
        const int nv_old = g_ts.nv_old, nv_new = (int)g_s_out[NSUM];
        float chi2 = g_ts.chi2, new_chi2, seq_old = g_ts.seq_old;
        int seq_old_ok = g_ts.seq_old_ok;
        unsigned n_exact = 0;
        bool accept;
        // the reference compares two float sums of n terms each: |sum_float - sum_exact| <= ((1 + 2^-24)^(n - 1) - 1) sum_exact.  Outside that band the f64 sums decide as
        // the float sums would; inside it the float sums are formed, bit for bit (exact_seq_sum_f32)
        const double gam = 1.0001 * (double)max(nv_old, nv_new) * 5.9604644775390625e-08 + 1e-12;
        const bool near = SVS_TRK_LAZY && tb0 && !G.terms_only && !(fabs(S_old - S_new) > gam * (S_old + S_new));
        if (near) {
          float seq_new = 0.f;
          {
            ++n_decisions;
            if (!seq_old_ok) ++n_old_sums;
            const NearSums ns = shared_near_sums<MULTI>(tb0 + (size_t)cur * G.terms_b, tb0 + (size_t)(cur ^ 1) * G.terms_b, n_lvl, wg, nwg, !seq_old_ok, seq_old, n_decisions, n_old_sums,
                                                        MULTI ? G.bcast + (size_t)slot * 16 : nullptr, MULTI ? G.bar + G.fail_off + slot : nullptr, n_exact);
            if (ns.failed) { if (tid == 0) s_failed = true; failed = true; break; }
            seq_old = ns.seq_old; seq_new = ns.seq_new;
          }
          seq_old_ok = 1;
          chi2 = seq_old; new_chi2 = seq_new;
          accept = (double)chi2 - (double)new_chi2 > 0;
          if (accept) seq_old = seq_new;
        } else {
          new_chi2 = (float)S_new;
          accept = S_old > S_new;
          if (accept) seq_old_ok = 0;
        }
        double mx = -1;
        for (int q = 0; q < 6; ++q) mx = fmax(mx, fabs(g_sx[q]));
        const int it = g_ts.it + 1;
        level_done = !accept || mx <= 1e-10 || it >= 15;
        __syncthreads();                             // (everybody has read g_s_out / the state / g_sx / s_word)
        if (accept) {
          if (tid < 12) g_sT[tid] = g_sTn[tid];
          if (tid < 27) g_sH[tid] = g_s_out[tid];
        }
        if (tid == 0) {
          if (accept) { g_ts.S_old = S_new; g_ts.nv_old = nv_new; g_ts.cur = cur ^ 1; g_ts.chi2 = new_chi2; }
          g_ts.seq_old = seq_old; g_ts.seq_old_ok = seq_old_ok; g_ts.it = it; g_ts.n_exact += n_exact;
          ++g_ts.passes; ++g_ts.n_rec; g_ts.work16 += 16 >> (2 * level);
          if (A.rec && wg == 0 && n_rec < A.rec_cap) A.rec[(size_t)slot * A.rec_cap + n_rec] = svs_dense_lm_record{level, accept ? 1 : 0, chi2, new_chi2};
        }
        // first launch: a stream that is still going after K trials on the finest level parks here (state of the accepted pass, before the solve)
        if constexpr (!MULTI) {
          if (!level_done && level == 0 && C.K >= 0 && it >= C.K) parked = true;
        }
      }
      __syncthreads();                               // the state, g_sT and g_sH are those of the accepted pass
      if (parked) break;
    } else {
      resume = false;
      level = g_ts.level;
    }
    if (level_done) {
      if (level == 0) break;
      if (tid == 0) enter_level(level - 1);
      if (tid >= 64 && tid < 76) g_pa.T[tid - 64] = g_sT[tid - 64];
    } else {                                       // the next trial: solve at the accepted pose, sweep at the trial pose into the other term buffer
      if (tid >= 64 && tid < 76) g_sTj[level][tid - 64] = g_sT[tid - 64];      // the reference's H,b pass of this iteration ran at the accepted pose
      if (tid < 64) track_solve_call();
      __syncthreads();
      if (tid < 12) g_pa.T[tid] = g_sTn[tid];
      if (tid == 12) { g_pa.t_buf = tb0 ? tb0 + (size_t)(g_ts.cur ^ 1) * G.terms_b : nullptr; g_ts.phase = 1; }
    }
    __syncthreads();
  }
  if constexpr (!MULTI) {
    if (parked) {
      TrkPark &P = C.park[slot];
      if (tid == 0) { P.ts = g_ts; C.list[atomicAdd(C.count, 1u)] = slot; }
      if (tid < 12) P.T[tid] = g_sT[tid];
      if (tid >= 64 && tid < 64 + 27) P.H[tid - 64] = g_sH[tid - 64];
      if (tid >= 128 && tid < 128 + 36) P.Tj[tid - 128] = g_sTj[(tid - 128) / 12][(tid - 128) % 12];
      return;
    }
  }
  if (wg != 0) return;
  if (failed) { if (tid == 0 && passes_out) passes_out[slot] = -1; return; }      // (MULTI: a sibling never arrived) pose left as it came in
  if (tid < 12) T_io[(size_t)slot * 12 + tid] = g_sT[tid];
  if (tid == 0 && passes_out) passes_out[slot] = g_ts.passes;
  if (tid == 0 && A.n_rec) A.n_rec[slot] = g_ts.n_rec;
  if (tid == 0 && G.work16) G.work16[slot] = g_ts.work16;
  if (A.T_jac && tid < 36) A.T_jac[(size_t)slot * 36 + tid] = g_sTj[tid / 12][tid % 12];
  if (tid == 0 && G.seq_stats && g_ts.n_exact) {
    atomicAdd(G.seq_stats, g_ts.n_exact);
    if (SVS_TRK_LAZY && g_seq_sh.fell_back) atomicAdd(G.seq_stats + 1, 1u);
  }
}

// DenseTracker::residual_img[level] (dense_tracking.cpp:279-329): the GUI image an H,b pass at pose T leaves behind.
// Green = no depth, red = projects out of frame, grey = 1 - 50 res^2 of the clamped residual.
template <bool U8SRC>
__global__ __launch_bounds__(256) void residual_image_cpu_sem_kernel(LevelArgs L, size_t cloud_b, size_t prev_b, size_t f_b, size_t c8_b,
                                                                     const double *__restrict__ Tarr, size_t T_b,
                                                                     float *__restrict__ rimg, size_t rimg_b) {
  const int slot = blockIdx.y;
  const int cw = L.cam.w / 4, ch = L.cam.h / 4;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cw * ch) return;
  L.cloud += slot * cloud_b; L.prev += slot * prev_b;
  if (U8SRC) L.cur8 += slot * c8_b; else L.cur += slot * f_b;
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = Tarr[(size_t)slot * T_b + k];
  const int u = i % cw, v = i / cw;
  const float4 c4 = reinterpret_cast<const float4 *>(L.cloud)[i];
  float4 o = make_float4(0.f, 1.f, 0.f, 1.f);
  if (c4.w > 0) {
    const double xp0 = c4.x, xp1 = c4.y, xp2 = c4.z;
    const double x = T[0] * xp0 + T[1] * xp1 + T[2] * xp2 + T[3];
    const double y = T[4] * xp0 + T[5] * xp1 + T[6] * xp2 + T[7];
    const double z = T[8] * xp0 + T[9] * xp1 + T[10] * xp2 + T[11];
    const float uvx = (float)(L.cam.f * (x / z) + L.cam.cx);
    const float uvy = (float)(L.cam.f * (y / z) + L.cam.cy);
    bool ok = fabsf(uvx) < 1e9f && fabsf(uvy) < 1e9f;
    const int ui = ok ? (int)uvx : 0, vi = ok ? (int)uvy : 0;
    ok = ok && (ui >= 2 && vi >= 2 && ui < L.cam.w - 2 && vi < L.cam.h - 2);
    o = make_float4(1.f, 0.f, 0.f, 1.f);
    if (ok) {
      const float ip = (float)((1. / 255.) * L.prev[(size_t)(v * 4) * L.pstride + u * 4]);
      float ic, gx, gy;
      if (U8SRC) taps_u8(L.cur8, L.c8stride, uvx, uvy, ic, gx, gy);
      else ic = interp32f(L.cur, L.fstride, uvx, uvy);
      float res = ip - ic;
      if (res > 0.1) res = 0.1;
      if (res < -0.1) res = -0.1;
      float g = 1 - 50.f * res * res;
      if (g < 0.f) g = 0.f;
      o = make_float4(g, g, g, 1.f);
    }
  }
  reinterpret_cast<float4 *>(rimg + slot * rimg_b)[i] = o;
}

// computeDensePointCloudCpu (dense_tracking.cpp:393-423)
// TQ = [Ti;0 0 0 1] * Q, Ti = inverse of the pose, Q = [1 0 0 -cx; 0 1 0 -cy; 0 0 0 f; 0 0 1/b 0] (stereo_camera.cpp:24-34), evaluated with the same term order
// as a dense 4x4 product
__device__ __forceinline__ void cloud_TQ(const double (&T)[12], const svs_cam &cam, double (&TQ)[16]) {
  double Ti[12];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Ti[4 * r + c] = T[4 * c + r];
#pragma unroll
  for (int r = 0; r < 3; ++r) Ti[4 * r + 3] = -(Ti[4 * r] * T[3] + Ti[4 * r + 1] * T[7] + Ti[4 * r + 2] * T[11]);
  const double Q[16] = {1, 0, 0, -cam.cx, 0, 1, 0, -cam.cy, 0, 0, 0, cam.f, 0, 0, 1.0 / cam.b, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { double tik = r < 3 ? Ti[4 * r + k] : (k == 3 ? 1.0 : 0.0); s += tik * Q[4 * k + c]; }
      TQ[4 * r + c] = s;
    }
}
// quarter-grid sample i of one stream's level: disp = the stream's level-0 disparity image, cloud = the stream's cloud of this level
__device__ __forceinline__ void cloud_sample(const float *__restrict__ disp, int ds, int cw, int level, const double (&TQ)[16], int i, float *__restrict__ cloud) {
  const int u = i % cw, v = i / cw;
  const double inv_factor = 1.0 / (double)(1 << level);
  const float d = (float)(disp[(size_t)((v * 4) << level) * ds + ((u * 4) << level)] * inv_factor);
  float4 o;
  if (d <= 0) o = make_float4(0.f, 0.f, 0.f, -1.f);
  else {
    const double q[4] = {(double)(u * 4), (double)(v * 4), (double)d, 1.0};
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = TQ[4 * k] * q[0] + TQ[4 * k + 1] * q[1] + TQ[4 * k + 2] * q[2] + TQ[4 * k + 3] * q[3];
    o = make_float4((float)(r[0] / r[3]), (float)(r[1] / r[3]), (float)(r[2] / r[3]), 1.f);
  }
  reinterpret_cast<float4 *>(cloud)[i] = o;
}
__global__ __launch_bounds__(256) void pointcloud_cpu_sem_kernel(const float *__restrict__ disp, int ds, size_t disp_b, svs_cam cam,
                                                                 int level, const double *__restrict__ Tarr,
                                                                 float *__restrict__ cloud, size_t cloud_b) {
  const int slot = blockIdx.y;
  const int cw = cam.w / 4, ch = cam.h / 4;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cw * ch) return;
  double T[12], TQ[16];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = Tarr[(size_t)slot * 12 + k];
  cloud_TQ(T, cam, TQ);
  cloud_sample(disp + slot * disp_b, ds, cw, level, TQ, i, cloud + slot * cloud_b);
}

// the three levels' clouds of a frame in ONE launch (latency mode: three launches of 75 / 19 / 5 blocks each paid a launch gap); the same per-sample function: identical bits
struct CloudLevels { svs_cam cam[3]; float *cloud[3]; size_t cloud_b[3]; int blocks[3]; };
__global__ __launch_bounds__(256) void pointcloud_cpu_sem_levels_kernel(const float *__restrict__ disp, int ds, size_t disp_b, CloudLevels Q, const double *__restrict__ Tarr) {
  const int slot = blockIdx.y;
  int bx = blockIdx.x, level = 0;
  if (bx >= Q.blocks[0]) { bx -= Q.blocks[0]; level = 1; if (bx >= Q.blocks[1]) { bx -= Q.blocks[1]; level = 2; } }
  const svs_cam cam = level == 0 ? Q.cam[0] : (level == 1 ? Q.cam[1] : Q.cam[2]);
  float *cloud = level == 0 ? Q.cloud[0] : (level == 1 ? Q.cloud[1] : Q.cloud[2]);
  const size_t cloud_b = level == 0 ? Q.cloud_b[0] : (level == 1 ? Q.cloud_b[1] : Q.cloud_b[2]);
  const int cw = cam.w / 4, ch = cam.h / 4;
  const int i = bx * 256 + threadIdx.x;
  if (i >= cw * ch) return;
  double T[12], TQ[16];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = Tarr[(size_t)slot * 12 + k];
  cloud_TQ(T, cam, TQ);
  cloud_sample(disp + slot * disp_b, ds, cw, level, TQ, i, cloud + slot * cloud_b);
}

}  // namespace

// internal (frontend.hip): computeDensePointCloudCpu of all three levels, one launch
int svs_pointcloud_cpu_sem_levels(svs_ctx *ctx, const float *d_disp, int disp_stride, size_t disp_bstride, const svs_cam *cams, const double *d_T, float *const *d_cloud,
                                  const size_t *cloud_bstride, int batch) {
  SVS_REQUIRE(ctx, ctx && d_disp && cams && d_T && d_cloud && cloud_bstride && batch >= 1);
  SVS_DEVICE(ctx);
  CloudLevels Q;
  int total = 0;
  for (int l = 0; l < 3; ++l) {
    SVS_REQUIRE(ctx, cams[l].w % 4 == 0 && cams[l].h % 4 == 0 && d_cloud[l]);
    Q.cam[l] = cams[l]; Q.cloud[l] = d_cloud[l]; Q.cloud_b[l] = cloud_bstride[l];
    Q.blocks[l] = div_up((cams[l].w / 4) * (cams[l].h / 4), 256);
    total += Q.blocks[l];
  }
  hipLaunchKernelGGL(pointcloud_cpu_sem_levels_kernel, dim3(total, batch), dim3(256), 0, ctx->stream, d_disp, disp_stride, disp_bstride, Q, d_T);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

static int ensure_scratch(svs_ctx *ctx, double **p, size_t count) {      // ctx-owned (common.h)
  void *v = nullptr;
  const int rc = svs_ctx_scratch(ctx, count * sizeof(double), &v);
  *p = static_cast<double *>(v);
  return rc;
}

extern "C" int svs_pointcloud_cpu_sem(svs_ctx *ctx, const float *d_disp, int disp_stride, size_t disp_bstride,
                                      const svs_cam *cam, int level, const double *d_T, float *d_cloud,
                                      size_t cloud_bstride, int batch) {
  SVS_REQUIRE(ctx, ctx && d_disp && cam && d_T && d_cloud && level >= 0 && level < 3 && batch >= 1);
  SVS_DEVICE(ctx);
  SVS_REQUIRE(ctx, cam->w % 4 == 0 && cam->h % 4 == 0);            // dense_tracking.cpp:45-46 asserts
  int n = (cam->w / 4) * (cam->h / 4);
  hipLaunchKernelGGL(pointcloud_cpu_sem_kernel, dim3(div_up(n, 256), batch), dim3(256), 0, ctx->stream, d_disp, disp_stride,
                     disp_bstride, *cam, level, d_T, d_cloud, cloud_bstride);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

extern "C" int svs_dense_pass_cpu_sem(svs_ctx *ctx, const float *d_cloud, size_t cloud_bstride, const uint8_t *d_prev_u8,
                                      int pstride, size_t p_bstride, const float *d_cur, const float *d_dx,
                                      const float *d_dy, int fstride, size_t f_bstride, const svs_cam *cam,
                                      const double *d_T, int do_jac, svs_dense_sums *d_out, int batch) {
  SVS_REQUIRE(ctx, ctx && d_cloud && d_prev_u8 && d_cur && cam && d_T && d_out && batch >= 1);
  SVS_DEVICE(ctx);
  SVS_REQUIRE(ctx, !do_jac || (d_dx && d_dy));
  SVS_REQUIRE(ctx, cam->w % 4 == 0 && cam->h % 4 == 0);
  LevelArgs L{d_cloud, d_prev_u8, d_cur, d_dx, d_dy, pstride, fstride, *cam, nullptr, 0};
  int n = (cam->w / 4) * (cam->h / 4);
  int nblocks = std::min(div_up(n, 256), 256);
  double *part = nullptr;
  int rc = ensure_scratch(ctx, &part, (size_t)batch * nblocks * (NSUM + 1));
  if (rc) return rc;
  if (do_jac)
    hipLaunchKernelGGL(dense_pass_cpu_sem_kernel<true>, dim3(nblocks, batch), dim3(256), 0, ctx->stream, L, cloud_bstride, p_bstride, f_bstride, d_T, part);
  else
    hipLaunchKernelGGL(dense_pass_cpu_sem_kernel<false>, dim3(nblocks, batch), dim3(256), 0, ctx->stream, L, cloud_bstride, p_bstride, f_bstride, d_T, part);
  SVS_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(dense_finalize_kernel, dim3(batch), dim3(64), 0, ctx->stream, part, nblocks, d_out);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

namespace {
// "trk_balance": workgroups per stream from the work of the last frame, and the grid order (one workgroup, B <= 4096 streams).
//   nwg_b = min(4, ceil(w_b / M)), M = 1.2 x the mean work, raised in steps of 8 % until the extra workgroups fit x_max and the multi-workgroup streams fit
//   the resident slots (a function of the multiset of works only: streams with equal history get equal treatment, e.g. replicas of one stream);
//   grid: the multi-workgroup streams in index order, then the others by decreasing work (ties: index), then -1 (idle workgroups).
constexpr int BAL_MAX_STREAMS = 4096;
constexpr int ASSIGN_THREADS = 1024;
__global__ __launch_bounds__(ASSIGN_THREADS) void trk_assign_kernel(const svs_dense_lm_record *__restrict__ rec, int rec_cap, const int32_t *__restrict__ n_rec, int B, int slots, int x_max, int grid, float ratio, int *__restrict__ map,
                                                          unsigned char *__restrict__ nwg_of, const int *__restrict__ work16) {      // work16 (optional): the streams' work in 1/16 sweeps, left by the flat tracker kernel
  __shared__ __attribute__((aligned(16))) float s_w[BAL_MAX_STREAMS];
  __shared__ unsigned char s_n[BAL_MAX_STREAMS];
  __shared__ float s_red[ASSIGN_THREADS / 64];
  __shared__ int s_cnt[2];
  const int tid = threadIdx.x;
  // serial work of every stream's last frame in level-0 sweeps: a sweep of level l counts 4^-l (one LM record per sweep).  All records of the batch are read
  // with the whole workgroup, eight 16-byte loads in flight per lane (a lane walking its own stream's 64 records one load at a time took 80 us), and summed
  // per stream with integer LDS atomics (units of 1/16 sweep: the sum does not depend on the order)
  int *s_wi = reinterpret_cast<int *>(s_w);
  for (int b = tid; b < B; b += ASSIGN_THREADS) s_wi[b] = 0;
  __syncthreads();
  if (work16) {      // round 6: the tracker counted while it ran (one word per stream instead of 64 records per stream: 524 KB through one workgroup was half of this kernel)
    for (int b = tid; b < B; b += ASSIGN_THREADS) s_wi[b] = work16[b];
  } else if (rec) {
    const int total = B * rec_cap;
    for (int base = tid; base < total; base += 8 * ASSIGN_THREADS) {
      int4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int idx = base + u * ASSIGN_THREADS; r[u] = idx < total ? *reinterpret_cast<const int4 *>(rec + idx) : make_int4(3, 0, 0, 0); }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * ASSIGN_THREADS;
        if (idx < total) {
          const int b = idx / rec_cap, i = idx - b * rec_cap, l = r[u].x;
          if (i < min(n_rec[b], rec_cap) && l >= 0 && l <= 2) atomicAdd(&s_wi[b], l == 0 ? 16 : (l == 1 ? 4 : 1));
        }
      }
    }
  }
  __syncthreads();
  float sum = 0.f;
  for (int b = tid; b < B; b += ASSIGN_THREADS) { const float w = (float)s_wi[b] * 0.0625f; s_w[b] = w; sum += w; }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = sum;
  __syncthreads();
  float mean = 0.f;
  for (int k = 0; k < ASSIGN_THREADS / 64; ++k) mean += s_red[k];
  mean /= (float)B;
  float M = ratio * mean;
  if (x_max == 0) {      // order only ("trk_balance" = 1): every stream one workgroup -- nothing to fit (the loop below needed ~7 rounds of two barriers to find that out)
    for (int b = tid; b < B; b += ASSIGN_THREADS) s_n[b] = 1;
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
  } else
  for (int round = 0; round < 24; ++round) {
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    int extra = 0, multi = 0;
    for (int b = tid; b < B; b += ASSIGN_THREADS) {
      int n = 1;
      if (mean > 0.f) { n = (int)ceilf(s_w[b] / M); n = n < 1 ? 1 : (n > 4 ? 4 : n); }
      s_n[b] = (unsigned char)n;
      extra += n - 1; multi += n > 1 ? n : 0;
    }
    for (int o = 32; o > 0; o >>= 1) { extra += __shfl_xor(extra, o, 64); multi += __shfl_xor(multi, o, 64); }
    if ((tid & 63) == 0 && extra) { atomicAdd(&s_cnt[0], extra); atomicAdd(&s_cnt[1], multi); }
    __syncthreads();
    const bool fits = s_cnt[0] <= x_max && s_cnt[1] <= slots;
    __syncthreads();
    if (fits) break;
    M *= 1.08f;
    if (round == 23) { for (int b = tid; b < B; b += ASSIGN_THREADS) s_n[b] = 1; __syncthreads(); if (tid < 2) s_cnt[tid] = 0; __syncthreads(); }
  }
  // positions.  Single-workgroup streams: rank by (work descending, index ascending) = the number of keys greater than the own one, keys = work in 1/16 sweeps
  // << 12 | (4095 - index) packed in LDS and compared four per read (multi-workgroup streams carry key -1: they sit in front of all singles)
  const int n_multi_wgs = s_cnt[1];
  __syncthreads();
  int *s_key = reinterpret_cast<int *>(s_w);                 // the works are not needed as floats any more
  const int Bp = (B + 3) & ~3;
  for (int b = tid; b < Bp; b += ASSIGN_THREADS) {
    const int key = b < B && s_n[b] == 1 ? ((int)(s_w[b] * 16.f) << 12) | (4095 - b) : -1;      // (read and written by the same lane)
    s_key[b] = key;
  }
  for (int g = tid; g < grid; g += ASSIGN_THREADS) map[g] = -1;
  __syncthreads();
  for (int b = tid; b < B; b += ASSIGN_THREADS) {
    const int n = s_n[b];
    int pos = 0;
    if (n > 1) { for (int k = 0; k < b; ++k) pos += s_n[k] > 1 ? s_n[k] : 0; }
    else {
      const int mine = s_key[b];
      pos = n_multi_wgs;
      for (int k = 0; k < Bp; k += 4) {
        const int4 q = *reinterpret_cast<const int4 *>(s_key + k);
        pos += (q.x > mine) + (q.y > mine) + (q.z > mine) + (q.w > mine);
      }
    }
    for (int k = 0; k < n; ++k) map[pos + k] = (b << 4) | k;
    nwg_of[b] = (unsigned char)n;
  }
}

// the balanced launch of a big batch.  State (owned by the caller, frontend.hip; persistent from frame to frame): grid map [batch + batch/2] i32 | workgroups
// per stream [batch] u8 | arrival counters, failure flags, hand-over words (zero before every launch).  The assignment for the NEXT frame is made right behind
// this frame's tracker (from the LM records it leaves), so that nothing small sits between the fork of the side stream and the tracker's launch.
// scratch of the continuation launch (dense_track_batch_kernel<., ., 1>): partial sums of the shared sweeps, arrival counters + failure flags, decision words, the
// parked states and their list.  Returns G / C filled in; enqueues the memset of the words that must start at zero.
static int cont_setup(svs_ctx *ctx, int batch, TrackMulti &G, TrackCont &C) {
  const size_t n_part = (size_t)batch * 2 * CONT_MAX_NWG * 32, n_bar = (size_t)batch, n_bc = (size_t)batch * 16, n_cnt = 2, n_list = (size_t)batch / 2 + 1;
  const size_t n_park = ((size_t)batch * sizeof(TrkPark) + 7) / 8;
  double *scr = nullptr;
  const int rc = ensure_scratch(ctx, &scr, n_part + n_bar + n_bc + n_cnt + n_list + n_park);
  if (rc) return rc;
  G.part = scr;
  G.bar = reinterpret_cast<unsigned *>(scr + n_part);
  G.fail_off = batch;
  G.bcast = scr + n_part + n_bar;
  C.count = reinterpret_cast<unsigned *>(scr + n_part + n_bar + n_bc);
  C.list = reinterpret_cast<int *>(scr + n_part + n_bar + n_bc + n_cnt);
  C.park = reinterpret_cast<TrkPark *>(scr + n_part + n_bar + n_bc + n_cnt + n_list);
  C.K = ctx->trk_split; C.slots = ctx->trk_cont_slots > 0 ? ctx->trk_cont_slots : 2 * ctx->n_cu;
  SVS_HIP(ctx, hipMemsetAsync(G.bar, 0, sizeof(double) * (n_bar + n_bc + n_cnt), ctx->stream));
  return SVS_OK;
}
// the first launch of a big batch (grid: one workgroup per stream / the balanced table) and, with "trk_split", its continuation
template <bool BAL>
static int launch_batch_tracker(svs_ctx *ctx, const TrackArgs &A, bool u8src, double *d_T_io, int32_t *d_passes_out, int batch, int grid, TrackMulti G) {
  TrackCont C{nullptr, nullptr, nullptr, -1, 0};
  if (ctx->trk_split > 0) { const int rc = cont_setup(ctx, batch, G, C); if (rc) return rc; }
  if (u8src) hipLaunchKernelGGL((dense_track_batch_kernel<true, BAL, 0>), dim3(grid), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G, C);
  else hipLaunchKernelGGL((dense_track_batch_kernel<false, BAL, 0>), dim3(grid), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G, C);
  SVS_LAUNCH_CHECK(ctx);
  if (C.K < 0) return SVS_OK;
  // the parked streams, several workgroups each: they wait for each other inside the launch -- through the spin gate like every launch of that kind (common.h)
  SvsSpinScope gate(ctx);
  if (gate.rc) return gate.rc;
  if (u8src) hipLaunchKernelGGL((dense_track_batch_kernel<true, false, 1>), dim3(CONT_MAX_NWG, batch), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G, C);
  else hipLaunchKernelGGL((dense_track_batch_kernel<false, false, 1>), dim3(CONT_MAX_NWG, batch), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G, C);
  SVS_LAUNCH_CHECK(ctx);
  return gate.leave();
}
struct BalState { int *map; unsigned char *nwg_of; double *flags; size_t n_flags; int grid, x_max; int *work16; };      // work16 [batch]: each stream's LM work of the last frame in 1/16 level-0 sweeps (written by the flat tracker kernel)
BalState bal_state(void *state, int batch) {
  BalState S;
  S.x_max = batch / 2; S.grid = batch + S.x_max;
  char *p = static_cast<char *>(state);
  auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) & ~(size_t)255; return q; };
  S.map = reinterpret_cast<int *>(take(sizeof(int) * (size_t)S.grid));
  S.nwg_of = reinterpret_cast<unsigned char *>(take((size_t)batch));
  S.n_flags = (size_t)batch + (size_t)batch * 16;
  S.flags = reinterpret_cast<double *>(take(sizeof(double) * S.n_flags));
  S.work16 = reinterpret_cast<int *>(take(sizeof(int) * (size_t)batch));
  return S;
}
int bal_assign(svs_ctx *ctx, const BalState &S, int batch, const svs_dense_lm_record *rec, int rec_cap, const int32_t *n_rec, bool have_work = false) {
  const int x_max = ctx->trk_balance == 2 ? S.x_max : 0;
  const float ratio = 1.2f;
  hipLaunchKernelGGL(trk_assign_kernel, dim3(1), dim3(ASSIGN_THREADS), 0, ctx->stream, rec, rec_cap, n_rec, batch, 2 * ctx->n_cu, x_max, S.grid, ratio, S.map, S.nwg_of,
                     have_work ? S.work16 : nullptr);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
int svs_dense_track_cpu_sem_balanced(svs_ctx *ctx, const TrackArgs &A, bool u8src, double *d_T_io, int32_t *d_passes_out, int batch, void *state, const TrackMulti &G0) {
  const BalState S = bal_state(state, batch);
  double *scratch = nullptr;
  int rc = ensure_scratch(ctx, &scratch, (size_t)batch * 2 * 4 * 32);
  if (rc) return rc;
  TrackMulti G = G0;      // (the term buffers of the accept test)
  G.part = scratch;
  G.bar = reinterpret_cast<unsigned *>(S.flags);
  G.fail_off = batch;
  G.bcast = S.flags + (size_t)batch;
  G.map = S.map; G.nwg_of = S.nwg_of;
  if (ctx->trk_balance == 2) {      // order + split (experimental): sibling workgroups wait for each other -- one such launch on the device at a time (common.h)
    SVS_HIP(ctx, hipMemsetAsync(S.flags, 0, sizeof(double) * S.n_flags, ctx->stream));      // arrival counters, failure flags, hand-over words
    SvsSpinScope gate(ctx);
    if (gate.rc) return gate.rc;
    if (u8src) hipLaunchKernelGGL((dense_track_cpu_sem_kernel<true, true, TRK_MINW_BIG, false, true>), dim3(S.grid), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
    else hipLaunchKernelGGL((dense_track_cpu_sem_kernel<false, true, TRK_MINW_BIG, false, true>), dim3(S.grid), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
    SVS_LAUNCH_CHECK(ctx);
    if (int grc = gate.leave()) return grc;
  } else if (ctx->trk_flat) {       // order only: one workgroup per stream, nothing waits for anything -- the flat kernel (round 6), bit-identical to the one below
    G.work16 = S.work16;
    if (int lrc = launch_batch_tracker<true>(ctx, A, u8src, d_T_io, d_passes_out, batch, S.grid, G)) return lrc;
    return bal_assign(ctx, S, batch, A.rec, A.rec_cap, A.n_rec, true);
  } else {
    if (u8src) hipLaunchKernelGGL((dense_track_cpu_sem_kernel<true, false, TRK_MINW_BIG, false, true>), dim3(S.grid), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
    else hipLaunchKernelGGL((dense_track_cpu_sem_kernel<false, false, TRK_MINW_BIG, false, true>), dim3(S.grid), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
  }
  SVS_LAUNCH_CHECK(ctx);
  return bal_assign(ctx, S, batch, A.rec, A.rec_cap, A.n_rec);
}
}  // namespace
// the grid order the balanced launch will use for the next tracker launch, as (stream << 4) entries -- a permutation of the streams while "trk_balance" is 1
const int *svs_dense_track_balance_order(svs_ctx *ctx, void *state, int batch) {
  return state && ctx->trk_balance == 1 ? bal_state(state, batch).map : nullptr;
}
size_t svs_dense_track_balance_bytes(int batch) {
  const uintptr_t base = 1 << 16;                                  // (layout arithmetic only: nothing is dereferenced)
  const BalState S = bal_state(reinterpret_cast<void *>(base), batch);
  return (size_t)(reinterpret_cast<uintptr_t>(S.work16) - base) + sizeof(int) * (size_t)batch + 256;
}
// zero history: every stream one workgroup, streams in index order
int svs_dense_track_balance_init(svs_ctx *ctx, void *state, int batch) {
  SVS_REQUIRE(ctx, ctx && state && batch >= 1 && batch <= BAL_MAX_STREAMS);
  SVS_DEVICE(ctx);
  SVS_HIP(ctx, hipMemsetAsync(state, 0, svs_dense_track_balance_bytes(batch), ctx->stream));
  return bal_assign(ctx, bal_state(state, batch), batch, nullptr, 0, nullptr);
}
extern "C" int svs_dense_track_cpu_sem(svs_ctx *ctx, const svs_dense_track_args *a, double *d_T_io, int32_t *d_passes_out,
                                       int batch) {
  return svs_dense_track_cpu_sem_work(ctx, a, d_T_io, d_passes_out, batch, nullptr);
}
// d_bal_state != NULL (frontend.hip; svs_dense_track_balance_bytes / _init): the balanced launch of big batches
int svs_dense_track_cpu_sem_work(svs_ctx *ctx, const svs_dense_track_args *a, double *d_T_io, int32_t *d_passes_out, int batch, void *d_bal_state) {
  SVS_REQUIRE(ctx, ctx && a && d_T_io && batch >= 1);
  SVS_DEVICE(ctx);
  TrackArgs A;
  const bool u8src = a->d_cur_u8[0] != nullptr;
  for (int l = 0; l < 3; ++l) {
    SVS_REQUIRE(ctx, a->d_cloud[l] && a->d_prev_u8[l]);
    SVS_REQUIRE(ctx, u8src ? a->d_cur_u8[l] != nullptr : (a->d_cur[l] && a->d_dx[l] && a->d_dy[l]));
    SVS_REQUIRE(ctx, a->cam_vec[l].w % 4 == 0 && a->cam_vec[l].h % 4 == 0);
    A.lv[l] = LevelArgs{a->d_cloud[l], a->d_prev_u8[l], a->d_cur[l], a->d_dx[l], a->d_dy[l], a->pstride[l], a->fstride[l], a->cam_vec[l],
                        a->d_cur_u8[l], a->c8stride[l]};
    A.cloud_b[l] = a->cloud_bstride[l]; A.prev_b[l] = a->p_bstride[l]; A.f_b[l] = a->f_bstride[l]; A.c8_b[l] = a->c8_bstride[l];
  }
  A.T_jac = a->d_T_jac_out;
  A.rec = a->d_record_out; A.rec_cap = a->d_record_out ? a->record_cap : 0; A.n_rec = a->d_n_record_out;
  SVS_REQUIRE(ctx, !a->d_record_out || a->record_cap > 0);
  // latency mode: with few streams, NW workgroups share each stream's sweeps (they must all be resident: NW * batch <= 128 CUs)
  // workgroups per stream in latency mode.  Round 3 (exchange without a device-wide fence, coarsest level on one workgroup), B = 1:
  // 2 -> 0.211, 4 -> 0.165, 8 -> 0.160, 16 -> 0.184 ms per frame; round 2 (fence per sweep): 4 -> 0.26, 8 -> 0.26, 16 -> 0.28, 1 -> 0.34 ms
  int nwg = batch <= 16 ? 8 : (batch <= 32 ? 4 : 1);
  if (ctx->trk_nwg) nwg = ctx->trk_nwg;
  float *t_buf = nullptr; size_t t_b = 0;
  // the float terms of the passes (dense_tracking.cpp:229-262: `chi2 += res * res`).  "trk_seq_chi2": one buffer per stream, every sum by the sequential chain, one
  // workgroup per stream.  Default ("trk_lazy_chi2"): two buffers per stream, read only where the f64 sums cannot decide the accept test (seqsum.h)
  const bool seq_all = ctx->trk_seq_chi2 != 0, lazy = !seq_all && ctx->trk_lazy_chi2 != 0;
  float *terms = nullptr;
  if (seq_all || lazy) {
    if (seq_all) nwg = 1;
    t_b = (size_t)(a->cam_vec[0].w / 4) * (a->cam_vec[0].h / 4);
    if (lazy) t_b = ((t_b + 3) & ~(size_t)3) + 64;      // 16-byte aligned buffers, padded by more than the longest run a lane reads in one go (SEQ_RUN_MAX)
    const size_t want = (size_t)batch * t_b * sizeof(float) * (lazy ? 2 : 1) + 256;
    if (ctx->seq_buf_bytes < want) {
      if (ctx->seq_buf) { SVS_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->seq_buf); ctx->seq_buf = nullptr; ctx->seq_buf_bytes = 0; }
      SVS_HIP(ctx, hipMalloc(&ctx->seq_buf, want));
      ctx->seq_buf_bytes = want;
    }
    if (seq_all) t_buf = static_cast<float *>(ctx->seq_buf); else terms = static_cast<float *>(ctx->seq_buf);
    if (!ctx->seq_stats) {
      SVS_HIP(ctx, hipMalloc(&ctx->seq_stats, 256));
      SVS_HIP(ctx, hipMemsetAsync(ctx->seq_stats, 0, 256, ctx->stream));
    }
  }
  TrackMulti G{};
  G.terms = terms; G.terms_b = t_b; G.seq_stats = static_cast<unsigned *>(ctx->seq_stats); G.terms_only = ctx->trk_lazy_chi2 == 2;
  if (t_buf) {
    G.part = reinterpret_cast<double *>(t_buf); G.fail_off = (int)t_b;
    if (u8src) hipLaunchKernelGGL((dense_track_cpu_sem_kernel<true, false, 2, true>), dim3(batch), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
    else hipLaunchKernelGGL((dense_track_cpu_sem_kernel<false, false, 2, true>), dim3(batch), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
  } else if (nwg >= 2) {
    double *scratch = nullptr;
    const size_t n_part = (size_t)batch * 2 * nwg * 64;      // per stream, parity and workgroup: 58 (+ 6) flagged 8-byte words (all_workgroups)
    // (the cleared size rounded up to 64 bytes: a clear whose size is no multiple of 16 bytes is TWO fill kernels in the runtime -- an aligned bulk and a tail --
    //  and in latency mode every launch in front of the tracker is on the frame's critical path)
    const size_t n_clear = (n_part + (size_t)batch + (size_t)batch * 16 + 7) & ~(size_t)7;
    int rc = ensure_scratch(ctx, &scratch, n_clear);      // + one counter word and one failure flag (4 + 4 bytes) per stream + the hand-over words
    if (rc) return rc;
    G.part = scratch;
    G.bar = reinterpret_cast<unsigned *>(scratch + n_part);
    G.fail_off = batch;
    G.bcast = scratch + n_part + (size_t)batch;
    SVS_HIP(ctx, hipMemsetAsync(scratch, 0, sizeof(double) * n_clear, ctx->stream));      // flagged words, failure flags, hand-over words: one clear
    SvsSpinScope gate(ctx, nwg * batch);      // the workgroups of a stream wait for each other: one such launch on the device at a time (common.h); one stream: priority lane
    if (gate.rc) return gate.rc;
    if (u8src) hipLaunchKernelGGL((dense_track_cpu_sem_kernel<true, true, 2>), dim3(nwg, batch), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
    else hipLaunchKernelGGL((dense_track_cpu_sem_kernel<false, true, 2>), dim3(nwg, batch), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
    SVS_LAUNCH_CHECK(ctx);
    if (int grc = gate.leave()) return grc;
  } else if (d_bal_state && ctx->trk_balance && batch >= 2 * ctx->n_cu && batch <= BAL_MAX_STREAMS && A.rec && A.n_rec) {
    return svs_dense_track_cpu_sem_balanced(ctx, A, u8src, d_T_io, d_passes_out, batch, d_bal_state, G);
  } else if (((batch > ctx->n_cu && ctx->trk_regs != 1) || ctx->trk_regs == 2) && ctx->trk_flat) {
    if (int lrc = launch_batch_tracker<false>(ctx, A, u8src, d_T_io, d_passes_out, batch, batch, G)) return lrc;
  } else if ((batch > ctx->n_cu && ctx->trk_regs != 1) || ctx->trk_regs == 2) {      // trk_regs: tests / experiments, latched at svs_ctx_create
    if (u8src) hipLaunchKernelGGL((dense_track_cpu_sem_kernel<true, false, TRK_MINW_BIG>), dim3(batch), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
    else hipLaunchKernelGGL((dense_track_cpu_sem_kernel<false, false, TRK_MINW_BIG>), dim3(batch), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
  } else {
    if (u8src) hipLaunchKernelGGL((dense_track_cpu_sem_kernel<true, false, 2>), dim3(batch), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
    else hipLaunchKernelGGL((dense_track_cpu_sem_kernel<false, false, 2>), dim3(batch), dim3(TRK_THREADS), 0, ctx->stream, A, d_T_io, d_passes_out, G);
  }
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

// diagnostic: the sum of the reference's `float chi2 += t[i]` over batch rows of n terms, formed (how = 0) as the tracker's accept test forms it (exact_seq_sum_f32),
// (how = 1) the same with the terms read past the caches as the latency-mode tracker reads them, (how = 2) by the literal sequential chain
namespace {
template <int HOW>
__global__ __launch_bounds__(TRK_THREADS) void seq_sum_probe_kernel(const float *__restrict__ t, int n, size_t bstride, float *__restrict__ out, int *__restrict__ fell_back) {
  if (threadIdx.x == 0) g_seq_sh.fell_back = 0;
  __syncthreads();
  const float *row = t + (size_t)blockIdx.x * bstride;
  float v;
  if constexpr (HOW == 2) v = seq_chi2_f32(row, n); else v = exact_seq_sum_f32<HOW == 1>(row, n);
  if (threadIdx.x == 0) { out[blockIdx.x] = v; if (fell_back) fell_back[blockIdx.x] = HOW == 2 ? 1 : g_seq_sh.fell_back; }
}
}  // namespace
extern "C" int svs_dense_seq_sum_f32(svs_ctx *ctx, const float *d_terms, int n, size_t bstride, int batch, int how, float *d_out, int32_t *d_fell_back) {
  SVS_REQUIRE(ctx, ctx && d_terms && d_out && n >= 0 && batch >= 1 && how >= 0 && how <= 2);
  SVS_DEVICE(ctx);
  if (how == 0) hipLaunchKernelGGL(seq_sum_probe_kernel<0>, dim3(batch), dim3(TRK_THREADS), 0, ctx->stream, d_terms, n, bstride, d_out, d_fell_back);
  else if (how == 1) hipLaunchKernelGGL(seq_sum_probe_kernel<1>, dim3(batch), dim3(TRK_THREADS), 0, ctx->stream, d_terms, n, bstride, d_out, d_fell_back);
  else hipLaunchKernelGGL(seq_sum_probe_kernel<2>, dim3(batch), dim3(TRK_THREADS), 0, ctx->stream, d_terms, n, bstride, d_out, d_fell_back);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

extern "C" int svs_dense_residual_image_cpu_sem(svs_ctx *ctx, const float *d_cloud, size_t cloud_bstride, const uint8_t *d_prev_u8,
                                                int pstride, size_t p_bstride, const float *d_cur, int fstride, size_t f_bstride,
                                                const uint8_t *d_cur_u8, int c8stride, size_t c8_bstride, const svs_cam *cam,
                                                const double *d_T, size_t T_bstride, float *d_res_img4, size_t res_bstride, int batch) {
  SVS_REQUIRE(ctx, ctx && d_cloud && d_prev_u8 && (d_cur || d_cur_u8) && cam && d_T && d_res_img4 && batch >= 1);
  SVS_DEVICE(ctx);
  SVS_REQUIRE(ctx, cam->w % 4 == 0 && cam->h % 4 == 0);
  LevelArgs L{d_cloud, d_prev_u8, d_cur, nullptr, nullptr, pstride, fstride, *cam, d_cur_u8, c8stride};
  const int n = (cam->w / 4) * (cam->h / 4);
  if (d_cur_u8)
    hipLaunchKernelGGL(residual_image_cpu_sem_kernel<true>, dim3(div_up(n, 256), batch), dim3(256), 0, ctx->stream, L, cloud_bstride, p_bstride,
                       f_bstride, c8_bstride, d_T, T_bstride, d_res_img4, res_bstride);
  else
    hipLaunchKernelGGL(residual_image_cpu_sem_kernel<false>, dim3(div_up(n, 256), batch), dim3(256), 0, ctx->stream, L, cloud_bstride, p_bstride,
                       f_bstride, c8_bstride, d_T, T_bstride, d_res_img4, res_bstride);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

// ---- motion-only pose refinement: PoseOptimizer<SE3,6,IdObs<3>,3>::calcFastMotionOnly ---------------------------
// (pose_optimizer.h:134-298, called at stereo_frontend.cpp:1058-1063 right behind the guided matcher).  One workgroup
// per camera stream runs the whole LM loop on the device: observations are the status-OK entries of the matcher's
// result array (= TrackData::obs_list / point_list, in list order); every pass is a strided sweep over them with a
// wave-shuffle + LDS reduction of the 21 + 6 normal-equation sums (or chi2 / max error); one lane solves the 6x6 system.
namespace {

__device__ __forceinline__ double mo_kernel(double delta, double b) {      // pseudo-Huber cost, pose_optimizer.h:426-435
  const double a = fabs(delta);
  return a < b ? delta * delta : 2 * b * a - b * b;
}
// f = obs - map_uvu(T xyz) (stereo_camera.cpp:37-44); J = SE3XYZ_STEREO::frameJac (transformations.h:424-447) if wanted
template <bool JAC>
__device__ __forceinline__ void mo_residual(const double *T, const svs_match_result &o, const svs_cam &cam, double *f, double *J) {
  const double *q = o.xyz_actkey;
  const double x = T[0] * q[0] + T[1] * q[1] + T[2] * q[2] + T[3];
  const double y = T[4] * q[0] + T[5] * q[1] + T[6] * q[2] + T[7];
  const double z = T[8] * q[0] + T[9] * q[1] + T[10] * q[2] + T[11];
  const double fl = cam.f;
  f[0] = o.obs[0] - (x / z * fl + cam.cx);
  f[1] = o.obs[1] - (y / z * fl + cam.cy);
  f[2] = o.obs[2] - ((x - cam.b) / z * fl + cam.cx);
  if (JAC) {
    const double ibz = 1. / z, ibz2 = 1. / (z * z);
    const double A = -fl * ibz, B = -fl * ibz, C = fl * x * ibz2, D = fl * y * ibz2, E = fl * (x - cam.b) * ibz2;
    J[0] = A; J[1] = 0; J[2] = C; J[3] = y * C; J[4] = z * A - x * C; J[5] = -y * A;
    J[6] = 0; J[7] = B; J[8] = D; J[9] = -z * B + y * D; J[10] = -x * D; J[11] = x * B;
    J[12] = A; J[13] = 0; J[14] = E; J[15] = y * E; J[16] = z * A - x * E; J[17] = -y * A;
  }
}
__device__ __forceinline__ double mo_weighted_sq(double *f, int robust, double b) {
  if (robust) {
    const double nrm = fmax(1e-10, sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]));
    const double w = sqrt(mo_kernel(nrm, b)) / nrm;
    f[0] *= w; f[1] *= w; f[2] *= w;
  }
  return f[0] * f[0] + f[1] * f[1] + f[2] * f[2];
}

constexpr int MO_THREADS = 256;
// block reduction of N doubles per thread: sums for k < n_sum, maxima for the rest.  Result in s_red[0..N) (all threads may read)
template <int N>
__device__ __forceinline__ void mo_reduce(double (&v)[N], int n_sum, double *s_red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double x = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(x, o, 64); x = k < n_sum ? x + y : fmax(x, y); }
    if (lane == 0) s_red[wave * N + k] = x;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    const int k = threadIdx.x;
    double x = s_red[k];
    for (int w = 1; w < MO_THREADS / 64; ++w) x = k < n_sum ? x + s_red[w * N + k] : fmax(x, s_red[w * N + k]);
    s_red[(MO_THREADS / 64) * N + k] = x;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = s_red[(MO_THREADS / 64) * N + k];
  __syncthreads();
}

__global__ __launch_bounds__(MO_THREADS) void motion_only_kernel(const svs_match_result *__restrict__ res, int n, size_t res_bstride, svs_cam cam,
                                                                 svs_pose_opt_params prm, double *__restrict__ T_io, svs_pose_opt_stats *__restrict__ stats) {
  __shared__ double s_red[(MO_THREADS / 64 + 1) * 28];
  __shared__ double s_T[12], s_Tn[12], s_B[6];
  const int tid = threadIdx.x, slot = blockIdx.x;
  res += (size_t)slot * res_bstride;
  if (tid < 12) s_T[tid] = T_io[12 * slot + tid];
  __syncthreads();
  // initial residuals: chi2, max error, number of observations, max diag(J^T J)
  double chi2, max_err, mu, nu = 2;
  int num_obs;
  {
    double v[4] = {0, 0, 0, 0};      // chi2, num_obs | max_err, norm_max_A
    for (int i = tid; i < n; i += MO_THREADS) {
      if (res[i].status != 0) continue;
      double f[3], J[18];
      mo_residual<true>(s_T, res[i], cam, f, J);
#pragma unroll
      for (int c = 0; c < 6; ++c) v[3] = fmax(v[3], fabs(J[c] * J[c] + J[6 + c] * J[6 + c] + J[12 + c] * J[12 + c]));
      v[0] += mo_weighted_sq(f, prm.robust_kernel, prm.kernel_param);
      v[1] += 1.0;
      v[2] = fmax(v[2], fmax(fabs(f[0]), fmax(fabs(f[1]), fabs(f[2]))));
    }
    mo_reduce<4>(v, 2, s_red);
    chi2 = v[0]; num_obs = (int)v[1]; max_err = v[2];
    mu = prm.initial_mu == -1 ? prm.tau * v[3] : prm.initial_mu;
  }
  const double initial_chi2 = chi2;
  int status = num_obs == 0 ? 1 : (num_obs < prm.min_obs ? 3 : 0), trial = 0;
  bool stop = status != 0;
  for (int ig = 0; ig < prm.num_iter && !stop; ++ig) {
    double rho = 0;
    do {
      double v[27];      // 21 unique of sum J^T J (upper, row-major), 6 of sum J^T (w f)
#pragma unroll
      for (int k = 0; k < 27; ++k) v[k] = 0;
      for (int i = tid; i < n; i += MO_THREADS) {
        if (res[i].status != 0) continue;
        double f[3], J[18];
        mo_residual<true>(s_T, res[i], cam, f, J);
        mo_weighted_sq(f, prm.robust_kernel, prm.kernel_param);
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
          for (int c = r; c < 6; ++c) v[k++] += J[r] * J[c] + J[6 + r] * J[6 + c] + J[12 + r] * J[12 + c];
          v[21 + r] += J[r] * f[0] + J[6 + r] * f[1] + J[12 + r] * f[2];
        }
      }
      mo_reduce<27>(v, 27, s_red);
      if (tid == 0) {
        double A[36], B[6], delta[6];
        int k = 0;
        for (int r = 0; r < 6; ++r)
          for (int c = r; c < 6; ++c) { A[6 * r + c] = A[6 * c + r] = v[k++]; }
        for (int r = 0; r < 6; ++r) { A[7 * r] += mu; B[r] = -v[21 + r]; s_B[r] = B[r]; }
        d_solve6(A, B, delta);                                 // A.ldlt().solve(B)
        double Tn[12];
        d_se3_exp_mul(delta, s_T, Tn);                         // prediction.add: exp(delta) * T
        for (int i = 0; i < 12; ++i) s_Tn[i] = Tn[i];
      }
      __syncthreads();
      double w[2] = {0, 0};      // new chi2 | new max error
      for (int i = tid; i < n; i += MO_THREADS) {
        if (res[i].status != 0) continue;
        double f[3];
        mo_residual<false>(s_Tn, res[i], cam, f, nullptr);
        w[0] += mo_weighted_sq(f, prm.robust_kernel, prm.kernel_param);
        w[1] = fmax(w[1], fmax(fabs(f[0]), fmax(fabs(f[1]), fabs(f[2]))));
      }
      mo_reduce<2>(w, 1, s_red);
      const double new_chi2 = w[0];
      if (isnan(new_chi2)) { status = 2; stop = true; break; }      // the reference throws here
      rho = chi2 - new_chi2;
      if (rho > 0) {
        if (tid < 12) s_T[tid] = s_Tn[tid];
        chi2 = new_chi2; max_err = w[1];
        double bm = -1;
        for (int c = 0; c < 6; ++c) bm = fmax(bm, fabs(s_B[c]));
        stop = bm <= 1e-10;
        const double q = 2 * rho - 1, sc = 1 - q * q * q;
        mu *= fmax(1. / 3., sc);
        nu = 2.; trial = 0;
      } else {
        mu *= nu; nu *= 2.; ++trial;
        if (trial == 5) stop = true;
      }
      __syncthreads();
    } while (!(rho > 0 || stop));
  }
  if (tid < 12) T_io[12 * slot + tid] = s_T[tid];
  if (tid == 0) {
    svs_pose_opt_stats st;
    st.initial_chi2 = initial_chi2; st.chi2 = chi2; st.max_err = max_err; st.num_obs = num_obs; st.status = status;
    stats[slot] = st;
  }
}



// ---- the same loop, restructured for latency (round 3) ---------------------------------------------------------------------------
// The kernel above walks the matcher's record array (status test + 64-byte record from global memory per observation) twice per LM
// trial and solves on one lane: 202 us for ~850 observations at B = 1.  Here:
//  * the status-OK records are compacted ONCE (order-preserving: ballots + a wave prefix) into an index list in LDS, and every thread keeps
//    up to MO2_RC observations in registers for the whole loop (observations beyond MO2_RC * MO2_THREADS are re-read through the list);
//  * ONE sweep per LM trial: chi2 / max error at the trial pose and the normal equations at the same pose share their residuals, so the
//    reference's "new chi2" pass and the next iteration's J^T J pass are one (a rejected trial keeps the stored system and only re-damps);
//  * 28 sums per wave by recursive halving (32 exchanges instead of 168 butterflies), two block barriers per trial;
//  * the 6x6 solve runs on seven lanes of wave 0 (lane = column of [A + mu I | B], Gauss-Jordan with readlane broadcasts; the matrix is
//    symmetric positive definite, so no pivoting is needed where the reference's ldlt() pivots), exp(delta) * T on the same wave.
// Same LM schedule and stopping rules as pose_optimizer.h:134-298; sums in a different order and one reciprocal instead of five divisions
// per residual => the pose agrees with the oracle to ~1e-12 (test bar 1e-9).
// Round 5: 256 lanes and six observations per lane in registers (rounds 3-4: 512 and four).  The kernel needs ~235 vector registers whatever its width, i.e. two
// waves per SIMD: a 512-lane workgroup had a CU to itself and a batch of 512 streams ran as two rounds of a latency-bound loop (<= 15 dependent LM iterations, two
// barriers and a 6 x 6 solve on one wave each); two 256-lane workgroups share a CU and the serial parts of one stream hide behind the sweeps of the other:
// 0.32 -> 0.22 ms per 512 streams.
#ifndef SVS_MO2_THREADS
#define SVS_MO2_THREADS 256
#endif
#ifndef SVS_MO2_RC
#define SVS_MO2_RC 6
#endif
constexpr int MO2_THREADS = SVS_MO2_THREADS, MO2_RC = SVS_MO2_RC, MO2_WAVES = MO2_THREADS / 64;
struct MoObs { double o[3], q[3]; };

template <bool FIRST, bool JAC = true>      // JAC = false: chi2 and max error only (the sweep of a trial that is expected to be rejected)
__device__ __forceinline__ void mo2_terms(const double (&T)[12], const MoObs &ob, const svs_cam &cam, int robust, double kb, double (&x)[32], double &max_err,
                                          double &max_diag) {
  static_assert(JAC || !FIRST, "the first sweep needs the Jacobian");
  const double *q = ob.q;
  const double X = T[0] * q[0] + T[1] * q[1] + T[2] * q[2] + T[3];
  const double Y = T[4] * q[0] + T[5] * q[1] + T[6] * q[2] + T[7];
  const double Z = T[8] * q[0] + T[9] * q[1] + T[10] * q[2] + T[11];
  const double fl = cam.f, iz = 1.0 / Z, fiz = fl * iz, Xb = X - cam.b;
  double f0 = ob.o[0] - (X * fiz + cam.cx), f1 = ob.o[1] - (Y * fiz + cam.cy), f2 = ob.o[2] - (Xb * fiz + cam.cx);
  // frameJac (transformations.h:424-447): rows (A 0 C yC zA-xC -yA), (0 A D -zA+yD -xD xA), (A 0 E yE zA-xE -yA) with A = -f/z, C = f x/z^2, ...
  const double A = -fiz, C = fiz * X * iz, D = fiz * Y * iz, E = fiz * Xb * iz;
  const double a3 = Y * C, a4 = Z * A - X * C, a5 = -Y * A;      // row 0, columns 3..5
  const double b3 = Y * D - Z * A, b4 = -X * D, b5 = X * A;      // row 1
  const double c3 = Y * E, c4 = Z * A - X * E;                   // row 2 (column 5 = a5)
  if (FIRST) {
    const double AA = A * A;
    double m = fmax(AA + AA, AA);
    m = fmax(m, C * C + D * D + E * E);
    m = fmax(m, a3 * a3 + b3 * b3 + c3 * c3);
    m = fmax(m, a4 * a4 + b4 * b4 + c4 * c4);
    m = fmax(m, a5 * a5 + b5 * b5 + a5 * a5);
    max_diag = fmax(max_diag, m);
  }
  // pseudo-Huber weight (pose_optimizer.h:169-175,426-435): f *= sqrt(k(|f|)) / |f|, chi2 += |w f|^2 = k(|f|)
  const double ss = f0 * f0 + f1 * f1 + f2 * f2;
  double chi = ss;
  if (robust) {
    const double nrm = fmax(1e-10, sqrt(ss));
    if (nrm >= kb) {                                             // outliers only: inside the kernel width the weight is sqrt(n^2) / n = 1
      const double k = 2 * kb * nrm - kb * kb, w = sqrt(k) / nrm;
      f0 *= w; f1 *= w; f2 *= w;
      chi = f0 * f0 + f1 * f1 + f2 * f2;
    }
  }
  x[27] += chi;
  max_err = fmax(max_err, fmax(fabs(f0), fmax(fabs(f1), fabs(f2))));
  if constexpr (!JAC) return;
  // J^T J, upper triangle row-major (21) -- the structural zeros of frameJac are not multiplied out -- and J^T (w f) (6)
  x[0] += A * A + A * A;          x[1] += 0.0;                     x[2] += A * C + A * E;
  x[3] += A * a3 + A * c3;        x[4] += A * a4 + A * c4;         x[5] += A * a5 + A * a5;
  x[6] += A * A;                  x[7] += A * D;                   x[8] += A * b3;
  x[9] += A * b4;                 x[10] += A * b5;
  x[11] += C * C + D * D + E * E; x[12] += C * a3 + D * b3 + E * c3; x[13] += C * a4 + D * b4 + E * c4;
  x[14] += C * a5 + D * b5 + E * a5;
  x[15] += a3 * a3 + b3 * b3 + c3 * c3; x[16] += a3 * a4 + b3 * b4 + c3 * c4; x[17] += a3 * a5 + b3 * b5 + c3 * a5;
  x[18] += a4 * a4 + b4 * b4 + c4 * c4; x[19] += a4 * a5 + b4 * b5 + c4 * a5;
  x[20] += a5 * a5 + b5 * b5 + a5 * a5;
  x[21] += A * f0 + A * f2;       x[22] += A * f1;                 x[23] += C * f0 + D * f1 + E * f2;
  x[24] += a3 * f0 + b3 * f1 + c3 * f2; x[25] += a4 * f0 + b4 * f1 + c4 * f2; x[26] += a5 * f0 + b5 * f1 + a5 * f2;
}
// value id held by lane `lane` (< 32) after the recursive halving of mo2_wave_reduce
__device__ __forceinline__ int mo2_id(int lane) { return ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4); }
__device__ __forceinline__ double mo2_wave_reduce(double (&x)[32]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int step = 0; step < 5; ++step) {
    const int half = 16 >> step, bit = 1 << step;
    const bool up = (lane & bit) != 0;
#pragma unroll
    for (int k = 0; k < half; ++k) {
      // the two candidates are pinned as VALUES: left alone, the optimiser turns "select of two array elements" into one load at a selected address
      // before the loops are unrolled -- and the accumulator array then lives (and is stored after every observation) in scratch memory
      double lo = x[k], hi = x[k + half];
      asm volatile("" : "+v"(lo), "+v"(hi));
      const double send = up ? lo : hi;
      const double keep = up ? hi : lo;
      x[k] = keep + __shfl_xor(send, bit, 64);
    }
  }
  return x[0] + __shfl_xor(x[0], 32, 64);
}

// one sweep over the observations at pose T: per-wave sums -> s_part[wave]
template <bool FIRST>
__device__ __forceinline__ void mo2_sweep(const double (&T)[12], const MoObs (&ob)[MO2_RC], int n_ok, const int *s_idx, const svs_match_result *__restrict__ res,
                                          const svs_cam &cam, int robust, double kb, double (*s_part)[32]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = 0.0;
  double me = 0.0, md = 0.0;
#pragma unroll
  for (int k = 0; k < MO2_RC; ++k)
    if (tid + k * MO2_THREADS < n_ok) mo2_terms<FIRST>(T, ob[k], cam, robust, kb, x, me, md);
  for (int j = tid + MO2_RC * MO2_THREADS; j < n_ok; j += MO2_THREADS) {
    const svs_match_result &r = res[s_idx[j]];
    MoObs t;
#pragma unroll
    for (int c = 0; c < 3; ++c) { t.o[c] = r.obs[c]; t.q[c] = r.xyz_actkey[c]; }
    mo2_terms<FIRST>(T, t, cam, robust, kb, x, me, md);
  }
  const double v = mo2_wave_reduce(x);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { me = fmax(me, __shfl_xor(me, o, 64)); if (FIRST) md = fmax(md, __shfl_xor(md, o, 64)); }
  const int id = mo2_id(lane);
  if (lane < 32 && id < 28) s_part[wave][id] = v;
  if (lane == 0) { s_part[wave][28] = me; s_part[wave][29] = md; }
}

// chi2 (and the largest residual) at pose T only -> s_part[wave][27], [28].  Bit-identical to what mo2_sweep leaves there: the same per-lane terms in the same
// order, and the butterfly below adds them in the association mo2_wave_reduce's halving gives element 27 (own + partner over lane distances 1, 2, 4, 8, 16, 32).
__device__ __forceinline__ void mo2_sweep_chi2(const double (&T)[12], const MoObs (&ob)[MO2_RC], int n_ok, const int *s_idx, const svs_match_result *__restrict__ res,
                                               const svs_cam &cam, int robust, double kb, double (*s_part)[32]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double x[32];
  x[27] = 0.0;
  double me = 0.0, md = 0.0;
#pragma unroll
  for (int k = 0; k < MO2_RC; ++k)
    if (tid + k * MO2_THREADS < n_ok) mo2_terms<false, false>(T, ob[k], cam, robust, kb, x, me, md);
  for (int j = tid + MO2_RC * MO2_THREADS; j < n_ok; j += MO2_THREADS) {
    const svs_match_result &r = res[s_idx[j]];
    MoObs t;
#pragma unroll
    for (int c = 0; c < 3; ++c) { t.o[c] = r.obs[c]; t.q[c] = r.xyz_actkey[c]; }
    mo2_terms<false, false>(T, t, cam, robust, kb, x, me, md);
  }
  double v = x[27];
#pragma unroll
  for (int o = 1; o <= 32; o <<= 1) v += __shfl_xor(v, o, 64);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) me = fmax(me, __shfl_xor(me, o, 64));
  if (lane == 0) { s_part[wave][27] = v; s_part[wave][28] = me; }
}

// the gate of one stream, run by 256 lanes (tid < 256) of a workgroup: s_cnt [18], s_sum [4] in LDS; the caller has a barrier in front of it
__device__ __forceinline__ void gate_stream(const svs_match_result *__restrict__ res, const svs_candidate_point *__restrict__ pts, int n, int n_new, const svs_cam &cam,
                                            const double (&T)[12], float mre, svs_gated_point *__restrict__ out, svs_point_stats *__restrict__ st, int tid, int *s_cnt,
                                            double *s_sum) {
  if (tid < 18) s_cnt[tid] = 0;
  __syncthreads();
  const int half_w = (int)(cam.w * 0.5), half_h = (int)(cam.h * 0.5);
  const float third = (float)(1. / 3.);
  const int third_w = (int)(cam.w * third), third_h = (int)(cam.h * third);
  const int tt_w = (int)(cam.w * 2 * third), tt_h = (int)(cam.h * 2 * third);
  double len = 0;
  for (int i = tid; i < n && tid < 256; i += 256) {
    svs_gated_point g{};
    if (res[i].status == 0) {
      atomicAdd(&s_cnt[17], 1);
      double d[3];
      mo_residual<false>(T, res[i], cam, d, nullptr);          // uvu - se3xyz_stereo_.map(T_cur_from_actkey_, point)
      const int level = pts[i].anchor_level;
      const int factor = 1 << level;                             // zeroFromPyr_i(1, anchor_level)
      if (fabs(d[0]) < mre * factor && fabs(d[1]) < mre * factor && fabs(d[2]) < 3. * mre) {
        const double *uvu = res[i].obs, *q = res[i].xyz_actkey;
        const int i2 = uvu[0] < half_w ? 0 : 1, j2 = uvu[1] < half_h ? 0 : 1;
        const int i3 = uvu[0] < third_w ? 0 : (uvu[0] < tt_w ? 1 : 2), j3 = uvu[1] < third_h ? 0 : (uvu[1] < tt_h ? 1 : 2);
        atomicAdd(&s_cnt[i2 * 2 + j2], 1);
        atomicAdd(&s_cnt[4 + i3 * 3 + j3], 1);
        atomicAdd(&s_cnt[13 + level], 1);
        atomicAdd(&s_cnt[16], 1);
        const double inv = 1.0 / (double)factor;                 // exact power of two: x * inv == x / factor
        g.accepted = 1;
        g.is_new = i < n_new ? 1 : 0;
        g.uv_pyr[0] = uvu[0] * inv; g.uv_pyr[1] = uvu[1] * inv;
        g.curkey_uv_pyr[0] = (q[0] / q[2] * cam.f + cam.cx) * inv;
        g.curkey_uv_pyr[1] = (q[1] / q[2] * cam.f + cam.cy) * inv;
        const double dx = g.uv_pyr[0] - g.curkey_uv_pyr[0], dy = g.uv_pyr[1] - g.curkey_uv_pyr[1];
        len += sqrt(dx * dx + dy * dy);
      }
    }
    out[i] = g;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) len += __shfl_xor(len, o, 64);
  if ((tid & 63) == 0 && tid < 256) s_sum[tid >> 6] = len;
  __syncthreads();
  if (tid < 4) st->num_points_grid2x2[tid] = s_cnt[tid];
  else if (tid < 13) st->num_points_grid3x3[tid - 4] = s_cnt[tid];
  else if (tid < 16) st->num_matched_points[tid - 13] = s_cnt[tid];
  else if (tid == 16) st->num_track_points = s_cnt[16];
  else if (tid == 17) st->num_obs = s_cnt[17];
  else if (tid == 18) { st->pad_[0] = st->pad_[1] = 0; st->sum_track_length = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]); }
}
// TAIL: processMatchedPoints' gate and the three dense clouds of a stream run at the end of its refinement workgroup instead of in four launches of their own --
// the streams that converge early do that work while the slow ones still iterate (the stage lasts as long as its slowest stream: <= 15 dependent iterations),
// and four launches per frame disappear.  Same device functions as the stand-alone kernels, the gate on the same 256 lanes: identical bits.
using MoTail = svs_mo_tail;      // common.h
template <bool TAIL>
__global__ __launch_bounds__(MO2_THREADS) void motion_only_fused_kernel(const svs_match_result *__restrict__ res, int n, size_t res_bstride, svs_cam cam,
                                                                        svs_pose_opt_params prm, double *__restrict__ T_io,
                                                                        svs_pose_opt_stats *__restrict__ stats, MoTail Q, int spec) {
  extern __shared__ int s_idx[];                   // [n]: indices of the status-OK records, in list order
  __shared__ double s_part[MO2_WAVES][32];         // per wave: 28 sums (21 of J^T J, 6 of J^T w f, chi2), max error, max diag
  __shared__ double s_Tn[12], s_Tc[12];        // trial pose / accepted pose (the latter is wave 0's)
  __shared__ int s_wcnt[8][MO2_WAVES];
  __shared__ int s_conv;                           // |B|_inf <= 1e-10 at the pose the pending step was taken from
  // TAIL (batches): the streams in the grid order of the tracker (by the last frame's LM work: replicas / neighbours of the caller's order dealt evenly over the XCDs)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, slot = TAIL && Q.order ? (Q.order[blockIdx.x] >> 4) : (int)blockIdx.x;
  res += (size_t)slot * res_bstride;
  // ---- compaction of obs_list / point_list (the OK records, in order).  Eight chunks per round: their status words are requested together
  // (one memory round trip instead of eight), one barrier per round
  int n_ok = 0;
  for (int base = 0; base < n; base += 8 * MO2_THREADS) {
    int st[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { const int i = base + c * MO2_THREADS + tid; st[c] = i < n ? res[i].status : 1; }
    unsigned long long m[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { m[c] = __ballot(st[c] == 0); if (lane == 0) s_wcnt[c][wave] = __popcll(m[c]); }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int off = n_ok, tot = 0;
#pragma unroll
      for (int w = 0; w < MO2_WAVES; ++w) { const int k = s_wcnt[c][w]; off += w < wave ? k : 0; tot += k; }
      if (st[c] == 0) s_idx[off + __popcll(m[c] & ((1ull << lane) - 1ull))] = base + c * MO2_THREADS + tid;
      n_ok += tot;
    }
    __syncthreads();
  }
  MoObs ob[MO2_RC];
#pragma unroll
  for (int k = 0; k < MO2_RC; ++k) {
    const int j = tid + k * MO2_THREADS;
    if (j < n_ok) {
      const svs_match_result &r = res[s_idx[j]];
#pragma unroll
      for (int c = 0; c < 3; ++c) { ob[k].o[c] = r.obs[c]; ob[k].q[c] = r.xyz_actkey[c]; }
    }
  }
  auto total = [&](int id) { double s = s_part[0][id]; for (int w = 1; w < MO2_WAVES; ++w) s += s_part[w][id]; return s; };
  auto total_max = [&](int id) { double s = s_part[0][id]; for (int w = 1; w < MO2_WAVES; ++w) s = fmax(s, s_part[w][id]); return s; };
  // the pose of a sweep is wave-uniform: read once, moved to scalar registers (24 SGPRs instead of 24 VGPRs per lane)
  auto uniform_pose = [&](const double *p, double (&T)[12]) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const double v = p[i];
      T[i] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
  };
  double Tn[12];
  uniform_pose(T_io + 12 * slot, Tn);
  if (tid < 12) s_Tc[tid] = T_io[12 * slot + tid];
  mo2_sweep<true>(Tn, ob, n_ok, s_idx, res, cam, prm.robust_kernel, prm.kernel_param, s_part);
  __syncthreads();                                 // (A) sums of the sweep are in s_part
  const int num_obs = n_ok;
  double chi2 = total(27), max_err = total_max(28), mu = prm.initial_mu == -1 ? prm.tau * total_max(29) : prm.initial_mu, nu = 2;
  const double initial_chi2 = chi2;
  int status = num_obs == 0 ? 1 : (num_obs < prm.min_obs ? 3 : 0), trial = 0;
  bool stop = status != 0;
  // wave 0, lane c < 6: column c of the stored J^T J; lane 6: -J^T w f
  double col[6] = {0, 0, 0, 0, 0, 0};
  auto load_system = [&]() {
    if (wave == 0 && lane < 7) {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const int a = r < lane ? r : lane, b = r < lane ? lane : r;      // upper-triangular index of (r, lane)
        col[r] = lane < 6 ? total(a * (13 - a) / 2 + (b - a)) : -total(21 + r);
      }
    }
  };
  load_system();
  // Two barriers per LM trial: (B) the trial pose is in s_Tn and every wave has read the sums of the sweep before; (A) the new sums are in s_part.
  for (int ig = 0; ig < prm.num_iter && !stop; ++ig) {
    double rho = 0;
    do {
      if (wave == 0) {
        // (A + mu I) delta = B on lanes 0..6 (wave_solve6)
        double a[6], bmax = 0, delta[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) a[r] = col[r] + (r == lane ? mu : 0.0);
#pragma unroll
        for (int r = 0; r < 6; ++r) bmax = fmax(bmax, fabs(mo2_bcast(col[r], 6)));
        wave_solve6(a, delta);
        double Tc[12], Tx[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) Tc[i] = s_Tc[i];
        mo2_exp_mul(delta, Tc, Tx);                  // prediction.add: exp(delta) * T (every lane of the wave, same values)
        if (lane < 12) {
          double v = Tx[0];
#pragma unroll
          for (int i = 1; i < 12; ++i) v = lane == i ? Tx[i] : v;
          s_Tn[lane] = v;
        }
        if (lane == 0) s_conv = bmax <= 1e-10 ? 1 : 0;
      }
      __syncthreads();                               // (B)
      uniform_pose(s_Tn, Tn);
      const int conv = s_conv;
      // A trial that follows a rejection is almost always rejected too (every refinement ends in five rejections in a row at the noise floor of its chi2: a third
      // of all sweeps), and a rejected trial's Jacobian sums are never used: such a trial gets a chi2-only sweep ("mo_spec"); should it be accepted after all, the
      // full sweep at the same pose follows.  Either way the sums the loop goes on with are those of the full sweep: identical bits.
      const bool lean = spec != 0 && trial >= 1;
      if (lean) mo2_sweep_chi2(Tn, ob, n_ok, s_idx, res, cam, prm.robust_kernel, prm.kernel_param, s_part);
      else mo2_sweep<false>(Tn, ob, n_ok, s_idx, res, cam, prm.robust_kernel, prm.kernel_param, s_part);
      __syncthreads();                               // (A)
      const double new_chi2 = total(27), new_max = total_max(28);
      if (isnan(new_chi2)) { status = 2; stop = true; break; }      // the reference throws here
      rho = chi2 - new_chi2;
      if (rho > 0 && lean) {
        __syncthreads();                             // every lane has read the totals
        mo2_sweep<false>(Tn, ob, n_ok, s_idx, res, cam, prm.robust_kernel, prm.kernel_param, s_part);
        __syncthreads();
      }
      if (rho > 0) {
        if (tid < 12) s_Tc[tid] = s_Tn[tid];         // wave 0 only: it is the one that reads s_Tc (next solve) and rewrites s_Tn (after it)
        chi2 = new_chi2; max_err = new_max;
        load_system();
        stop = conv != 0;                            // |B|_inf <= 1e-10 at the pose the accepted step started from
        const double q = 2 * rho - 1, sc = 1 - q * q * q;
        mu *= fmax(1. / 3., sc);
        nu = 2.; trial = 0;
      } else {
        mu *= nu; nu *= 2.; ++trial;
        if (trial == 5) stop = true;
      }
    } while (!(rho > 0 || stop));
  }
  if (tid < 12) T_io[12 * slot + tid] = s_Tc[tid];
  if (tid == 0) {
    svs_pose_opt_stats st;
    st.initial_chi2 = initial_chi2; st.chi2 = chi2; st.max_err = max_err; st.num_obs = num_obs; st.status = status;
    stats[slot] = st;
  }
  if constexpr (TAIL) {
    __shared__ int s_gcnt[18];
    __shared__ double s_gsum[4];
    __syncthreads();                                 // s_Tc is final
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = s_Tc[i];
    gate_stream(res, Q.pts + slot * Q.pts_b, n, Q.n_new[slot], cam, T, Q.mre, Q.gated + slot * Q.gated_b, Q.ptstats + slot, tid, s_gcnt, s_gsum);
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
      double TQ[16];
      cloud_TQ(T, Q.cams[l], TQ);
      const int cw = Q.cams[l].w / 4, n_px = cw * (Q.cams[l].h / 4);
      const float *disp = Q.disp + slot * Q.disp_b;
      float *cloud = Q.cloud[l] + slot * Q.cloud_b[l];
      for (int i = tid; i < n_px; i += MO2_THREADS) cloud_sample(disp, Q.ds, cw, l, TQ, i, cloud);
    }
  }
}

}  // namespace
// ---- StereoFrontend::processMatchedPoints (stereo_frontend.cpp:834-974), data-parallel part ---------------------------
// One workgroup per stream sweeps the matcher records: reprojection gate at the refined pose, the 2x2 / 3x3 / per-level
// counters (LDS atomics), the pyramid-level positions the host needs for point_tree and the draw lists, and the
// track-length sum.  The host-side remainder (building new_point_list / track_point_list) walks the flags.
namespace {
__global__ __launch_bounds__(256) void gate_matched_kernel(const svs_match_result *__restrict__ res, const svs_candidate_point *__restrict__ pts, int n,
                                                           size_t res_b, size_t pts_b, int n_new, const int32_t *__restrict__ n_new_arr, svs_cam cam,
                                                           const double *__restrict__ Tarr, float mre, svs_gated_point *__restrict__ out, size_t out_b,
                                                           svs_point_stats *__restrict__ stats) {
  __shared__ int s_cnt[18];          // 4 + 9 + 3 + num_track + num_obs
  __shared__ double s_sum[4];
  const int tid = threadIdx.x, slot = blockIdx.x;
  if (n_new_arr) n_new = n_new_arr[slot];
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = Tarr[12 * slot + k];
  gate_stream(res + slot * res_b, pts + slot * pts_b, n, n_new, cam, T, mre, out + slot * out_b, stats + slot, tid, s_cnt, s_sum);
}
}  // namespace

extern "C" int svs_process_matched_points(svs_ctx *ctx, const svs_match_result *d_results, const svs_candidate_point *d_pts, int n,
                                          size_t res_bstride, size_t pts_bstride, int n_new_records, const svs_cam *cam, const double *d_T,
                                          float max_reproj_error, svs_gated_point *d_gated, size_t gated_bstride, svs_point_stats *d_stats,
                                          int batch) {
  SVS_REQUIRE(ctx, ctx && cam && d_T && d_stats && batch >= 1 && n >= 0 && (n == 0 || (d_results && d_pts && d_gated)));
  SVS_DEVICE(ctx);
  hipLaunchKernelGGL(gate_matched_kernel, dim3(batch), dim3(256), 0, ctx->stream, d_results, d_pts, n, res_bstride, pts_bstride, n_new_records,
                     (const int32_t *)nullptr, *cam, d_T, max_reproj_error, d_gated, gated_bstride, d_stats);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
// the same with one record count of the new-feature lists PER STREAM, on the device (frontend.hip: streams carry different candidate lists)
int svs_process_matched_points_dev(svs_ctx *ctx, const svs_match_result *d_results, const svs_candidate_point *d_pts, int n, size_t res_bstride,
                                   size_t pts_bstride, const int32_t *d_n_new_records, const svs_cam *cam, const double *d_T, float max_reproj_error,
                                   svs_gated_point *d_gated, size_t gated_bstride, svs_point_stats *d_stats, int batch) {
  SVS_REQUIRE(ctx, ctx && cam && d_T && d_stats && d_n_new_records && batch >= 1 && n >= 0 && (n == 0 || (d_results && d_pts && d_gated)));
  SVS_DEVICE(ctx);
  hipLaunchKernelGGL(gate_matched_kernel, dim3(batch), dim3(256), 0, ctx->stream, d_results, d_pts, n, res_bstride, pts_bstride, 0, d_n_new_records, *cam, d_T,
                     max_reproj_error, d_gated, gated_bstride, d_stats);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

// calcFastMotionOnly + processMatchedPoints' gate + the three dense clouds of every stream in ONE launch (frontend.hip; see MoTail).  Returns SVS_ERR_UNSUPPORTED
// when the fused refinement kernel does not apply (the caller then takes the separate launches).
int svs_motion_only_gate_cloud(svs_ctx *ctx, const svs_match_result *d_results, int n, size_t res_bstride, const svs_cam *cam, const svs_pose_opt_params *prm,
                               double *d_T_io, svs_pose_opt_stats *d_stats, const svs_mo_tail *tail, int batch) {
  SVS_REQUIRE(ctx, ctx && cam && prm && d_T_io && d_stats && tail && batch >= 1 && n >= 1 && d_results);
  SVS_DEVICE(ctx);
  if ((size_t)n * sizeof(int) > 48 * 1024 || ctx->mo_legacy) return SVS_ERR_UNSUPPORTED;
  for (int l = 0; l < 3; ++l) SVS_REQUIRE(ctx, tail->cams[l].w % 4 == 0 && tail->cams[l].h % 4 == 0 && tail->cloud[l]);
  hipLaunchKernelGGL(motion_only_fused_kernel<true>, dim3(batch), dim3(MO2_THREADS), (size_t)n * sizeof(int), ctx->stream, d_results, n, res_bstride, *cam, *prm, d_T_io,
                     d_stats, *tail, ctx->mo_spec);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

extern "C" int svs_motion_only(svs_ctx *ctx, const svs_match_result *d_results, int n, size_t res_bstride, const svs_cam *cam,
                               const svs_pose_opt_params *prm, double *d_T_io, svs_pose_opt_stats *d_stats, int batch) {
  SVS_REQUIRE(ctx, ctx && cam && prm && d_T_io && d_stats && batch >= 1 && n >= 0 && (n == 0 || d_results));
  SVS_DEVICE(ctx);
  // the record-walking kernel stays for candidate lists whose index list does not fit LDS, and as the A/B partner ("mo_legacy")
  if ((size_t)n * sizeof(int) <= 48 * 1024 && !ctx->mo_legacy)
    hipLaunchKernelGGL(motion_only_fused_kernel<false>, dim3(batch), dim3(MO2_THREADS), (size_t)std::max(n, 1) * sizeof(int), ctx->stream, d_results, n, res_bstride, *cam,
                       *prm, d_T_io, d_stats, MoTail{}, ctx->mo_spec);
  else
    hipLaunchKernelGGL(motion_only_kernel, dim3(batch), dim3(MO_THREADS), 0, ctx->stream, d_results, n, res_bstride, *cam, *prm, d_T_io, d_stats);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
