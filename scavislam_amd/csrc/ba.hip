// ba.hip -- double-window bundle adjustment (Schur-complement LM) for gfx950.
// Replaces SlamGraph::optimize (slam_graph.cpp:312-355): the g2o pipeline BlockSolver_6_3 +
// OptimizationAlgorithmLevenberg + RobustKernelHuber + sparse Cholesky, with the edge
// arithmetic of g2o_types/anchored_points.cpp:148-235 and transformations.h:62-95.
//
// MI355X-first design (f64 throughout; no MFMA: blocks are 3x3 / 6x3 / 6x6):
//  * edges are sorted by (landmark, observer) and packed into chunks of <= 64 whole landmarks'
//    edges; ONE LANE PER EDGE, one wavefront per chunk.  Everything a landmark needs from its
//    edges (H_ll, b_l, the anchor's W and H_pp block) is a SEGMENTED wave64 shuffle reduction
//    over the landmark's lanes -- no LDS, no per-landmark scratch in HBM;
//  * the 3x3 block inverse and all 6x6 Schur outer products W_i D^-1 W_j^T are formed in
//    registers (pairs of a landmark = pairs of lanes of its segment, partner data by shuffle)
//    and go straight into the packed upper-block reduced camera system with hardware f64
//    atomics (global_atomic_add_f64); H_pl is never materialised;
//  * back-substitution re-linearises instead of reloading stored W blocks (64 B/edge read
//    instead of 144+ B/edge) and fuses the trial-state chi2 into the same launch;
//  * landmark shards (multi-GPU) only exchange the packed reduced system (one all-reduce of
//    36*P(P+1)/2 + 12P + 1 doubles) and two scalars per LM trial; the 6P x 6P Cholesky is
//    replicated (deterministic, no broadcast).
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// ---- small f64 helpers ---------------------------------------------------------------------
__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }
// LDS f64 atomic (ds_add_f64): workgroup-scope relaxed add on a __shared__ double
__device__ __forceinline__ void lds_add_f64(double *p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Per-workgroup accumulation window over WIN consecutive poses: the packed upper block triangle
// of the WIN x WIN pose window (plus b_p / b_s rows) lives in LDS; contributions whose poses fall
// outside the window go straight to global atomics.  Landmarks are processed sorted by anchor,
// so a workgroup's landmarks touch a narrow band of poses and nearly everything lands in LDS.
// Trial scalars (d_scal): [2] scale_p [3] fail [4] chi2_cur copy [5..11] solve phase times; the sums every workgroup contributes to --
// chi2 at the trial state and x_l (lambda x_l + b_l) -- are spread over SC_SLOTS words each (a workgroup adds to slot blockIdx % 16:
// hundreds of same-address f64 atomics serialise at the memory side) and folded by their readers
constexpr int SC_SLOTS = 16, SC_CHI = 16, SC_SCL = 32, SC_N = 48;
constexpr int DBG_N = 10;   // phase stamps per wave of the Schur kernel timeline (SVS_BA_DEBUG=2)
constexpr int DBG_X = 8;    // + per-wave structure words: landmarks, same-address multiplicity of the observer / anchor adds, pair rounds
constexpr int DBG_W = DBG_N + DBG_X;
constexpr int WIN = 16;               // widest window of the small pool (two workgroups per CU); the host lays out anchor groups against it
constexpr int WIN_BIG = 22;           // widest window of the big pool (one workgroup per CU)
constexpr int WIN_BLOCKS_MAX = WIN_BIG * (WIN_BIG + 1) / 2;
// LDS stride of one 6x6 window block in doubles: 37 (not 36) spreads the same element of different
// blocks over all banks -- measured 11 vs 32 cycles per ds_add_f64 wave instruction (tools/ubench.hip)
constexpr int WBLK = 37;
// The window is sized PER WORKGROUP: win = the pose span its edges actually touch (pmin .. pmax, at most WIN / WIN_BIG), and the pool
// holds as many COPIES of that window as fit (at most WCOPIES_MAX).  The cost of a ds_add_f64 wave instruction is set by the number of
// lanes that hit the SAME address (~3.6 cycles each: 13 lanes 47 cycles, <= 3 lanes 11 = the floor, tools/ubench.hip) -- lanes of
// different landmarks seen from the same keyframe.  A lane adds into copy (ordinal of its landmark in the wave) % copies, so with c
// copies at most ceil(landmarks per wave / c) lanes collide; a one-anchor workgroup spans ~9 poses = 45 blocks = 6 copies.  The
// flush adds the copies.
constexpr int WCOPIES_MAX = 8;
constexpr int SEG_DPP_MAX = 12;      // longest landmark of a wave for which the segmented sums use DPP shifts (seg_allreduce)
__device__ __host__ constexpr int win_pool_doubles(int nw, int wc) { return wc == 2 ? (163840 - nw * 64 * 18 * 8 - 4096) / 8 : 5226; }      // small pool: one 16-pose window (136 * 37 + 192 doubles, odd stride)
__device__ __forceinline__ int win_blk(int wi, int wj, int win) { return wi * win - wi * (wi - 1) / 2 + (wj - wi); }

// value of the lane below (wave_shr:1 reaches across the 16-lane DPP rows on gfx9), 0 in lane 0 and wherever `keep` is 0: one
// v_and_b32_dpp per word, no LDS-pipeline slot (a __shfl is two ds_bpermute_b32 that queue behind the kernel's LDS atomics)
__device__ __forceinline__ double shr1_keep(double x, unsigned keep) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true) & keep;
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true) & keep;
  return __hiloint2double(hi, lo);
}
// Sum of v over the lanes of each segment (= landmark).  The total is formed in the segment's LAST lane; its first NB values are
// handed to every lane of the segment, the remaining N - NB stay valid in the last lane only (the per-landmark anchor terms, which
// one lane consumes).  Short segments (the common case: a landmark has 3..9 observations) run a sliding sum with DPP shifts: after
// step k a lane holds v[lane] + ... + v[lane - k], cut at its segment head -- maxlen - 1 steps of 3 VALU instructions per value and
// no LDS-pipeline slot.  The recursive-doubling form (long segments) needs fewer steps, but each is two bpermutes per value on the
// LDS pipeline, which the Schur kernel's atomics already saturate (27 values x 4 x 2 = 216 per wave were 5 us of its 40,
// SVS_BA_DEBUG=2 timeline).  The hand-over goes through `slot` (SEG_SLOT doubles of LDS per lane, 16-byte aligned): 9 wide
// writes by the tail lanes + 9 wide reads instead of 54 bpermutes.
// (Measured alternatives at 50 KF / 20k, against 2.7 us per wave for this form: a column sum through LDS -- every lane parks its values,
// one lane per column walks the segment's rows -- 4.3 us, ~40 dependent read groups per lane; ds_add_f64 of the 27 values into the
// segment's first slot -- 4.1 us, the seven waves of a workgroup queue on the LDS pipeline.)
constexpr int SEG_SLOT = 18;
template <int N, int NB>
__device__ __forceinline__ void seg_allreduce(double (&v)[N], int lane, int seg_begin, int seg_end, int maxlen, double *wave_slots) {
  static_assert(NB <= SEG_SLOT && NB <= N, "");
  if (maxlen <= SEG_DPP_MAX) {
    const unsigned keep = lane == seg_begin ? 0u : 0xffffffffu;      // a head lane takes nothing from below
    double s[N];
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = v[i];
    for (int k = 1; k < maxlen; ++k) {
#pragma unroll
      for (int i = 0; i < N; ++i) { s[i] = shr1_keep(s[i], keep); v[i] += s[i]; }
    }
  } else {
    for (int o = 1; o < maxlen; o <<= 1) {
      const bool take = (lane - o) >= seg_begin;
#pragma unroll
      for (int i = 0; i < N; ++i) { double up = __shfl_up(v[i], o, 64); if (take) v[i] += up; }
    }
  }
  double *slot = wave_slots + seg_end * SEG_SLOT;
  if (lane == seg_end) {
#pragma unroll
    for (int i = 0; i < NB; ++i) slot[i] = v[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int i = 0; i < NB; ++i) v[i] = slot[i];
}

__device__ __forceinline__ long blk_index(int i, int j, int P) {   // i <= j, packed upper block row-major
  return (long)i * P - (long)i * (i - 1) / 2 + (j - i);
}

__device__ __forceinline__ void huber(double e2, double delta, int robust, double &rho0, double &rho1) {
  if (!robust) { rho0 = e2; rho1 = 1; return; }
  const double dsqr = delta * delta;
  if (e2 <= dsqr) { rho0 = e2; rho1 = 1.; }
  else { const double s = sqrt(e2); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

// Compact linearisation of one G2oEdgeProjectPSI2UVU (anchored_points.cpp:148-189).  With
//   Jc = d_stereoproj_d_y(y),  D = d_Tinvpsi_d_psi,  Eo = [I | -hat(y)],  Ea = [I | -hat(x_a)]
// the three Jacobians are  J_psi = -Jc D,  J_obs = -Jc Eo,  J_anc = Jc R Ea  (transformations.h:62-95),
// so every block of the normal equations derives from the 3x3 matrix  A = Jc^T (rho1 Lambda) Jc  and
// g = Jc^T (-rho1 Lambda e):
//   H_ll = D^T A D        b_l = -D^T g        W_obs = Eo^T (A D)        W_anc = -Ea^T R^T (A D)
//   M_oo = Eo^T A Eo      M_aa = Ea^T (R^T A R) Ea      M_oa = -Eo^T (A R) Ea
//   b_obs = -Eo^T g       b_anc = Ea^T R^T g
// and E^T X = [X ; v x X(:,c)],  X E = [X | v x X(r,:)]: cross products instead of 3x6 / 6x6 products
// (~350 f64 operations per edge instead of ~1400 for the explicit J^T Omega J forms).
struct EdgeCore {
  double R[9], xa[3], y[3];
  double A[9];             // symmetric 3x3, stored full
  double g[3];
  double D[9];
  double rho0;
};
__device__ __forceinline__ void cross3(const double *v, double x0, double x1, double x2, double &o0, double &o1, double &o2) {
  o0 = v[1] * x2 - v[2] * x1; o1 = v[2] * x0 - v[0] * x2; o2 = v[0] * x1 - v[1] * x0;
}
// M = E^T X E (6x6, full) for symmetric 3x3 X and E = [I | -hat(v)]
__device__ __forceinline__ void sym_block(const double *X, const double *v, double *M) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) M[6 * r + c] = X[3 * r + c];
    cross3(v, X[3 * r], X[3 * r + 1], X[3 * r + 2], M[6 * r + 3], M[6 * r + 4], M[6 * r + 5]);       // top-right rows
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) cross3(v, M[3 + c], M[6 + 3 + c], M[12 + 3 + c], M[18 + 3 + c], M[24 + 3 + c], M[30 + 3 + c]);   // bottom-right cols
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) M[6 * (3 + i) + c] = M[6 * c + 3 + i];
}
// N = Eo^T X Ea (6x6, full) for general 3x3 X, Eo = [I | -hat(y)], Ea = [I | -hat(xa)]
__device__ __forceinline__ void cross_block(const double *X, const double *y, const double *xa, double *N) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) N[6 * r + c] = X[3 * r + c];
    cross3(xa, X[3 * r], X[3 * r + 1], X[3 * r + 2], N[6 * r + 3], N[6 * r + 4], N[6 * r + 5]);
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) cross3(y, N[c], N[6 + c], N[12 + c], N[18 + c], N[24 + c], N[30 + c]);
}

__device__ __forceinline__ void rel_pose(const double *To, const double *Ta, double *R, double *t) {
  // T_ca = T_obs * T_anc^-1
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[3 * i + j] = To[4 * i] * Ta[4 * j] + To[4 * i + 1] * Ta[4 * j + 1] + To[4 * i + 2] * Ta[4 * j + 2];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = To[4 * i + 3] - (R[3 * i] * Ta[3] + R[3 * i + 1] * Ta[7] + R[3 * i + 2] * Ta[11]);
}

__device__ __forceinline__ double edge_chi2(const double *psi, const double *To, const double *Ta, const svs_ba_edge &ed,
                                            const svs_cam &cam, double delta, int robust) {
  double R[9], t[3];
  rel_pose(To, Ta, R, t);
  const double xa[3] = {psi[0] / psi[2], psi[1] / psi[2], 1. / psi[2]};
  double y[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) y[i] = R[3 * i] * xa[0] + R[3 * i + 1] * xa[1] + R[3 * i + 2] * xa[2] + t[i];
  const double e0 = ed.obs[0] - ((y[0] / y[2]) * cam.f + cam.cx);
  const double e1 = ed.obs[1] - ((y[1] / y[2]) * cam.f + cam.cy);
  const double e2_ = ed.obs[2] - (((y[0] - cam.b) / y[2]) * cam.f + cam.cx);
  const double e2 = e0 * e0 * ed.info[0] + e1 * e1 * ed.info[1] + e2_ * e2_ * ed.info[2];
  double r0, r1;
  huber(e2, delta, robust, r0, r1);
  return r0;
}

__device__ __forceinline__ void linearize_edge(const double *psi, const double *To, const double *Ta, const svs_ba_edge &ed,
                                               const svs_cam &cam, double delta, int robust, EdgeCore &o) {
  double t[3];
  rel_pose(To, Ta, o.R, t);
  const double ipz = 1. / psi[2];
  o.xa[0] = psi[0] * ipz; o.xa[1] = psi[1] * ipz; o.xa[2] = ipz;        // invert_depth, maths_utils.h:66-69
  double Rx[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { Rx[i] = o.R[3 * i] * o.xa[0] + o.R[3 * i + 1] * o.xa[1] + o.R[3 * i + 2] * o.xa[2]; o.y[i] = Rx[i] + t[i]; }
  const double iz = 1. / o.y[2], fz = cam.f * iz;
  double err[3];
  err[0] = ed.obs[0] - (o.y[0] * fz + cam.cx);
  err[1] = ed.obs[1] - (o.y[1] * fz + cam.cy);
  err[2] = ed.obs[2] - ((o.y[0] - cam.b) * fz + cam.cx);
  const double e2 = err[0] * err[0] * ed.info[0] + err[1] * err[1] * ed.info[1] + err[2] * err[2] * ed.info[2];
  double rho1;
  huber(e2, delta, robust, o.rho0, rho1);
  const double om0 = rho1 * ed.info[0], om1 = rho1 * ed.info[1], om2 = rho1 * ed.info[2];
  const double w0 = -om0 * err[0], w1 = -om1 * err[1], w2 = -om2 * err[2];
  // Jc = [a 0 c0; 0 a c1; a 0 c2]  (d_stereoproj_d_y, transformations.h:62-71)
  const double a = fz, c0 = -fz * o.y[0] * iz, c1 = -fz * o.y[1] * iz, c2 = -fz * (o.y[0] - cam.b) * iz;
  o.A[0] = a * a * (om0 + om2); o.A[1] = 0.0; o.A[2] = a * (om0 * c0 + om2 * c2);
  o.A[4] = a * a * om1; o.A[5] = a * om1 * c1;
  o.A[8] = om0 * c0 * c0 + om1 * c1 * c1 + om2 * c2 * c2;
  o.A[3] = o.A[1]; o.A[6] = o.A[2]; o.A[7] = o.A[5];
  o.g[0] = a * (w0 + w2); o.g[1] = a * w1; o.g[2] = c0 * w0 + c1 * w1 + c2 * w2;
  // D = [r1 r2 -R x_a] / psi_z  (d_Tinvpsi_d_psi, transformations.h:82-95)
#pragma unroll
  for (int i = 0; i < 3; ++i) { o.D[3 * i] = o.R[3 * i] * ipz; o.D[3 * i + 1] = o.R[3 * i + 1] * ipz; o.D[3 * i + 2] = -Rx[i] * ipz; }
}

struct BaDev {
  int P, L, E, C, n_chunks;
  const double *poses, *psi;            // current state
  double *poses_trial, *psi_trial;      // trial state
  const svs_ba_edge *edges;
  const int *chunk_start, *chunk_len;
  const svs_ba_constraint *cons;
  double *H;                            // packed upper blocks [nblk][36]
  double *bp, *bs;                      // [6P] pure b, Schur correction
  double *chi2_cur;                     // [SC_SLOTS] partial sums of chi2 at the current state (a workgroup adds to slot blockIdx % SC_SLOTS)
  const double *x;                      // [6P] pose solution
  double *scal;                         // [0]=chi2_trial [1]=scale_l [2]=scale_p [3]=fail [4]=chi2_cur copy
  svs_cam cam;
  double delta, lambda;
  int robust, self_mode;
  long long *dbg;                       // SVS_BA_DEBUG=2: per-wave phase stamps (100 MHz ticks), DBG_N per chunk
  double *ctl;                          // device-side LM control (speculative trials): [0] lambda, [1] abort flag, [8 + 8 it ..] trial records
  int fuse_cons;                        // constraints ride in extra workgroups of the landmark kernels
  int n_wide;                           // landmarks with more than 64 observations: one workgroup each (ba_wide_landmark_kernel)
  const int *wide_start, *wide_len;     // their edge ranges (behind the chunked edges)
  int wide_split;                       // workgroups per wide landmark in the Schur pass (they share its pair rounds)
};
__device__ __forceinline__ double chi2_cur_sum(const BaDev &B) { double t = 0; for (int i = 0; i < SC_SLOTS; ++i) t += B.chi2_cur[i]; return t; }

// ---- pose-pose constraints: G2oEdgeSE3 (anchored_points.cpp:207-235) -------------------------
__device__ void d_so3_log(const double *R, double *w, double &theta) {
  double q[4];
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) { double s = sqrt(tr + 1.0) * 2; q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s; }
  else if (R[0] > R[4] && R[0] > R[8]) { double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s; }
  else if (R[4] > R[8]) { double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s; }
  else { double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s; }
  const double n = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), ww = q[0];
  double two_atan;
  if (n < 1e-10) two_atan = 2.0 / ww - 2.0 * (n * n) / (ww * ww * ww);
  else if (fabs(ww) < 1e-10) two_atan = (ww > 0 ? M_PI : -M_PI) / n;
  else two_atan = 2.0 * atan(n / ww) / n;
  w[0] = two_atan * q[1]; w[1] = two_atan * q[2]; w[2] = two_atan * q[3];
  theta = two_atan * n;
}
__device__ void d_pose_mul(const double *A, const double *Bm, double *Cm) {
  double t[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) t[4 * i + j] = A[4 * i] * Bm[j] + A[4 * i + 1] * Bm[4 + j] + A[4 * i + 2] * Bm[8 + j];
    t[4 * i + 3] += A[4 * i + 3];
  }
  for (int i = 0; i < 12; ++i) Cm[i] = t[i];
}
__device__ void d_pose_inv(const double *A, double *Bm) {
  double t[12];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[4 * i + j] = A[4 * j + i];
  for (int i = 0; i < 3; ++i) t[4 * i + 3] = -(t[4 * i] * A[3] + t[4 * i + 1] * A[7] + t[4 * i + 2] * A[11]);
  for (int i = 0; i < 12; ++i) Bm[i] = t[i];
}
__device__ void d_se3_log(const double *T, double *x) {
  double R[9], W[9], W2[9], Vi[9], th;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = T[4 * i + j];
  d_so3_log(R, x + 3, th);
  const double *w = x + 3;
  W[0] = 0; W[1] = -w[2]; W[2] = w[1]; W[3] = w[2]; W[4] = 0; W[5] = -w[0]; W[6] = -w[1]; W[7] = w[0]; W[8] = 0;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
  const double c = fabs(th) < 1e-10 ? 1.0 / 12.0 : (1.0 - th / (2.0 * tan(th / 2.0))) / (th * th);
  for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * W[i] + c * W2[i];
  Vi[0] += 1; Vi[4] += 1; Vi[8] += 1;
  for (int i = 0; i < 3; ++i) x[i] = Vi[3 * i] * T[3] + Vi[3 * i + 1] * T[7] + Vi[3 * i + 2] * T[11];
}
// One wavefront per constraint; lane (i,j) = element of the 6x6 products, operands staged in LDS.
// (A one-thread-per-constraint version spent 60 us in scratch-spilled serial 6x6x6 products.)
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }

// one wave per constraint (lanes 0..63 of the calling workgroup); s_m: 8 x 36 doubles of LDS
// 0 Adj(T21) 1 dl(e) 2 t1 3 J1 4 dl(-e) 5 J2 6 O*J1 7 O*J2
template <int MODE>
__device__ __forceinline__ void ba_constraint_body(const BaDev &B, int c, double (*s_m)[36]) {
  const int lane = threadIdx.x;
  const svs_ba_constraint &cc = B.cons[c];
  const double *poses = MODE == 0 ? B.poses : B.poses_trial;
  double T2i[12], t[12], err[6];
  d_pose_inv(poses + 12 * cc.pose2, T2i);
  d_pose_mul(cc.T_21, poses + 12 * cc.pose1, t);
  d_pose_mul(t, T2i, t);
  d_se3_log(t, err);                                    // redundant per lane: ~300 flop
  double oe[6], e2 = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) { oe[i] = 0; for (int j = 0; j < 6; ++j) oe[i] += cc.info[6 * i + j] * err[j]; e2 += err[i] * oe[i]; }
  if (MODE == 1) { if (lane == 0) atomic_add_f64(&B.scal[SC_CHI + (c & (SC_SLOTS - 1))], e2); return; }
  if (lane == 0) atomic_add_f64(B.chi2_cur + (c & (SC_SLOTS - 1)), e2);
  const int i = lane / 6, j = lane - 6 * i;             // lanes 0..35 own element (i,j)
  const bool on = lane < 36;
  // Adj(T21) = [[R, t^R],[0, R]];  d_lieBracketab_by_d_a(d) = -ad_d   
  if (on) {
    const double *A = cc.T_21;
    const double tt[3] = {A[3], A[7], A[11]};
    const int bi = i % 3, bj = j % 3;
    const double Rij = A[4 * bi + bj];
    // (t^ R)[bi][bj]
    const double th[9] = {0, -tt[2], tt[1], tt[2], 0, -tt[0], -tt[1], tt[0], 0};
    const double tR = th[3 * bi] * A[bj] + th[3 * bi + 1] * A[4 + bj] + th[3 * bi + 2] * A[8 + bj];
    double adj = 0, dlp = 0, dlm = 0;
    const double hu[9] = {0, -err[2], err[1], err[2], 0, -err[0], -err[1], err[0], 0};
    const double hw[9] = {0, -err[5], err[4], err[5], 0, -err[3], -err[4], err[3], 0};
    if (i < 3 && j < 3) { adj = Rij; dlp = -hw[3 * bi + bj]; }
    else if (i < 3 && j >= 3) { adj = tR; dlp = -hu[3 * bi + bj]; }
    else if (i >= 3 && j >= 3) { adj = Rij; dlp = -hw[3 * bi + bj]; }
    dlm = -dlp;                                         // dl(-e)
    s_m[0][lane] = adj; s_m[1][lane] = dlp; s_m[4][lane] = dlm;
  }
  wave_lds_sync();
  auto mm = [&](int a, int b) { double s = 0; for (int k = 0; k < 6; ++k) s += s_m[a][6 * i + k] * s_m[b][6 * k + j]; return s; };
  // J1 = third(T21, e) = Adj + 1/2 dl Adj + 1/12 dl dl Adj
  double t1 = on ? mm(1, 0) : 0.0;
  if (on) s_m[2][lane] = t1;
  wave_lds_sync();
  double t2 = on ? mm(1, 2) : 0.0;
  const double J1 = on ? s_m[0][lane] + 0.5 * t1 + (1. / 12.) * t2 : 0.0;
  // J2 = -third(I, -e) = -(I + 1/2 dl(-e) + 1/12 dl(-e)^2)
  double u2 = on ? mm(4, 4) : 0.0;
  const double J2 = on ? -((i == j ? 1.0 : 0.0) + 0.5 * s_m[4][lane] + (1. / 12.) * u2) : 0.0;
  wave_lds_sync();
  if (on) { s_m[3][lane] = J1; s_m[5][lane] = J2; }
  wave_lds_sync();
  // O*J1, O*J2
  if (on) {
    double a1 = 0, a2 = 0;
    for (int k = 0; k < 6; ++k) { a1 += cc.info[6 * i + k] * s_m[3][6 * k + j]; a2 += cc.info[6 * i + k] * s_m[5][6 * k + j]; }
    s_m[6][lane] = a1; s_m[7][lane] = a2;
  }
  wave_lds_sync();
  const int p1 = cc.pose1, p2 = cc.pose2, P = B.P;
  if (on) {
    double s11 = 0, s22 = 0, s12 = 0;
    for (int k = 0; k < 6; ++k) { s11 += s_m[3][6 * k + i] * s_m[6][6 * k + j]; s22 += s_m[5][6 * k + i] * s_m[7][6 * k + j]; s12 += s_m[3][6 * k + i] * s_m[7][6 * k + j]; }
    if (j >= i) { atomic_add_f64(&B.H[blk_index(p1, p1, P) * 36 + lane], s11); atomic_add_f64(&B.H[blk_index(p2, p2, P) * 36 + lane], s22); }
    if (p1 != p2) {
      const bool up = p1 < p2;
      atomic_add_f64(&B.H[(up ? blk_index(p1, p2, P) : blk_index(p2, p1, P)) * 36 + (up ? 6 * i + j : 6 * j + i)], s12);
    }
  }
  if (lane < 6) {
    double s1 = 0, s2 = 0;
    for (int k = 0; k < 6; ++k) { s1 += s_m[3][6 * k + lane] * oe[k]; s2 += s_m[5][6 * k + lane] * oe[k]; }
    atomic_add_f64(&B.bp[6 * p1 + lane], -s1);
    atomic_add_f64(&B.bp[6 * p2 + lane], -s2);
  }
}
// stand-alone launch (windows without landmarks on this rank; otherwise the constraints ride in extra workgroups of
// the landmark kernels)
template <int MODE>
__global__ __launch_bounds__(64) void ba_constraint_kernel(BaDev B) {
  if (B.ctl) { if (B.ctl[1] != 0.0) return; B.lambda = B.ctl[0]; }      // speculative trial: lambda decided on the device, skipped after a rejection
  __shared__ double s_m[8][36];
  ba_constraint_body<MODE>(B, blockIdx.x, s_m);
}

// Levenberg accept / reject of one speculative trial (OptimizationAlgorithmLevenberg::solve), on one lane
__device__ __forceinline__ void ba_lm_decide(const BaDev &B, int it) {
  double *rec = B.ctl + 8 + 8 * it;
  const double chi_cur = B.scal[4], fail = B.scal[3];
  double chi_t = 0, scale_l = 0;
  for (int i = 0; i < SC_SLOTS; ++i) { chi_t += B.scal[SC_CHI + i]; scale_l += B.scal[SC_SCL + i]; }
  const double tempChi = fail != 0.0 ? 1.7976931348623157e308 : chi_t;
  const double scale = scale_l + B.scal[2] + 1e-3;
  const double rho = (chi_cur - tempChi) / scale;
  double lambda = B.ctl[0];
  const bool accept = rho > 0 && isfinite(tempChi);
  if (accept) {
    const double q = 2 * rho - 1;
    double alpha = 1. - q * q * q;
    alpha = fmin(alpha, 2. / 3.);
    lambda *= fmax(1. / 3., alpha);
    B.ctl[0] = lambda;
  } else {
    B.ctl[1] = 1.0;
  }
  rec[0] = chi_cur; rec[1] = tempChi; rec[2] = scale; rec[3] = rho; rec[4] = accept ? 1.0 : 0.0; rec[5] = lambda; rec[6] = fail; rec[7] = 1.0;
}

// MODE 0: accumulate reduced system + chi2 at the current state.
// MODE 1: back-substitute landmarks (psi_trial = psi + x_l), scale_l, chi2 at the trial state.
// NW waves (chunks) per workgroup: 4 by default; the host picks more when that brings the number of workgroups down to
// one per CU (at 50 KF / 20k: 1 612 chunks -> 231 workgroups of 7 waves instead of 403 of 4 that load the CUs unevenly).
template <int MODE, int NW, int WC = 1>
__global__ __launch_bounds__(NW * 64, (NW <= 4 && WC == 1) ? 2 : 1) void ba_landmark_kernel(BaDev B) {
  constexpr int NT = NW * 64;
  constexpr int POOL = win_pool_doubles(NW, WC), WMAX = WC == 2 ? WIN_BIG : WIN;
  if (B.ctl) { if (B.ctl[1] != 0.0) return; B.lambda = B.ctl[0]; }      // speculative trial: lambda decided on the device, skipped after a rejection
  const int lane = threadIdx.x & 63;
  __shared__ double s_cons[8][36];
  const int n_lm_blocks = (B.n_chunks + NW - 1) / NW;
  if (B.fuse_cons && (int)blockIdx.x >= n_lm_blocks) {                      // pose-pose constraints: one wave each
    if (threadIdx.x < 64) ba_constraint_body<MODE>(B, blockIdx.x - n_lm_blocks, s_cons);
    return;
  }
  const int chunk = blockIdx.x * NW + (threadIdx.x >> 6);
  const bool wave_valid = chunk < B.n_chunks;                        // wave-uniform
  const int e0 = wave_valid ? B.chunk_start[chunk] : 0, len = wave_valid ? B.chunk_len[chunk] : 0;
  const bool active = lane < len;
#define SVS_STAMP(k) do { if (B.dbg && lane == 0 && wave_valid) B.dbg[DBG_W * (size_t)chunk + (k)] = (long long)wall_clock64(); } while (0)
  SVS_STAMP(0);
  svs_ba_edge ed;
  if (active) ed = B.edges[e0 + lane];
  else { ed.point = -1 - lane; ed.pose = 0; ed.anchor = 0; }
  __shared__ __attribute__((aligned(16))) double s_win[MODE == 0 ? POOL : 1];      // `copies` x [window blocks (stride WBLK) | b_p rows | b_s rows]
  __shared__ __attribute__((aligned(16))) double s_wo[NW * 64 * 18];   // W_obs of every edge lane (MODE 0); before that, hand-over slots of the segmented sums
  __shared__ int s_wmin[NW], s_wmax[NW];
  __shared__ unsigned char s_wrow[MODE == 0 ? WIN_BLOCKS_MAX : 1];
  __shared__ double s_scal[2];      // per-workgroup chi2 (MODE 0) / trial chi2 and scale (MODE 1)
  if (threadIdx.x < 2) s_scal[threadIdx.x] = 0.0;
  if (MODE == 0 && blockIdx.x == 0 && threadIdx.x < SC_N) B.scal[threadIdx.x] = 0.0;      // trial scalars: zeroed here instead of by a memset launch
  if (MODE == 0) {      // the whole pool is zeroed while the edge records are on their way (how much of it the window uses is known after them)
    double2 *w2 = reinterpret_cast<double2 *>(s_win);
    const double2 z2 = {0.0, 0.0};
    for (int i = threadIdx.x; i < POOL / 2; i += NT) w2[i] = z2;
  }
#define SVS_SUBSTAMP(k) do { if (B.dbg) { __builtin_amdgcn_s_waitcnt(0x0F70); if (lane == 0 && wave_valid) B.dbg[DBG_W * (size_t)chunk + DBG_N + (k)] = (long long)wall_clock64(); } } while (0)
  SVS_SUBSTAMP(4);
  // the state gathers depend on the record alone: in flight while the window is set up (three barriers further down)
  double psi[3] = {1, 1, 1}, To[12], Ta[12];
  if (active) {
#pragma unroll
    for (int i = 0; i < 3; ++i) psi[i] = B.psi[3 * (size_t)ed.point + i];
#pragma unroll
    for (int i = 0; i < 12; ++i) { To[i] = B.poses[12 * (size_t)ed.pose + i]; Ta[i] = B.poses[12 * (size_t)ed.anchor + i]; }
  }
  SVS_SUBSTAMP(5);
  int pmin = 0, win = 1, nblk = 1, cstride = 1, copies = 1;
  if (MODE == 0) {
    int mn = active ? min(ed.pose, ed.anchor) : 0x7fffffff, mx = active ? max(ed.pose, ed.anchor) : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, 64)); mx = max(mx, __shfl_xor(mx, o, 64)); }
    if (lane == 0) { s_wmin[threadIdx.x >> 6] = mn; s_wmax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    SVS_SUBSTAMP(6);
    pmin = s_wmin[0];
    int pmax = s_wmax[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) { pmin = min(pmin, s_wmin[k]); pmax = max(pmax, s_wmax[k]); }
    if (pmin != 0x7fffffff) {
      win = min(pmax - pmin + 1, WMAX) + 1;
      do {                                                  // (WMAX fits the pool once; the loop is the guarantee, not the rule)
        --win;
        nblk = win * (win + 1) / 2;
        cstride = (nblk * WBLK + 12 * win) | 1;             // odd: the copies start in different banks
      } while (cstride > POOL && win > 1);
      copies = max(1, min(WCOPIES_MAX, POOL / cstride));
    }
    if (threadIdx.x < nblk) {                               // block row of every packed window block (for the flush, behind the next barrier)
      int wi = 0, rem = threadIdx.x;
      while (rem >= win - wi) { rem -= win - wi; ++wi; }
      s_wrow[threadIdx.x] = (unsigned char)wi;
    }
  }
  SVS_SUBSTAMP(7);
  // segment (= landmark) bounds inside the wave
  const int prev_point = __shfl_up(ed.point, 1, 64);
  const bool head = lane == 0 || prev_point != ed.point;
  const unsigned long long heads = __ballot(head);
  const unsigned long long le_mask = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
  const int seg_begin = 63 - __clzll((long long)(heads & le_mask));
  const unsigned long long gt = (lane == 63) ? 0ull : (heads & ~le_mask);
  const int seg_end = gt ? (__ffsll((long long)gt) - 2) : 63;
  int maxlen = seg_end - seg_begin + 1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
  // The waves of a workgroup share the CU's LDS pipeline and meet at the flush barrier: the ones holding the largest landmarks (most
  // pair rounds) are the critical path, so they get the higher issue priority
  if (MODE == 0) {
    const int ml = __builtin_amdgcn_readfirstlane(maxlen);
    if (ml >= 7) __builtin_amdgcn_s_setprio(3);
    else if (ml >= 5) __builtin_amdgcn_s_setprio(2);
    else if (ml >= 4) __builtin_amdgcn_s_setprio(1);
  }

  const int P = B.P;
  EdgeCore lin;
  const bool self = active && ed.pose == ed.anchor;
  if (active) {
    linearize_edge(psi, To, Ta, ed, B.cam, B.delta, B.robust, lin);
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) { lin.R[i] = 0; lin.A[i] = 0; lin.D[i] = 0; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { lin.xa[i] = 0; lin.y[i] = 0; lin.g[i] = 0; }
    lin.rho0 = 0;
  }
  SVS_STAMP(1);
  // one global atomic per WORKGROUP for the scalar sums: same-address f64 atomics serialise in L2 (~20 ns each), and one
  // per wave cost more than the rest of the back-substitution kernel
  if (MODE == 0) {
    const double c = wave_sum_f64(lin.rho0);
    if (lane == 0 && c != 0.0) lds_add_f64(&s_scal[0], c);      // window init barrier above ordered the zeroing
  }
  // observer role: this lane owns pose ed.pose of its landmark.  A self edge (observer == anchor: R = I, y = x_a up to
  // rounding) has J_obs + J_anc = 0: it contributes to H_ll / b_l only, plus -- in G2O_LITERAL mode, SURVEY.md B-7 --
  // the slot terms M_oo + M_aa + sym(M_oa) on the anchor's diagonal block, which collapse to M_aa = Ea^T A Ea
  // (differences are O(eps |M|)); its b and W slot terms cancel.
  const bool obs_role = active && !self;
  const bool self_lit = self && B.self_mode == 0;

  // ---- per landmark sums, one segmented reduction: H_ll = sum D^T A D, b_l = -sum D^T g, and -- because x_a is common
  //      to all edges of a landmark -- W_anc = -Ea^T (sum R^T A D), M_aa = Ea^T (sum R^T A R) Ea, b_anc = Ea^T (sum R^T g)
  double Bm[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Bm[3 * i + j] = lin.A[3 * i] * lin.D[j] + lin.A[3 * i + 1] * lin.D[3 + j] + lin.A[3 * i + 2] * lin.D[6 + j];
  constexpr int NRED = MODE == 0 ? 27 : 18;
  double red[NRED];      // [0..5] H_ll upper, [6..8] b_l, [9..17] S_RB = sum R^T Bm, (MODE 0:) [18..23] S_RAR upper, [24..26] S_Rg
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) red[k++] = lin.D[i] * Bm[j] + lin.D[3 + i] * Bm[3 + j] + lin.D[6 + i] * Bm[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) red[6 + i] = -(lin.D[i] * lin.g[0] + lin.D[3 + i] * lin.g[1] + lin.D[6 + i] * lin.g[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) red[9 + 3 * i + j] = obs_role ? lin.R[i] * Bm[j] + lin.R[3 + i] * Bm[3 + j] + lin.R[6 + i] * Bm[6 + j] : 0.0;
    if (MODE == 0) {
      double AR[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) AR[3 * i + j] = lin.A[3 * i] * lin.R[j] + lin.A[3 * i + 1] * lin.R[3 + j] + lin.A[3 * i + 2] * lin.R[6 + j];
      const bool in_maa = obs_role || self_lit;
      k = 18;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) red[k++] = in_maa ? lin.R[i] * AR[j] + lin.R[3 + i] * AR[3 + j] + lin.R[6 + i] * AR[6 + j] : 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) red[24 + i] = obs_role ? lin.R[i] * lin.g[0] + lin.R[3 + i] * lin.g[1] + lin.R[6 + i] * lin.g[2] : 0.0;
    }
  }
  SVS_STAMP(2);
  seg_allreduce<NRED, 18>(red, lane, seg_begin, seg_end, maxlen, s_wo + (threadIdx.x >> 6) * 64 * 18);
  SVS_STAMP(3);
  double Di[9], bl[3] = {red[6], red[7], red[8]};
  {
    const double a00 = red[0] + B.lambda, a01 = red[1], a02 = red[2], a11 = red[3] + B.lambda, a12 = red[4], a22 = red[5] + B.lambda;
    // closed-form inverse of the symmetric 3x3 (Eigen fixed-size inverse = cofactors / det)
    const double c00 = a11 * a22 - a12 * a12, c01 = a12 * a02 - a01 * a22, c02 = a01 * a12 - a11 * a02;
    const double id = 1.0 / (a00 * c00 + a01 * c01 + a02 * c02);
    Di[0] = c00 * id; Di[1] = c01 * id; Di[2] = c02 * id;
    Di[3] = Di[1]; Di[4] = (a00 * a22 - a02 * a02) * id; Di[5] = (a02 * a01 - a00 * a12) * id;
    Di[6] = Di[2]; Di[7] = Di[5]; Di[8] = (a00 * a11 - a01 * a01) * id;
  }
  double Db[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) Db[i] = Di[3 * i] * bl[0] + Di[3 * i + 1] * bl[1] + Di[3 * i + 2] * bl[2];
  // W_obs = Eo^T Bm = [Bm ; y x Bm(:,c)]  (0 for a self edge);  W_anc = -Ea^T S_RB = -[S_RB ; x_a x S_RB(:,c)], rebuilt where needed
  double Wo[18];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double t0, t1, t2;
    cross3(lin.y, Bm[c], Bm[3 + c], Bm[6 + c], t0, t1, t2);
    Wo[c] = obs_role ? Bm[c] : 0.0; Wo[3 + c] = obs_role ? Bm[3 + c] : 0.0; Wo[6 + c] = obs_role ? Bm[6 + c] : 0.0;
    Wo[9 + c] = obs_role ? t0 : 0.0; Wo[12 + c] = obs_role ? t1 : 0.0; Wo[15 + c] = obs_role ? t2 : 0.0;
  }
  auto make_WA = [&](double (&WA)[18]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double t0, t1, t2;
      cross3(lin.xa, red[9 + c], red[12 + c], red[15 + c], t0, t1, t2);
      WA[c] = -red[9 + c]; WA[3 + c] = -red[12 + c]; WA[6 + c] = -red[15 + c];
      WA[9 + c] = -t0; WA[12 + c] = -t1; WA[15 + c] = -t2;
    }
  };
  SVS_STAMP(4);
  const int anchor = ed.anchor;

  if (MODE == 1) {
    // x_l = D^-1 (b_l - sum_i W_i^T x_i - W_A^T x_A)
    double c[3] = {0, 0, 0};
    if (obs_role) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) c[j] += Wo[3 * i + j] * B.x[6 * ed.pose + i];
    }
    __builtin_amdgcn_wave_barrier();      // every lane has read its slot of the first hand-over
    seg_allreduce<3, 3>(c, lane, seg_begin, seg_end, maxlen, s_wo + (threadIdx.x >> 6) * 64 * 18);
    double xl[3] = {0, 0, 0};
    if (active) {
      double WA[18];
      make_WA(WA);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double s = bl[j] - c[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) s -= WA[3 * i + j] * B.x[6 * anchor + i];
        c[j] = s;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) xl[i] = Di[3 * i] * c[0] + Di[3 * i + 1] * c[1] + Di[3 * i + 2] * c[2];
    }
    double npsi[3] = {psi[0] + xl[0], psi[1] + xl[1], psi[2] + xl[2]};   // G2oVertexPointXYZ::oplusImpl
    double sc = 0, chi = 0;
    if (active) {
      if (head) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { B.psi_trial[3 * (size_t)ed.point + i] = npsi[i]; sc += xl[i] * (B.lambda * xl[i] + bl[i]); }
      }
      double Tno[12], Tna[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) { Tno[i] = B.poses_trial[12 * (size_t)ed.pose + i]; Tna[i] = B.poses_trial[12 * (size_t)anchor + i]; }
      chi = edge_chi2(npsi, Tno, Tna, ed, B.cam, B.delta, B.robust);
    }
    sc = wave_sum_f64(sc);
    chi = wave_sum_f64(chi);
    __syncthreads();                                   // s_scal zeroed
    if (lane == 0) { lds_add_f64(&s_scal[0], chi); lds_add_f64(&s_scal[1], sc); }
    __syncthreads();
    if (threadIdx.x == 0) { atomic_add_f64(&B.scal[SC_CHI + (blockIdx.x & (SC_SLOTS - 1))], s_scal[0]); atomic_add_f64(&B.scal[SC_SCL + (blockIdx.x & (SC_SLOTS - 1))], s_scal[1]); }
    return;
  }

  // ---- MODE 0: reduced camera system -------------------------------------------------------
  // A block (pi <= pj) of the reduced system lives either in the workgroup's LDS window or (out of window) in global memory.  The
  // test is made ONCE per block and the six values of a block row are added under one branch: the per-element form of this
  // (a divergent `if` around every one of the ~270 adds of a lane) cost as many scalar instructions as the kernel has vector ones
  // (PMC: 2440 SALU vs 2490 VALU per wave, profiles/r2_notes.md).
  // (destinations are kept as INDICES into s_win / B.H, not as pointers: a pointer that may point to either would be a flat pointer and
  //  every add a flat atomic)
  const int lm_ord = __popcll(heads & le_mask) - 1;                            // ordinal of this lane's landmark in the wave
  const int my_copy = lm_ord % copies, my_base = my_copy * cstride, vec_base = my_base + nblk * WBLK;
  struct Dst { bool in_lds; int lds; long glb; };
  auto blk_dst = [&](int pi, int pj) __attribute__((always_inline)) {
    const int wi = pi - pmin, wj = pj - pmin;
    Dst d;
    d.in_lds = wj < win;
    d.lds = d.in_lds ? my_base + win_blk(wi, wj, win) * WBLK : 0;
    d.glb = d.in_lds ? 0 : blk_index(pi, pj, B.P) * 36;
    return d;
  };
  // v[c] -> element (r, c) of the block (stride 1), or -- transposed store -- element (c, r) (stride 6)
  auto add_row6 = [&](const Dst &d, int first, int stride, const double (&v)[6], int c0) __attribute__((always_inline)) {
    if (d.in_lds) {
#pragma unroll
      for (int c = 0; c < 6; ++c) if (c >= c0) lds_add_f64(&s_win[d.lds + first + stride * c], v[c]);
    } else {
#pragma unroll
      for (int c = 0; c < 6; ++c) if (c >= c0) atomic_add_f64(&B.H[d.glb + first + stride * c], v[c]);
    }
  };
  auto add_vec6 = [&](int which, int p, const double (&v)[6]) __attribute__((always_inline)) {      // which: 0 = b_p, 1 = b_s
    const int wp = p - pmin;
    if (wp < win) {
#pragma unroll
      for (int c = 0; c < 6; ++c) lds_add_f64(&s_win[vec_base + (which * win + wp) * 6 + c], v[c]);
    } else {
      double *g = (which ? B.bs : B.bp) + 6 * p;
#pragma unroll
      for (int c = 0; c < 6; ++c) atomic_add_f64(g + c, v[c]);
    }
  };
  // (1) anchor block, once per landmark: Ea^T S_RAR Ea - (W_A D^-1) W_A^T,  b_anc = Ea^T S_Rg,  Schur rhs W_A D^-1 b_l
  if (active && lane == seg_end) {      // the lane that holds the landmark's complete sums
    double WA[18], WAD[18], SR[9], Maa[36];
    make_WA(WA);
    SR[0] = red[18]; SR[1] = red[19]; SR[2] = red[20]; SR[3] = red[19]; SR[4] = red[21]; SR[5] = red[22]; SR[6] = red[20]; SR[7] = red[22]; SR[8] = red[23];
    sym_block(SR, lin.xa, Maa);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) WAD[3 * i + j] = WA[3 * i] * Di[j] + WA[3 * i + 1] * Di[3 + j] + WA[3 * i + 2] * Di[6 + j];
    const Dst dst = blk_dst(anchor, anchor);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double v[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) v[c] = c >= r ? Maa[6 * r + c] - (WAD[3 * r] * WA[3 * c] + WAD[3 * r + 1] * WA[3 * c + 1] + WAD[3 * r + 2] * WA[3 * c + 2]) : 0.0;
      add_row6(dst, 6 * r, 1, v, r);
    }
    double t0, t1, t2;
    cross3(lin.xa, red[24], red[25], red[26], t0, t1, t2);
    const double ba[6] = {red[24], red[25], red[26], t0, t1, t2};
    double bsv[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) bsv[r] = WA[3 * r] * Db[0] + WA[3 * r + 1] * Db[1] + WA[3 * r + 2] * Db[2];
    add_vec6(0, anchor, ba);
    add_vec6(1, anchor, bsv);
  }
  SVS_STAMP(5);
  // (2) observer part: blocks (i,i), (i,A), b_i;  W_obs is parked in LDS for the pair phase
  double *my_wo = s_wo + ((threadIdx.x >> 6) * 64 + lane) * 18;
#pragma unroll
  for (int i = 0; i < 18; ++i) my_wo[i] = Wo[i];
  double WoD[18];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) WoD[3 * i + j] = Wo[3 * i] * Di[j] + Wo[3 * i + 1] * Di[3 + j] + Wo[3 * i + 2] * Di[6 + j];
  if (obs_role) {
    const int pi = ed.pose;
    {
      double Moo[36];
      sym_block(lin.A, lin.y, Moo);                           // M_oo = Eo^T A Eo
      const Dst dst = blk_dst(pi, pi);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double v[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) v[c] = c >= r ? Moo[6 * r + c] - (WoD[3 * r] * Wo[3 * c] + WoD[3 * r + 1] * Wo[3 * c + 1] + WoD[3 * r + 2] * Wo[3 * c + 2]) : 0.0;
        add_row6(dst, 6 * r, 1, v, r);
      }
    }
    {
      double AR[9], Noa[36], WA[18];
      make_WA(WA);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) AR[3 * i + j] = lin.A[3 * i] * lin.R[j] + lin.A[3 * i + 1] * lin.R[3 + j] + lin.A[3 * i + 2] * lin.R[6 + j];
      cross_block(AR, lin.y, lin.xa, Noa);                    // M_oa = -Eo^T (A R) Ea
      const bool up = pi < anchor;
      const Dst dst = blk_dst(up ? pi : anchor, up ? anchor : pi);
      const int rs = up ? 6 : 1, cstr = up ? 1 : 6;           // element (r, c) of M goes to (r, c) of block (pi, anchor) or to (c, r) of block (anchor, pi)
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double v[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) v[c] = -Noa[6 * r + c] - (WoD[3 * r] * WA[3 * c] + WoD[3 * r + 1] * WA[3 * c + 1] + WoD[3 * r + 2] * WA[3 * c + 2]);
        add_row6(dst, rs * r, cstr, v, 0);
      }
    }
    double u0, u1, u2;
    cross3(lin.y, lin.g[0], lin.g[1], lin.g[2], u0, u1, u2);
    const double bo[6] = {-lin.g[0], -lin.g[1], -lin.g[2], -u0, -u1, -u2};    // b_obs = -Eo^T g
    double bsv[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) bsv[r] = Wo[3 * r] * Db[0] + Wo[3 * r + 1] * Db[1] + Wo[3 * r + 2] * Db[2];
    add_vec6(0, pi, bo);
    add_vec6(1, pi, bsv);
  }
  SVS_STAMP(6);
  // (3) observer-observer pairs of a landmark, circulant schedule: in round r the edge with local index a
  //     pairs with (a + r) mod m, so all m lanes of a landmark work for floor(m/2) rounds (instead of one
  //     lane-partner distance per round over m-1 rounds).  -(W_a D^-1) W_b^T goes to block (min, max).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  {
    // The circulant runs over the landmark's OBSERVERS only: its self edge (observer == anchor, at most one per landmark: one edge
    // per (point, keyframe)) has no W_obs and would only lengthen the schedule -- a landmark seen from its anchor and four more
    // keyframes takes 2 rounds instead of 2.5 -> 3, and the wave runs as many rounds as its largest landmark needs.
    const unsigned long long seg_mask = (seg_end == 63 ? ~0ull : ((1ull << (seg_end + 1)) - 1ull)) & ~((1ull << seg_begin) - 1ull);
    const unsigned long long self_in_seg = __ballot(active && !obs_role) & seg_mask;      // (a non-literal self edge included: it pairs with nobody)
    const int self_pos = self_in_seg ? (__ffsll((long long)self_in_seg) - 1) - seg_begin : 64;
    const int m = (seg_end - seg_begin + 1) - (self_in_seg ? 1 : 0);
    const int a_loc = (lane - seg_begin) - ((lane - seg_begin) > self_pos ? 1 : 0);
    int rounds = active ? (m >> 1) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rounds = max(rounds, __shfl_xor(rounds, o, 64));
    const double *wave_wo = s_wo + (threadIdx.x >> 6) * 64 * 18;
    if (B.dbg && wave_valid && lane == 0) B.dbg[DBG_W * (size_t)chunk + DBG_N + 3] = rounds;
    // LDS operations of a wave complete in order: a partner fetch issued right behind the 36 adds of the previous round would wait for
    // all of them.  So the partner of round r+1 (its pose, role and W_obs) is fetched BEFORE the adds of round r are issued, and the
    // products of round r are formed while that fetch (and the adds of round r-1 in front of it) drain.
    auto partner = [&](int r, int &pj, bool &on, double (&Wj)[18]) __attribute__((always_inline)) {
      int b_loc = a_loc + r;
      if (b_loc >= m) b_loc -= m;
      const int lane_b = min(seg_begin + b_loc + (b_loc >= self_pos ? 1 : 0), 63);
      pj = __shfl(ed.pose, lane_b, 64);
      on = obs_role && 2 * r <= m && !(2 * r == m && a_loc >= r);
      const double *wj = wave_wo + lane_b * 18;
#pragma unroll
      for (int i = 0; i < 18; ++i) Wj[i] = wj[i];
    };
    int pj_n = 0; bool on_n = false; double W_n[18];
    if (rounds >= 1) partner(1, pj_n, on_n, W_n);
    for (int r = 1; r <= rounds; ++r) {
      const int pj = pj_n; const bool on = on_n;
      double Wj[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) Wj[i] = W_n[i];
      if (r < rounds) partner(r + 1, pj_n, on_n, W_n);
      if (on) {
        const bool up = ed.pose < pj;
        const Dst dst = blk_dst(up ? ed.pose : pj, up ? pj : ed.pose);
        const int rs = up ? 6 : 1, cstr = up ? 1 : 6;
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {
          double v[6];
#pragma unroll
          for (int c = 0; c < 6; ++c) v[c] = -(WoD[3 * rr] * Wj[3 * c] + WoD[3 * rr + 1] * Wj[3 * c + 1] + WoD[3 * rr + 2] * Wj[3 * c + 2]);
          add_row6(dst, rs * rr, cstr, v, 0);
        }
      }
    }
  }
  SVS_STAMP(7);
  // flush the LDS window: one global atomic per touched element per workgroup
  __syncthreads();
  SVS_STAMP(8);
  if (threadIdx.x == 0 && s_scal[0] != 0.0) atomic_add_f64(B.chi2_cur + (blockIdx.x & (SC_SLOTS - 1)), s_scal[0]);
  if (pmin != 0x7fffffff) {
    for (int i = threadIdx.x; i < nblk * 36; i += NT) {
      const int wb = i / 36, rc = i - wb * 36;
      double v = s_win[wb * WBLK + rc];
      for (int k = 1; k < copies; ++k) v += s_win[k * cstride + wb * WBLK + rc];
      if (v != 0.0) {
        const int wi = s_wrow[wb], pi = pmin + wi, pj = pmin + wi + (wb - win_blk(wi, wi, win));
        if (pj < P) atomic_add_f64(&B.H[blk_index(pi, pj, P) * 36 + rc], v);
      }
    }
    for (int i = threadIdx.x; i < 2 * win * 6; i += NT) {
      double v = s_win[nblk * WBLK + i];
      for (int k = 1; k < copies; ++k) v += s_win[k * cstride + nblk * WBLK + i];
      if (v != 0.0) {
        const int which = i / (win * 6), rest = i - which * win * 6, wp = rest / 6, r = rest - wp * 6;
        if (pmin + wp < P) atomic_add_f64((which ? B.bs : B.bp) + 6 * (pmin + wp) + r, v);
      }
    }
  }
  SVS_STAMP(DBG_N - 1);
  if (B.dbg && wave_valid) {      // structure words of the timeline (debug only)
    const bool tail = active && lane == seg_end;
    int m_obs = 0, m_anc = 0;
    for (int j = 0; j < 64; ++j) {
      const int pj = __shfl(ed.pose, j, 64), aj = __shfl(ed.anchor, j, 64), cj = __shfl(my_copy, j, 64);
      const int oj = __shfl((int)obs_role, j, 64), tj = __shfl((int)tail, j, 64);
      if (obs_role && oj && pj == ed.pose && cj == my_copy) ++m_obs;
      if (tail && tj && aj == ed.anchor && cj == my_copy) ++m_anc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m_obs = max(m_obs, __shfl_xor(m_obs, o, 64)); m_anc = max(m_anc, __shfl_xor(m_anc, o, 64)); }
    const unsigned long long act_mask = __ballot(active);
    if (lane == 0) {
      B.dbg[DBG_W * (size_t)chunk + DBG_N + 0] = __popcll(heads & act_mask) * 100 + copies;      // landmarks * 100 + window copies
      B.dbg[DBG_W * (size_t)chunk + DBG_N + 1] = m_obs;
      B.dbg[DBG_W * (size_t)chunk + DBG_N + 2] = m_anc;
    }
  }
}

// ---- landmarks with more than 64 observations -------------------------------------------------------------------
// The reference adds an observation edge for every window pose in a point's vis_set, without a cap (slam_graph.cpp:1001-1027):
// in a loop a point is easily seen from more than 64 of the 230 window poses.  Such a landmark does not fit the
// one-wave-per-chunk kernel above; it gets ONE WORKGROUP of WIDE_THREADS lanes (lane = edge, up to WIDE_THREADS observations =
// more than the 256 poses a window can hold).  Same algebra as ba_landmark_kernel (compact edge blocks, per-landmark sums of
// 27 values, 3x3 closed-form inverse, circulant observer pairs with W_obs parked in LDS); sums go through the workgroup,
// blocks straight to global atomics -- these landmarks are few.
constexpr int WIDE_THREADS = 256;
template <int N>
__device__ __forceinline__ void wide_block_sum(double (&v)[N], double *s_red /* [WIDE_THREADS / 64][N] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double w = wave_sum_f64(v[i]);
    if (lane == 0) s_red[wave * N + i] = w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double t = 0;
#pragma unroll
    for (int w = 0; w < WIDE_THREADS / 64; ++w) t += s_red[w * N + i];
    v[i] = t;
  }
  __syncthreads();
}
template <int MODE>
__global__ __launch_bounds__(WIDE_THREADS) void ba_wide_landmark_kernel(BaDev B) {
  if (B.ctl) { if (B.ctl[1] != 0.0) return; B.lambda = B.ctl[0]; }
  __shared__ double s_red[(WIDE_THREADS / 64) * 27];
  __shared__ __attribute__((aligned(16))) double s_wo[MODE == 0 ? WIDE_THREADS * 18 : 1];
  __shared__ int s_pose[WIDE_THREADS], s_role[WIDE_THREADS];
  const int tid = threadIdx.x;
  // MODE 0: WIDE_SPLIT workgroups per landmark.  Every one of them forms the landmark's sums, D^-1 and the W blocks (cheap); the
  // landmark's own blocks (chi2, anchor, observer diagonal / anchor coupling, b) are added by part 0, the m (m - 1) / 2 observer pairs
  // -- 16 000 blocks of 36 global atomics at m = 180, 0.9 ms when one workgroup walked them -- are dealt round by round to the parts
  const int split = MODE == 0 ? B.wide_split : 1, lmi = blockIdx.x / split, part = blockIdx.x - lmi * split;
  const int e0 = B.wide_start[lmi], m = B.wide_len[lmi];
  const bool active = tid < m;
  svs_ba_edge ed;
  if (active) ed = B.edges[e0 + tid];
  else { ed.point = -1; ed.pose = 0; ed.anchor = 0; }
  {                                                     // every edge of the landmark has the same anchor / point: inactive lanes take
    __shared__ int s_ap[2];                             // them from the first edge
    if (tid == 0) { s_ap[0] = ed.anchor; s_ap[1] = ed.point; }
    __syncthreads();
    ed.anchor = s_ap[0]; if (!active) ed.point = s_ap[1];
  }
  const int anchor = ed.anchor;
  double psi[3], To[12], Ta[12];
  EdgeCore lin;
#pragma unroll
  for (int i = 0; i < 3; ++i) psi[i] = B.psi[3 * (size_t)ed.point + i];
#pragma unroll
  for (int i = 0; i < 12; ++i) { To[i] = B.poses[12 * (size_t)ed.pose + i]; Ta[i] = B.poses[12 * (size_t)anchor + i]; }
  linearize_edge(psi, To, Ta, ed, B.cam, B.delta, B.robust, lin);       // inactive lanes: a copy of a valid geometry, masked below
  const bool self = active && ed.pose == anchor;
  const bool obs_role = active && !self;
  const bool self_lit = self && B.self_mode == 0;
  if (!active) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { lin.A[i] = 0; }
#pragma unroll
    for (int i = 0; i < 3; ++i) lin.g[i] = 0;
    lin.rho0 = 0;
  }
  double Bm[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Bm[3 * i + j] = lin.A[3 * i] * lin.D[j] + lin.A[3 * i + 1] * lin.D[3 + j] + lin.A[3 * i + 2] * lin.D[6 + j];
  constexpr int NRED = MODE == 0 ? 27 : 18;
  double red[27];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) red[k++] = lin.D[i] * Bm[j] + lin.D[3 + i] * Bm[3 + j] + lin.D[6 + i] * Bm[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) red[6 + i] = -(lin.D[i] * lin.g[0] + lin.D[3 + i] * lin.g[1] + lin.D[6 + i] * lin.g[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) red[9 + 3 * i + j] = obs_role ? lin.R[i] * Bm[j] + lin.R[3 + i] * Bm[3 + j] + lin.R[6 + i] * Bm[6 + j] : 0.0;
    double AR[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) AR[3 * i + j] = lin.A[3 * i] * lin.R[j] + lin.A[3 * i + 1] * lin.R[3 + j] + lin.A[3 * i + 2] * lin.R[6 + j];
    const bool in_maa = obs_role || self_lit;
    k = 18;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) red[k++] = in_maa ? lin.R[i] * AR[j] + lin.R[3 + i] * AR[3 + j] + lin.R[6 + i] * AR[6 + j] : 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) red[24 + i] = obs_role ? lin.R[i] * lin.g[0] + lin.R[3 + i] * lin.g[1] + lin.R[6 + i] * lin.g[2] : 0.0;
  }
  (void)NRED;
  wide_block_sum<27>(red, s_red);
  double Di[9], bl[3] = {red[6], red[7], red[8]};
  {
    const double a00 = red[0] + B.lambda, a01 = red[1], a02 = red[2], a11 = red[3] + B.lambda, a12 = red[4], a22 = red[5] + B.lambda;
    const double c00 = a11 * a22 - a12 * a12, c01 = a12 * a02 - a01 * a22, c02 = a01 * a12 - a11 * a02;
    const double id = 1.0 / (a00 * c00 + a01 * c01 + a02 * c02);
    Di[0] = c00 * id; Di[1] = c01 * id; Di[2] = c02 * id;
    Di[3] = Di[1]; Di[4] = (a00 * a22 - a02 * a02) * id; Di[5] = (a02 * a01 - a00 * a12) * id;
    Di[6] = Di[2]; Di[7] = Di[5]; Di[8] = (a00 * a11 - a01 * a01) * id;
  }
  double Db[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) Db[i] = Di[3 * i] * bl[0] + Di[3 * i + 1] * bl[1] + Di[3 * i + 2] * bl[2];
  double Wo[18];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double t0, t1, t2;
    cross3(lin.y, Bm[c], Bm[3 + c], Bm[6 + c], t0, t1, t2);
    Wo[c] = obs_role ? Bm[c] : 0.0; Wo[3 + c] = obs_role ? Bm[3 + c] : 0.0; Wo[6 + c] = obs_role ? Bm[6 + c] : 0.0;
    Wo[9 + c] = obs_role ? t0 : 0.0; Wo[12 + c] = obs_role ? t1 : 0.0; Wo[15 + c] = obs_role ? t2 : 0.0;
  }
  double WA[18];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double t0, t1, t2;
    cross3(lin.xa, red[9 + c], red[12 + c], red[15 + c], t0, t1, t2);
    WA[c] = -red[9 + c]; WA[3 + c] = -red[12 + c]; WA[6 + c] = -red[15 + c];
    WA[9 + c] = -t0; WA[12 + c] = -t1; WA[15 + c] = -t2;
  }
  if (MODE == 1) {
    double c[3] = {0, 0, 0};
    if (obs_role) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) c[j] += Wo[3 * i + j] * B.x[6 * ed.pose + i];
    }
    wide_block_sum<3>(c, s_red);
    double xl[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double sj = bl[j] - c[j];
#pragma unroll
      for (int i = 0; i < 6; ++i) sj -= WA[3 * i + j] * B.x[6 * anchor + i];
      c[j] = sj;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) xl[i] = Di[3 * i] * c[0] + Di[3 * i + 1] * c[1] + Di[3 * i + 2] * c[2];
    const double npsi[3] = {psi[0] + xl[0], psi[1] + xl[1], psi[2] + xl[2]};
    double sc[2] = {0, 0};      // chi2 at the trial state, x_l (lambda x_l + b_l)
    if (tid == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { B.psi_trial[3 * (size_t)ed.point + i] = npsi[i]; sc[1] += xl[i] * (B.lambda * xl[i] + bl[i]); }
    }
    if (active) {
      double Tno[12], Tna[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) { Tno[i] = B.poses_trial[12 * (size_t)ed.pose + i]; Tna[i] = B.poses_trial[12 * (size_t)anchor + i]; }
      sc[0] = edge_chi2(npsi, Tno, Tna, ed, B.cam, B.delta, B.robust);
    }
    wide_block_sum<2>(sc, s_red);
    if (tid == 0) { atomic_add_f64(&B.scal[SC_CHI + (blockIdx.x & (SC_SLOTS - 1))], sc[0]); atomic_add_f64(&B.scal[SC_SCL + (blockIdx.x & (SC_SLOTS - 1))], sc[1]); }
    return;
  }
  // ---- MODE 0 ----
  if (part == 0) {            // (workgroup-uniform)
    double c1[1] = {lin.rho0};
    wide_block_sum<1>(c1, s_red);
    if (tid == 0 && c1[0] != 0.0) atomic_add_f64(B.chi2_cur + (blockIdx.x & (SC_SLOTS - 1)), c1[0]);
  }
  auto add_blk = [&](int pi, int pj, int rc, double v) { atomic_add_f64(&B.H[blk_index(pi, pj, B.P) * 36 + rc], v); };
  if (tid == 0 && part == 0) {             // anchor block, once: Ea^T S_RAR Ea - (W_A D^-1) W_A^T, b_anc, Schur rhs
    double WAD[18], SR[9], Maa[36];
    SR[0] = red[18]; SR[1] = red[19]; SR[2] = red[20]; SR[3] = red[19]; SR[4] = red[21]; SR[5] = red[22]; SR[6] = red[20]; SR[7] = red[22]; SR[8] = red[23];
    sym_block(SR, lin.xa, Maa);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) WAD[3 * i + j] = WA[3 * i] * Di[j] + WA[3 * i + 1] * Di[3 + j] + WA[3 * i + 2] * Di[6 + j];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = r; c < 6; ++c)
        add_blk(anchor, anchor, 6 * r + c, Maa[6 * r + c] - (WAD[3 * r] * WA[3 * c] + WAD[3 * r + 1] * WA[3 * c + 1] + WAD[3 * r + 2] * WA[3 * c + 2]));
    double t0, t1, t2;
    cross3(lin.xa, red[24], red[25], red[26], t0, t1, t2);
    const double ba[6] = {red[24], red[25], red[26], t0, t1, t2};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      atomic_add_f64(B.bp + 6 * anchor + r, ba[r]);
      atomic_add_f64(B.bs + 6 * anchor + r, WA[3 * r] * Db[0] + WA[3 * r + 1] * Db[1] + WA[3 * r + 2] * Db[2]);
    }
  }
  double *my_wo = s_wo + tid * 18;
#pragma unroll
  for (int i = 0; i < 18; ++i) my_wo[i] = Wo[i];
  s_pose[tid] = ed.pose; s_role[tid] = obs_role ? 1 : 0;
  double WoD[18];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) WoD[3 * i + j] = Wo[3 * i] * Di[j] + Wo[3 * i + 1] * Di[3 + j] + Wo[3 * i + 2] * Di[6 + j];
  if (obs_role && part == 0) {
    const int pi = ed.pose;
    {
      double Moo[36];
      sym_block(lin.A, lin.y, Moo);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c)
          add_blk(pi, pi, 6 * r + c, Moo[6 * r + c] - (WoD[3 * r] * Wo[3 * c] + WoD[3 * r + 1] * Wo[3 * c + 1] + WoD[3 * r + 2] * Wo[3 * c + 2]));
    }
    {
      double AR[9], Noa[36];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) AR[3 * i + j] = lin.A[3 * i] * lin.R[j] + lin.A[3 * i + 1] * lin.R[3 + j] + lin.A[3 * i + 2] * lin.R[6 + j];
      cross_block(AR, lin.y, lin.xa, Noa);
      const bool up = pi < anchor;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double mm = -Noa[6 * r + c] - (WoD[3 * r] * WA[3 * c] + WoD[3 * r + 1] * WA[3 * c + 1] + WoD[3 * r + 2] * WA[3 * c + 2]);
          if (up) add_blk(pi, anchor, 6 * r + c, mm); else add_blk(anchor, pi, 6 * c + r, mm);
        }
    }
    double u0, u1, u2;
    cross3(lin.y, lin.g[0], lin.g[1], lin.g[2], u0, u1, u2);
    const double bo[6] = {-lin.g[0], -lin.g[1], -lin.g[2], -u0, -u1, -u2};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      atomic_add_f64(B.bp + 6 * pi + r, bo[r]);
      atomic_add_f64(B.bs + 6 * pi + r, Wo[3 * r] * Db[0] + Wo[3 * r + 1] * Db[1] + Wo[3 * r + 2] * Db[2]);
    }
  }
  __syncthreads();
  // observer-observer pairs, circulant schedule over the m edges of the landmark (partners may sit in other waves: pose / role /
  // W_obs come from LDS)
  for (int r = 1 + part; r <= (m >> 1); r += split) {
    int b = tid + r;
    if (b >= m) b -= m;
    const bool on = obs_role && 2 * r <= m && !(2 * r == m && tid >= r) && s_role[b] != 0;
    if (on) {
      const int pj = s_pose[b];
      const double *wj = s_wo + b * 18;
      double Wj[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) Wj[i] = wj[i];
      const bool up = ed.pose < pj;
#pragma unroll
      for (int rr = 0; rr < 6; ++rr)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double v = -(WoD[3 * rr] * Wj[3 * c] + WoD[3 * rr + 1] * Wj[3 * c + 1] + WoD[3 * rr + 2] * Wj[3 * c + 2]);
          if (up) add_blk(ed.pose, pj, 6 * rr + c, v); else add_blk(pj, ed.pose, 6 * c + rr, v);
        }
    }
  }
}

// ---- reduced-system solve: blocked right-looking Cholesky on the packed upper 6x6 blocks -------
// One workgroup (1024 lanes); the factor overwrites H in L2-resident global memory, the current
// panel row and the rhs live in LDS.  A = U^T U, U upper.  Fused forward substitution; column
// oriented back substitution.  Then T_trial = exp(x_p) T, scale_p, bookkeeping scalars.
constexpr int SOLVE_THREADS = 256;
constexpr int GRID_NBAR = 16;        // arrival counters of the multi-workgroup solve (bar[32 i], failure flag at bar[32 GRID_NBAR])
constexpr int SOLVE_MAX_P = 256;

__device__ void d_se3_exp_mul(const double *x, const double *T, double *Tn) {   // exp(x) * T  (G2oVertexSE3::oplusImpl)
  const double *w = x + 3;
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
  double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, W2[9], R[9], V[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
  double a, b;
  if (th < 1e-10) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; } else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
  for (int i = 0; i < 9; ++i) R[i] = a * W[i] + b * W2[i];
  R[0] += 1; R[4] += 1; R[8] += 1;
  if (th < 1e-10) { for (int i = 0; i < 9; ++i) V[i] = R[i]; }
  else {
    const double c = (1.0 - cos(th)) / th2, d = (th - sin(th)) / (th2 * th);
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

// rowmax[k] = last block column of row k inside the (filled) block envelope of the reduced system:
// structurally H_kj == 0 and stays 0 during elimination for j > rowmax[k].  For a dense window
// rowmax[k] = P-1 and this is a plain blocked Cholesky; for the banded co-visibility structure of a
// sliding window it skips almost all of the P^3/6 block updates (what CSparse's sparse Cholesky
// does for the reference).  colmin[k] = first row whose envelope reaches column k.
__global__ __launch_bounds__(SOLVE_THREADS) void ba_solve_kernel(BaDev B, double *__restrict__ x_out, double *__restrict__ linv_ws,
                                                                 const int *__restrict__ rowmax, const int *__restrict__ colmin) {
  extern __shared__ double smem[];
  if (B.ctl) { if (B.ctl[1] != 0.0) return; B.lambda = B.ctl[0]; }      // speculative trial: lambda decided on the device, skipped after a rejection
  const int P = B.P, n = 6 * P, tid = threadIdx.x;
  double *s_b = smem;                 // [n] rhs -> y -> x
  double *s_panel = smem + n;         // [P*36] current panel row U_kj, j>k
  double *s_linv = s_panel + (size_t)P * 36;   // [36] (U_kk^T)^-1, lower
  __shared__ int s_fail;
  if (tid == 0) s_fail = 0;
  for (int i = tid; i < n; i += SOLVE_THREADS) s_b[i] = B.bp[i] - B.bs[i];
  __syncthreads();
  for (int k = 0; k < P; ++k) {
    const long kk = blk_index(k, k, P);
    if (tid == 0) {
      // 6x6 Cholesky of A_kk (+lambda), upper stored; fully unrolled so A/U/Li stay in registers
      double A[36], U[36], Li[36];
      const double *Akk = B.H + kk * 36;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) { double v = Akk[6 * r + c]; if (r == c) v += B.lambda; A[6 * r + c] = v; }
      int fail = 0;
#pragma unroll
      for (int i = 0; i < 36; ++i) { U[i] = 0; Li[i] = 0; }
      double rd[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
#pragma unroll
        for (int q = 0; q < j; ++q) d -= U[6 * q + j] * U[6 * q + j];
        if (!(d > 0) || !isfinite(d)) { fail = 1; d = 1; }
        d = sqrt(d);
        U[6 * j + j] = d;
        rd[j] = 1.0 / d;
#pragma unroll
        for (int c = j + 1; c < 6; ++c) {
          double s = A[6 * j + c];
#pragma unroll
          for (int q = 0; q < j; ++q) s -= U[6 * q + j] * U[6 * q + c];
          U[6 * j + c] = s * rd[j];
        }
      }
      // Li = (U^T)^-1 (lower triangular): forward substitution on identity
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int r = c; r < 6; ++r) {
          double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
          for (int q = c; q < r; ++q) s -= U[6 * q + r] * Li[6 * q + c];
          Li[6 * r + c] = s * rd[r];
        }
#pragma unroll
      for (int i = 0; i < 36; ++i) { s_linv[i] = Li[i]; linv_ws[(size_t)k * 36 + i] = Li[i]; }
      if (fail) s_fail = 1;
    }
    __syncthreads();
    if (s_fail) break;
    // y_k = Li * b_k (forward substitution, fused)
    double yk[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) { double s = 0; for (int q = 0; q <= r; ++q) s += s_linv[6 * r + q] * s_b[6 * k + q]; yk[r] = s; }
    // panel: U_kj = Li * A_kj, k < j <= rowmax[k]; one thread per (block, column)
    const int nj = rowmax[k] - k;
    for (int e = tid; e < nj * 6; e += SOLVE_THREADS) {
      const int jj = e / 6, c = e - jj * 6;
      const double *Akj = B.H + (kk + 1 + jj) * 36;
      double a[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) a[q] = Akj[6 * q + c];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double s = 0;
#pragma unroll
        for (int q = 0; q <= r; ++q) s += s_linv[6 * r + q] * a[q];
        s_panel[(size_t)jj * 36 + 6 * r + c] = s;
      }
    }
    __syncthreads();
    if (tid < 6) s_b[6 * k + tid] = yk[tid];
    for (int e = tid; e < nj * 36; e += SOLVE_THREADS) B.H[(kk + 1) * 36 + e] = s_panel[e];   // keep U for back substitution
    // rhs update: b_j -= U_kj^T y_k
    for (int e = tid; e < nj * 6; e += SOLVE_THREADS) {
      const int jj = e / 6, c = e - jj * 6;
      double s = 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) s += s_panel[(size_t)jj * 36 + 6 * q + c] * yk[q];
      s_b[6 * (k + 1 + jj) + c] -= s;
    }
    // trailing update inside the envelope: A_ij -= U_ki^T U_kj, k < i <= j <= rowmax[k];
    // one thread per (block, row): 6 outputs from 6 + 36 LDS operands
    const long nblk = (long)nj * (nj + 1) / 2;
    for (long e = tid; e < nblk * 6; e += SOLVE_THREADS) {
      const long bidx = e / 6;
      const int r = (int)(e - bidx * 6);
      int ii = (int)((2.0 * nj + 1.0 - sqrt((2.0 * nj + 1.0) * (2.0 * nj + 1.0) - 8.0 * (double)bidx)) * 0.5);
      while ((long)ii * nj - (long)ii * (ii - 1) / 2 > bidx) --ii;
      while ((long)(ii + 1) * nj - (long)(ii + 1) * ii / 2 <= bidx) ++ii;
      const int jj = ii + (int)(bidx - ((long)ii * nj - (long)ii * (ii - 1) / 2));
      double xi[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) xi[q] = s_panel[(size_t)ii * 36 + 6 * q + r];
      double *Aij = B.H + blk_index(k + 1 + ii, k + 1 + jj, P) * 36 + 6 * r;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double s = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) s += xi[q] * s_panel[(size_t)jj * 36 + 6 * q + c];
        Aij[c] -= s;
      }
    }
    __syncthreads();
  }
  const int fail = s_fail;
  __syncthreads();
  if (!fail) {
    // back substitution (column oriented): x_k = Li_k^T y_k ; y_i -= U_ik x_k for colmin[k] <= i < k
    for (int k = P - 1; k >= 0; --k) {
      if (tid < 6) {
        double s = 0;
        for (int q = tid; q < 6; ++q) s += linv_ws[(size_t)k * 36 + 6 * q + tid] * s_b[6 * k + q];
        s_linv[tid] = s;
      }
      __syncthreads();
      if (tid < 6) s_b[6 * k + tid] = s_linv[tid];
      const int i0 = colmin[k];
      for (int e = tid; e < (k - i0) * 6; e += SOLVE_THREADS) {
        const int i = i0 + e / 6, r = e % 6;
        if (rowmax[i] >= k) {
          const double *Uik = B.H + blk_index(i, k, P) * 36;
          double s = 0;
#pragma unroll
          for (int q = 0; q < 6; ++q) s += Uik[6 * r + q] * s_linv[q];
          s_b[6 * i + r] -= s;
        }
      }
      __syncthreads();
    }
  } else {
    for (int i = tid; i < n; i += SOLVE_THREADS) s_b[i] = 0;
    __syncthreads();
  }
  // outputs: x, scale_p = sum x (lambda x + b_p), trial poses
  double sc = 0;
  for (int i = tid; i < n; i += SOLVE_THREADS) { const double xv = s_b[i]; x_out[i] = xv; sc += xv * (B.lambda * xv + B.bp[i]); }
  sc = wave_sum_f64(sc);
  if ((tid & 63) == 0 && sc != 0.0) atomic_add_f64(&B.scal[2], sc);
  for (int p = tid; p < P; p += SOLVE_THREADS) {
    double Tn[12];
    d_se3_exp_mul(s_b + 6 * p, B.poses + 12 * (size_t)p, Tn);
    for (int i = 0; i < 12; ++i) B.poses_trial[12 * (size_t)p + i] = Tn[i];
  }
  if (tid == 0) { B.scal[3] = (double)fail; B.scal[4] = chi2_cur_sum(B); }
}

// ---- multi-workgroup variant for wide envelopes -------------------------------------------------------------------------
// The reference's real window is 30 inner + 200 outer poses (data/newcollege.cfg:21-22), and a loop closure or a landmark seen from
// 180 keyframes couples poses far apart: the filled envelope then spans most of the 230 block rows and the one-workgroup kernel
// above spends 0.26 ms per elimination step on its trailing update (60 ms per solve).  Here the trailing update of every step is
// spread over G workgroups (one per CU):
//   * every workgroup factorises the 6x6 pivot block and forms the WHOLE panel row U_k* = L_kk^-1 A_k* itself (<= 255 blocks,
//     50 kFLOP: cheaper than handing it around), into LDS;
//   * the trailing tiles A_ij -= U_ki^T U_kj of the step are dealt to all lanes of all workgroups (one lane per tile row);
//   * ONE device-scope arrival counter per step (monotonic, relaxed polls) makes row k+1 final before anybody reads it.  All
//     traffic on H goes through write-through stores / L1-bypassing loads (agent scope), so no cache has to be flushed or
//     invalidated and workgroups on different XCDs see each other's tiles (MI355X_MICROARCH.md, inter-workgroup visibility);
//   * workgroup 0 keeps the right-hand side (forward substitution rides along), stores U for the back substitution and finishes
//     alone: row-oriented back substitution, trial poses, scale.
// The G workgroups must be resident together (G <= #CUs, one 256-lane workgroup each); a bounded wait turns a missing sibling into
// the solve's failure path (the LM trial is then rejected like a non-positive pivot).
__device__ __forceinline__ void g_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double g_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double bcast_f64(double v, int l) {      // value of lane l (wave-uniform l)
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__global__ __launch_bounds__(SOLVE_THREADS) void ba_solve_grid_kernel(BaDev B, double *__restrict__ x_out, double *__restrict__ linv_ws,
                                                                      const int *__restrict__ rowmax, unsigned *__restrict__ bar, unsigned epoch0) {
  extern __shared__ double smem[];
  if (B.ctl) { if (B.ctl[1] != 0.0) return; B.lambda = B.ctl[0]; }
  const int P = B.P, n = 6 * P, tid = threadIdx.x, wg = blockIdx.x, G = gridDim.x;
  double *s_panel = smem;                          // [P*36] panel row U_kj, j > k
  double *s_linv = s_panel + (size_t)P * 36;       // [36] (U_kk^T)^-1, lower
  double *s_b = s_linv + 36;                       // [n] rhs -> y -> x (workgroup 0 only)
  __shared__ int s_fail;
  if (tid == 0) s_fail = 0;
  if (wg == 0) for (int i = tid; i < n; i += SOLVE_THREADS) s_b[i] = B.bp[i] - B.bs[i];
  __syncthreads();
  // Everything a step reads from H -- the pivot block, the panel row, this lane's trailing tiles -- was completed by the previous step's
  // arrival, and every such load bypasses the caches (other workgroups wrote the data) at the price of a memory round trip of ~2 us.
  // So a step issues ALL of its loads first and works from registers: one round trip per step instead of three dependent ones
  // (pivot block -> panel row -> trailing tiles; 13.5 -> 9 us per block row at 224 block columns).
  constexpr int NIT = (SOLVE_MAX_P * 6 + SOLVE_THREADS - 1) / SOLVE_THREADS;       // panel columns per lane
  long long acc_t[6] = {0, 0, 0, 0, 0, 0};      // workgroup 0, lane 0: where the time of a step goes (SVS_BA_DEBUG=1)
  for (int k = 0; k < P; ++k) {
    long long ts = wall_clock64();
#define GRID_LAP(i) do { const long long tn = wall_clock64(); acc_t[i] += tn - ts; ts = tn; } while (0)
    const long kk = blk_index(k, k, P);
    const int nj = rowmax[k] - k;
    const long nblk = (long)nj * (nj + 1) / 2, stride = (long)G * SOLVE_THREADS;
    // ---- loads: pivot block (lane 0), panel columns, the first three trailing tile rows of this lane
    double Acol[6] = {0, 0, 0, 0, 0, 0};         // lane c < 6: column c of the pivot block's upper triangle
    if (tid < 6) {
      const double *Akk = B.H + kk * 36;
#pragma unroll
      for (int r = 0; r < 6; ++r)
        if (r <= tid) { double v = g_ld(Akk + 6 * r + tid); if (r == tid) v += B.lambda; Acol[r] = v; }
    }
    double a[NIT][6];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * SOLVE_THREADS;
      if (e < nj * 6) {
        const int jj = e / 6, c = e - jj * 6;
        const double *Akj = B.H + (kk + 1 + jj) * 36;
#pragma unroll
        for (int q = 0; q < 6; ++q) a[it][q] = g_ld(Akj + 6 * q + c);
      }
    }
    double *Aij[3];
    int ti[3], tj[3], tr[3];
    double acc[3][6];
    auto tile_of = [&](long e, int u) {      // tile row e = (block (ii, jj) of the trailing triangle, row r): index + loads
      Aij[u] = nullptr;
      if (e < nblk * 6) {
        const long bidx = e / 6;
        tr[u] = (int)(e - bidx * 6);
        int i_ = (int)((2.0 * nj + 1.0 - sqrt((2.0 * nj + 1.0) * (2.0 * nj + 1.0) - 8.0 * (double)bidx)) * 0.5);
        while ((long)i_ * nj - (long)i_ * (i_ - 1) / 2 > bidx) --i_;
        while ((long)(i_ + 1) * nj - (long)(i_ + 1) * i_ / 2 <= bidx) ++i_;
        ti[u] = i_;
        tj[u] = i_ + (int)(bidx - ((long)i_ * nj - (long)i_ * (i_ - 1) / 2));
        Aij[u] = B.H + blk_index(k + 1 + ti[u], k + 1 + tj[u], P) * 36 + 6 * tr[u];
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[u][c] = g_ld(Aij[u] + c);
      }
    };
    const long e_first = (long)wg * SOLVE_THREADS + tid;
#pragma unroll
    for (int u = 0; u < 3; ++u) tile_of(e_first + u * stride, u);
    // ---- pivot block: U_kk and (U_kk^T)^-1 in every workgroup (cheaper than handing them around), lane c of wave 0 = column c,
    //      the elements of other columns by readlane: six short steps instead of ~600 dependent instructions on one lane
    if (tid < 64) {
      const int c = tid;
      double Ucol[6] = {0, 0, 0, 0, 0, 0}, rdv[6];      // U[q][c], q <= c
      int fail = 0;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double sv = Acol[j];                              // A[j][c], meaningful for c >= j
#pragma unroll
        for (int q = 0; q < j; ++q) sv -= bcast_f64(Ucol[q], j) * Ucol[q];      // U[q][j] U[q][c]
        double d = bcast_f64(sv, j);
        if (!(d > 0) || !isfinite(d)) { fail = 1; d = 1; }
        d = sqrt(d);
        rdv[j] = 1.0 / d;
        Ucol[j] = c == j ? d : (c > j ? sv * rdv[j] : 0.0);
      }
      double Lic[6];                                      // (U^T)^-1 [r][c], r >= c
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double sv = (r == c) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < r; ++q) sv -= bcast_f64(Ucol[q], r) * Lic[q];      // U[q][r] Li[q][c] (Li[q][c] = 0 for q < c)
        Lic[r] = r >= c ? sv * rdv[r] : 0.0;
      }
      if (c < 6) {
#pragma unroll
        for (int r = 0; r < 6; ++r) s_linv[6 * r + c] = Lic[r];
        if (wg == 0) {
#pragma unroll
          for (int r = 0; r < 6; ++r) linv_ws[(size_t)k * 36 + 6 * r + c] = Lic[r];
        }
      }
      if (fail && c == 0) s_fail = 1;
    }
    __syncthreads();
    GRID_LAP(0);
    if (s_fail) break;                             // every workgroup takes the same decision from the same pivot block
    // ---- panel: U_kj = Li * A_kj, one lane per (block, column)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * SOLVE_THREADS;
      if (e < nj * 6) {
        const int jj = e / 6, c = e - jj * 6;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double sv = 0;
#pragma unroll
          for (int q = 0; q <= r; ++q) sv += s_linv[6 * r + q] * a[it][q];
          s_panel[(size_t)jj * 36 + 6 * r + c] = sv;
        }
      }
    }
    __syncthreads();
    if (wg == 0) {
      double yk[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) { double sv = 0; for (int q = 0; q <= r; ++q) sv += s_linv[6 * r + q] * s_b[6 * k + q]; yk[r] = sv; }
      __syncthreads();
      if (tid < 6) s_b[6 * k + tid] = yk[tid];
      for (int e = tid; e < nj * 6; e += SOLVE_THREADS) {
        const int jj = e / 6, c = e - jj * 6;
        double sv = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) sv += s_panel[(size_t)jj * 36 + 6 * q + c] * yk[q];
        s_b[6 * (k + 1 + jj) + c] -= sv;
      }
    }
    GRID_LAP(1);
    // ---- trailing update inside the envelope, dealt to all lanes of all workgroups: one lane per (tile, row), three per trip
    for (long e0 = e_first;; e0 += 3 * stride) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (Aij[u]) {
          double xi[6];
#pragma unroll
          for (int q = 0; q < 6; ++q) xi[q] = s_panel[(size_t)ti[u] * 36 + 6 * q + tr[u]];
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            double sv = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) sv += xi[q] * s_panel[(size_t)tj[u] * 36 + 6 * q + c];
            g_st(Aij[u] + c, acc[u][c] - sv);
          }
        }
      }
      if (e0 + 3 * stride >= nblk * 6) break;
#pragma unroll
      for (int u = 0; u < 3; ++u) tile_of(e0 + (3 + u) * stride, u);      // (only windows with few workgroups get here)
    }
    // arrival: this workgroup's tiles of step k are on their way; wait until everybody's are
    GRID_LAP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GRID_LAP(3);
    // GRID_NBAR counters, 128 bytes apart, workgroup w arrives on counter w % GRID_NBAR: 256 same-address device-scope atomics per step
    // serialise at the memory side (3 of the 16 us a step took); 16 per counter do not.  Wave 0 polls, one counter per lane.
    if (tid < 64 && k + 1 < P) {
      if (tid == 0) __hip_atomic_fetch_add(bar + 32 * (wg % GRID_NBAR), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool mine = tid < GRID_NBAR && tid < G;
      const unsigned n_here = mine ? (unsigned)((G - tid + GRID_NBAR - 1) / GRID_NBAR) : 0u;      // workgroups that arrive on this lane's counter
      const unsigned want = epoch0 + n_here * (unsigned)(k + 1);
      long spin = 0;
      bool late = false;
      for (;;) {
        const unsigned v = mine ? __hip_atomic_load(bar + 32 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
        if (__all((int)(v - want) >= 0)) break;
        if (++spin >= (1l << 22)) { late = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (tid == 0) {
        if (late) __hip_atomic_store(bar + 32 * GRID_NBAR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(bar + 32 * GRID_NBAR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) s_fail = 2;
      }
    }
    __syncthreads();
    GRID_LAP(4);
    if (s_fail) break;
    // U_k* for the back substitution replaces A_k* only now: before the arrival above a slower workgroup could still be reading A_k*
    // for its own copy of the panel row
    if (wg == 0) {
      for (int e = tid; e < nj * 36; e += SOLVE_THREADS) g_st(B.H + (kk + 1) * 36 + e, s_panel[e]);
      __syncthreads();                              // s_panel is overwritten by the next step's panel
    }
    GRID_LAP(5);
  }
  if (wg != 0) return;
  const int fail = s_fail;
  __syncthreads();
  if (!fail) {
    // back substitution, row oriented: x_k = Li_k^T (y_k - sum_{j > k} U_kj x_j); the row of U is contiguous in the packed storage
    __shared__ double s_acc[SOLVE_THREADS / 64][6];
    for (int k = P - 1; k >= 0; --k) {
      const int nj = rowmax[k] - k;
      const double *Uk = B.H + (blk_index(k, k, P) + 1) * 36;
      double part[6] = {0, 0, 0, 0, 0, 0};
      for (int e = tid; e < nj * 6; e += SOLVE_THREADS) {      // thread = (block jj, column c): contributes U_kj[r][c] x_j[c] to every row r
        const int jj = e / 6, c = e - jj * 6;
        const double xv = s_b[6 * (k + 1 + jj) + c];
#pragma unroll
        for (int r = 0; r < 6; ++r) part[r] += Uk[(size_t)jj * 36 + 6 * r + c] * xv;      // plain loads: every U row was written by THIS workgroup (write-through), nothing of it sits in this CU's L1
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) part[r] = wave_sum_f64(part[r]);
      if ((tid & 63) == 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) s_acc[tid >> 6][r] = part[r];
      }
      __syncthreads();
      if (tid < 6) {
        double yv[6];
        for (int q = 0; q < 6; ++q) { double t = s_b[6 * k + q]; for (int w = 0; w < SOLVE_THREADS / 64; ++w) t -= s_acc[w][q]; yv[q] = t; }
        double sv = 0;
        for (int q = tid; q < 6; ++q) sv += linv_ws[(size_t)k * 36 + 6 * q + tid] * yv[q];
        s_linv[tid] = sv;
      }
      __syncthreads();
      if (tid < 6) s_b[6 * k + tid] = s_linv[tid];
      __syncthreads();
    }
  } else {
    for (int i = tid; i < n; i += SOLVE_THREADS) s_b[i] = 0;
    __syncthreads();
  }
  double sc = 0;
  for (int i = tid; i < n; i += SOLVE_THREADS) { const double xv = s_b[i]; x_out[i] = xv; sc += xv * (B.lambda * xv + B.bp[i]); }
  sc = wave_sum_f64(sc);
  if ((tid & 63) == 0 && sc != 0.0) atomic_add_f64(&B.scal[2], sc);
  for (int p = tid; p < P; p += SOLVE_THREADS) {
    double Tn[12];
    d_se3_exp_mul(s_b + 6 * p, B.poses + 12 * (size_t)p, Tn);
    for (int i = 0; i < 12; ++i) B.poses_trial[12 * (size_t)p + i] = Tn[i];
  }
  if (tid == 0) {
    B.scal[3] = (double)fail; B.scal[4] = chi2_cur_sum(B);
    // loads + pivot | panel + rhs | trailing issue | store drain | arrival | write-back  (read as: init, forward, load+row update, eliminate, -, barrier wait)
    B.scal[5] = acc_t[0] * 0.01; B.scal[6] = acc_t[1] * 0.01; B.scal[8] = acc_t[2] * 0.01; B.scal[9] = acc_t[3] * 0.01; B.scal[10] = acc_t[5] * 0.01; B.scal[11] = acc_t[4] * 0.01; B.scal[7] = 0;
  }
}

// ---- LDS-window variant of the solve -------------------------------------------------------------
// When the filled block envelope is narrow (R = max_k(rowmax[k]-k)+1 rows fit in LDS) the active
// R x R block window of the elimination lives entirely in LDS as a ring of envelope rows, and the
// P-step elimination runs as a software pipeline with ONE workgroup barrier per step:
//   wave 0 (pivot wave)   stage k: applies panel k to block row k+1 and to b_{k+1}, then factors the
//                         pivot block k+1 and forms panel k+1 (Z, Y; fused forward substitution)
//   waves 1-4 (update)    stage k: applies panel k to block rows k+2.. and to the rest of b
//   wave 5 (loader)       stage k: drops envelope row k+R (loaded two stages earlier) into the ring
//                         slot row k vacated, issues the loads of row k+R+2
// Panels are double-buffered, so panel k+1 is written while panel k is still being applied.  The
// panel rows Y_kj also go to a global buffer (written once, read once by the back substitution).
// The back substitution is run by wave 0 alone -- no workgroup barriers on its P-step chain.
// Measured (tools/ubench.hip): an f64 FMA issues every ~5 cycles per wave, dependent or not; what is
// expensive on the critical path is LDS round trips and workgroup barriers, so those are minimised.

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which would put
// the latency of the loader's global loads (~1 us) back on the critical path of every step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// orders LDS traffic between the lanes of one wave (LDS executes a wave's operations in order)
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
constexpr int PIPE_THREADS = 384;
constexpr int PIPE_LD = 16;     // loader registers per lane and buffer: R*36 <= 64*PIPE_LD  => R <= 28

// reciprocal by hardware estimate + 2 Newton steps
__device__ __forceinline__ double rcp_nr(double d) {
  double x = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, x, 1.0);
  x = __builtin_fma(x, e, x);
  e = __builtin_fma(-d, x, 1.0);
  return __builtin_fma(x, e, x);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// H + lambda I = U^T D U (U unit upper, D diagonal), organised in 6x6 block rows inside the envelope.
//   row k:  U_kk (unit upper), Y_kj = D_k^-1 U_kk^-T A~_kj  (j > k);   Z_kj = D_k Y_kj
//   trailing: A~_ij -= Z_ki^T Y_kj;  forward: z_k = U_kk^-T b~_k, b~_j -= Y_kj^T z_k;  w = D^-1 z
//   backward: x_k = U_kk^-1 (w_k - sum_j Y_kj x_j)
__global__ __launch_bounds__(PIPE_THREADS) void ba_solve_lds_kernel(BaDev B, double *__restrict__ x_out, double *__restrict__ upanel,
                                                                    const int *__restrict__ rowmax_g, int R) {
  extern __shared__ double smem[];
  if (B.ctl) { if (B.ctl[1] != 0.0) return; B.lambda = B.ctl[0]; }      // speculative trial: lambda decided on the device, skipped after a rejection
  const int P = B.P, n = 6 * P, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double *s_b = smem;                          // [n]
  double *s_win = s_b + n;                     // [R][R][36] ring of envelope rows
  double *s_pz = s_win + (size_t)R * R * 36;   // [2][R][36] Z_kj
  double *s_py = s_pz + 2 * (size_t)R * 36;    // [2][R][36] Y_kj
  double *s_ud = s_py + 2 * (size_t)R * 36;    // [P][36] U_kk (unit upper; strict upper part used)
  double *s_rd = s_ud + (size_t)P * 36;        // [P][6]  1/D_k
  double *s_y = s_rd + (size_t)P * 6;          // [2][8]  z_k
  double *s_part = s_y + 16;                   // [R][6]
  int *rowmax = reinterpret_cast<int *>(s_part + (size_t)R * 6);         // [P] envelope
  unsigned char *lut = reinterpret_cast<unsigned char *>(rowmax + P);     // [(R-1)(R-2)/2 + ..][2] (ii, jj) of trailing block t, 1 <= ii <= jj
  __shared__ int s_fail;
  const long long t_begin = wall_clock64();
  if (tid == 0) s_fail = 0;
  for (int i = tid; i < P; i += PIPE_THREADS) rowmax[i] = rowmax_g[i];
  for (int i = tid; i < n; i += PIPE_THREADS) s_b[i] = B.bp[i] - B.bs[i];
  for (int t = tid; t < R * (R - 1) / 2; t += PIPE_THREADS) {   // t = jj (jj - 1) / 2 + (ii - 1)
    int jj = 1;
    while ((jj + 1) * jj / 2 <= t) ++jj;
    lut[2 * t] = (unsigned char)(t - jj * (jj - 1) / 2 + 1);
    lut[2 * t + 1] = (unsigned char)jj;
  }
  for (int r = 0; r < R && r < P; ++r) {
    const long rb = blk_index(r, r, P);
    for (int e = tid; e < R * 36; e += PIPE_THREADS) {
      const int c = e / 36;
      s_win[((size_t)(r % R) * R) * 36 + e] = (r + c < P) ? B.H[(rb + c) * 36 + (e - c * 36)] : 0.0;
    }
  }
  // loader wave: envelope rows R and R+1 are in flight before the pipeline starts
  double ra[PIPE_LD], rb_[PIPE_LD];
  auto load_row = [&](double (&reg)[PIPE_LD], int rn) {
    if (rn < P) {
      const long rb = blk_index(rn, rn, P);
#pragma unroll
      for (int i = 0; i < PIPE_LD; ++i) {
        const int e = lane + 64 * i;
        const int c = e / 36;
        reg[i] = (e < R * 36 && rn + c < P) ? B.H[(rb + c) * 36 + (e - c * 36)] : 0.0;
      }
    }
  };
  auto store_row = [&](const double (&reg)[PIPE_LD], int k) {
    if (k + R < P) {
      double *row = s_win + ((size_t)(k % R) * R) * 36;
#pragma unroll
      for (int i = 0; i < PIPE_LD; ++i) { const int e = lane + 64 * i; if (e < R * 36) row[e] = reg[i]; }
    }
  };
  if (wave == 5) { load_row(ra, R); load_row(rb_, R + 1); }
  __syncthreads();
  // pivot block k -> U_kk, 1/D_k; panel k -> Z, Y (buffer k & 1), z_k, w_k.  Run by wave 0.
  long long t_mid = 0;
  auto factor_panel = [&](int k) {
    double *row = s_win + ((size_t)(k % R) * R) * 36;
    if (lane == 0) {
      // right-looking LDL^T of the 6x6 pivot block, in registers
      double A[36];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) { double v = row[6 * r + c]; if (r == c) v += B.lambda; A[6 * r + c] = v; }
      int fail = 0;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
        if (!(d > 0) || !isfinite(d)) { fail = 1; d = 1.0; }      // keep going (uniform control flow); x is zeroed at the end
        const double rj = rcp_nr(d);
        s_rd[(size_t)k * 6 + j] = rj;
        double u[6];
#pragma unroll
        for (int c = j + 1; c < 6; ++c) { u[c] = A[6 * j + c] * rj; s_ud[(size_t)k * 36 + 6 * j + c] = u[c]; }
#pragma unroll
        for (int r = j + 1; r < 6; ++r)
#pragma unroll
          for (int c = r; c < 6; ++c) A[6 * r + c] = __builtin_fma(-u[r], A[6 * j + c], A[6 * r + c]);
      }
      if (fail) s_fail = 1;
    }
    wave_lds_fence();
    t_mid = wall_clock64();
    const int nj = rowmax[k] - k;
    const double *U = s_ud + (size_t)k * 36, *rd = s_rd + (size_t)k * 6;
    double *pz = s_pz + (size_t)(k & 1) * R * 36, *py = s_py + (size_t)(k & 1) * R * 36, *zk = s_y + (k & 1) * 8;
    // panel columns (+ the rhs as one more column): unit-lower forward substitution
    for (int e = lane; e <= nj * 6; e += 64) {
      double a[6];
      const bool is_rhs = e == nj * 6;
      const int jj = e / 6, c = e - jj * 6;
#pragma unroll
      for (int q = 0; q < 6; ++q) a[q] = is_rhs ? s_b[6 * k + q] : row[(size_t)(1 + jj) * 36 + 6 * q + c];
#pragma unroll
      for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int r = q + 1; r < 6; ++r) a[r] = __builtin_fma(-U[6 * q + r], a[q], a[r]);
      if (is_rhs) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { zk[q] = a[q]; s_b[6 * k + q] = a[q] * rd[q]; }      // z_k, w_k
      } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const double y = a[q] * rd[q];
          pz[(size_t)jj * 36 + 6 * q + c] = a[q];
          py[(size_t)jj * 36 + 6 * q + c] = y;
          upanel[((size_t)k * R + jj) * 36 + 6 * q + c] = y;
        }
      }
    }
    wave_lds_fence();
  };
  // A~_(k+1+ii),(k+1+jj) row r -= Z_k,ii(:,r)^T Y_k,jj
  auto update_block_row = [&](int k, const double *pz, const double *py, int ii, int jj, int r) {
    double zi[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) zi[q] = pz[(size_t)ii * 36 + 6 * q + r];
    double *Aij = s_win + (((size_t)((k + 1 + ii) % R) * R) + (jj - ii)) * 36 + 6 * r;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double s0 = 0, s1 = 0;
#pragma unroll
      for (int q = 0; q < 6; q += 2) {
        s0 = __builtin_fma(zi[q], py[(size_t)jj * 36 + 6 * q + c], s0);
        s1 = __builtin_fma(zi[q + 1], py[(size_t)jj * 36 + 6 * (q + 1) + c], s1);
      }
      Aij[c] -= s0 + s1;
    }
  };
  auto update_rhs = [&](int k, const double *py, const double *zk, int jj, int c) {     // b~_(k+1+jj)[c] -= Y_k,jj(:,c)^T z_k
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int q = 0; q < 6; q += 2) {
      s0 = __builtin_fma(py[(size_t)jj * 36 + 6 * q + c], zk[q], s0);
      s1 = __builtin_fma(py[(size_t)jj * 36 + 6 * (q + 1) + c], zk[q + 1], s1);
    }
    s_b[6 * (k + 1 + jj) + c] -= s0 + s1;
  };
  if (wave == 0) factor_panel(0);
  lds_barrier();
  const long long t_loop = wall_clock64();
  long long acc_upd = 0, acc_piv = 0, acc_pan = 0, acc_bar = 0;      // pivot-wave stage breakdown (debug)
  for (int k = 0; k < P; ++k) {
    const int nj = rowmax[k] - k;
    const long long ts0 = wall_clock64();
    const double *pz = s_pz + (size_t)(k & 1) * R * 36, *py = s_py + (size_t)(k & 1) * R * 36, *zk = s_y + (k & 1) * 8;
    if (wave == 0) {
      for (int e = lane; e < nj * 6; e += 64) update_block_row(k, pz, py, 0, e / 6, e % 6);
      if (lane < 6 && nj > 0) update_rhs(k, py, zk, 0, lane);
      wave_lds_fence();
      const long long ts1 = wall_clock64();
      if (k + 1 < P) factor_panel(k + 1);
      const long long ts2 = wall_clock64();
      acc_upd += ts1 - ts0; acc_piv += t_mid - ts1; acc_pan += ts2 - t_mid;
    } else if (wave <= 4) {
      const int ul = tid - 64;
      for (int e = ul; e < (nj - 1) * 6; e += 256) update_rhs(k, py, zk, 1 + e / 6, e % 6);
      const int nb = nj * (nj - 1) / 2;
      for (int e = ul; e < nb * 6; e += 256) {
        const int t = e / 6;
        update_block_row(k, pz, py, lut[2 * t], lut[2 * t + 1], e - t * 6);
      }
    } else {
      if (k & 1) { store_row(rb_, k); load_row(rb_, k + R + 2); }
      else { store_row(ra, k); load_row(ra, k + R + 2); }
    }
    const long long ts3 = wall_clock64();
    lds_barrier();
    acc_bar += (long long)wall_clock64() - ts3;
  }
  const int fail = s_fail;
  const long long t_fwd = wall_clock64();
  if (wave == 0) {
    if (!fail) {
      double un[6];
      int njn = rowmax[P - 1] - (P - 1);
      if (lane < njn * 6) {
#pragma unroll
        for (int c = 0; c < 6; ++c) un[c] = upanel[((size_t)(P - 1) * R + lane / 6) * 36 + 6 * (lane % 6) + c];
      }
      for (int k = P - 1; k >= 0; --k) {
        const int nj = njn;
        double u[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) u[c] = un[c];
        if (k > 0) {
          njn = rowmax[k - 1] - (k - 1);
          if (lane < njn * 6) {
#pragma unroll
            for (int c = 0; c < 6; ++c) un[c] = upanel[((size_t)(k - 1) * R + lane / 6) * 36 + 6 * (lane % 6) + c];
          }
        }
        for (int e = lane; e < nj * 6; e += 64) {
          const int jj = e / 6, r = e - jj * 6;
          if (e >= 64) {
#pragma unroll
            for (int c = 0; c < 6; ++c) u[c] = upanel[((size_t)k * R + jj) * 36 + 6 * r + c];
          }
          double s0 = 0, s1 = 0;
#pragma unroll
          for (int c = 0; c < 6; c += 2) {
            s0 = __builtin_fma(u[c], s_b[6 * (k + 1 + jj) + c], s0);
            s1 = __builtin_fma(u[c + 1], s_b[6 * (k + 1 + jj) + c + 1], s1);
          }
          s_part[jj * 6 + r] = s0 + s1;
        }
        wave_lds_fence();
        double x = 0;
        if (lane < 6) {
          double p0 = 0, p1 = 0;
          int jj = 0;
          for (; jj + 1 < nj; jj += 2) { p0 += s_part[jj * 6 + lane]; p1 += s_part[(jj + 1) * 6 + lane]; }
          if (jj < nj) p0 += s_part[jj * 6 + lane];
          x = s_b[6 * k + lane] - (p0 + p1);
        }
        // unit-upper back substitution across lanes 0..5
        const double *U = s_ud + (size_t)k * 36;
        double ucol[6];
#pragma unroll
        for (int r = 1; r < 6; ++r) ucol[r] = (lane < 6) ? U[6 * (lane < 6 ? lane : 0) + r] : 0.0;
#pragma unroll
        for (int r = 5; r >= 1; --r) {
          const double xr = readlane_f64(x, r);
          if (lane < r) x = __builtin_fma(-ucol[r], xr, x);
        }
        if (lane < 6) s_b[6 * k + lane] = x;
        wave_lds_fence();
      }
    } else {
      for (int i = lane; i < n; i += 64) s_b[i] = 0;
    }
  }
  __syncthreads();
  const long long t_back = wall_clock64();
  double sc = 0;
  for (int i = tid; i < n; i += PIPE_THREADS) { const double xv = s_b[i]; x_out[i] = xv; sc += xv * (B.lambda * xv + B.bp[i]); }
  sc = wave_sum_f64(sc);
  if ((tid & 63) == 0 && sc != 0.0) atomic_add_f64(&B.scal[2], sc);
  for (int p = tid; p < P; p += PIPE_THREADS) {
    double Tn[12];
    d_se3_exp_mul(s_b + 6 * p, B.poses + 12 * (size_t)p, Tn);
    for (int i = 0; i < 12; ++i) B.poses_trial[12 * (size_t)p + i] = Tn[i];
  }
  if (tid == 0) {
    B.scal[3] = (double)fail; B.scal[4] = chi2_cur_sum(B);
    // phase timing (100 MHz wall clock ticks -> us), read by SVS_BA_DEBUG=1
    B.scal[5] = (double)(t_loop - t_begin) * 0.01; B.scal[6] = (double)(t_fwd - t_loop) * 0.01; B.scal[7] = (double)(t_back - t_fwd) * 0.01;
    B.scal[8] = acc_upd * 0.01; B.scal[9] = acc_piv * 0.01; B.scal[10] = acc_pan * 0.01; B.scal[11] = acc_bar * 0.01;
  }
}

// ---- register-resident variant for narrow envelopes (R <= 10 block rows) ------------------------------
// Same U^T D U elimination and the same pipeline as ba_solve_lds_kernel, but the pivot wave never
// round-trips through LDS inside a stage: lane (slot, c) of wave 0 owns column c of the block column
// j with j % 10 == slot for as long as j is inside the sliding window.  It keeps that column of the
// current panel (Y_k) in registers, applies panel k to its column of block row k+1, and the 6 pivots
// of block k+1 are eliminated on ALL columns of the row at once (panel columns and the rhs ride along
// as extra columns, multipliers broadcast with v_readlane) -- factorisation, panel and forward
// substitution are one pass.  Trailing blocks are updated by waves 1-3 (two block columns per lane,
// 128-bit LDS accesses), wave 4 streams envelope rows into the LDS ring.
// The back substitution is column-oriented: lane (slot, r) carries the running value of component r of
// one row, every solved x_k is broadcast from the pivot lanes and applied with 6 FMAs, panel rows are
// prefetched FUSE_PRE steps ahead into registers -- no LDS access and no barrier on its chain.
constexpr int FUSE_THREADS = 320;
constexpr int FUSE_SLOTS = 10;
constexpr int FUSE_PRE = 8;
// Two fronts ("burn at both ends"): with P1 > 0 the kernel runs as TWO workgroups.  Front 0 eliminates the block rows
// 0 .. P_top-1 in natural order; front 1 eliminates the last P1 rows (P_top = P - P1) in REVERSED block order -- the same
// code on the block-reversed matrix (blocks fetched transposed from the packed upper storage, envelope = the reversed
// profile).  The band structure decouples the two eliminations except for the R-1 rows just above front 1's set: front 1
// leaves their Schur-complement deltas (matrix corner + rhs) in `xfer`, raises flag 0, and front 0 adds them to its LDS
// window right before it consumes the first of those rows.  Back substitution mirrors it: front 0 publishes x of those
// R-1 rows as soon as they are solved (flag 1), front 1 treats them as known rows of its own (reversed) system.
// Flags hold the launch epoch (no reset needed); waits are bounded and fall into the solve's failure path.
struct FuseFronts { int P_top, P1; double *xfer; unsigned *flags; unsigned epoch; };
__device__ __forceinline__ bool fuse_wait_flag(const unsigned *flag, unsigned epoch) {
  for (int spin = 0; spin < (1 << 22); ++spin) {
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}
__global__ __launch_bounds__(FUSE_THREADS) void ba_solve_fused_kernel(BaDev B, double *__restrict__ x_out, double *__restrict__ upanel,
                                                                      const int *__restrict__ rowmax_g, int R, FuseFronts F) {
  extern __shared__ double smem[];
  if (B.ctl) { if (B.ctl[1] != 0.0) return; B.lambda = B.ctl[0]; }      // speculative trial: lambda decided on the device, skipped after a rejection
  const int front = blockIdx.x, PN = B.P;                                // PN: poses of the whole (natural) system
  const int Pe = front ? F.P1 : F.P_top;                                 // block rows this front eliminates
  const int P = front ? F.P1 + R - 1 : F.P_top;                          // block rows in its view (front 1: + the R-1 shared rows, kept symbolic)
  const int m0 = F.P_top - (R - 1);                                      // first shared row (natural index); shared rows = m0 .. P_top-1
  const bool two = F.P1 > 0;
  rowmax_g += front * PN;
  upanel += (size_t)front * ((size_t)PN * FUSE_SLOTS * 36 + 128);
  const int n = 6 * P, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // element e = 6 r + c of local block (rl, rl + cb) of this front's matrix
  auto hval = [&](int rl, int cb, int e) -> double {
    if (!front) return B.H[(blk_index(rl, rl, PN) + cb) * 36 + e];
    if (rl >= Pe) return 0.0;                                            // shared rows start from zero: they end up holding the deltas
    const int i = PN - 1 - rl, j = i - cb;                               // natural block (j, i), j <= i; off-diagonal blocks transposed
    const int r = e / 6, c = e - 6 * r;
    return B.H[blk_index(j, i, PN) * 36 + (cb == 0 ? e : 6 * c + r)];
  };
  auto nat = [&](int rl) { return front ? PN - 1 - rl : rl; };
  double *s_b = smem;                          // [n]
  double *s_win = s_b + n;                     // [R][R][36] ring of envelope rows
  double *s_pz = s_win + (size_t)R * R * 36;   // [2][R][36] Z_kj
  double *s_py = s_pz + 2 * (size_t)R * 36;    // [2][R][36] Y_kj
  double *s_ud = s_py + 2 * (size_t)R * 36;    // [P][36] U_kk (strict upper part)
  double *s_y = s_ud + (size_t)P * 36;         // [2][8]  z_k
  double *s_trash = s_y + 16;                  // [48] sink of the Z stores of the pivot block lanes
  double *s_zero = s_trash + 48;               // [48] zeros: source of the lanes outside the envelope (keeps the LDS loads unconditional)
  int *rowmax = reinterpret_cast<int *>(s_zero + 48);                     // [P] envelope
  unsigned char *lut = reinterpret_cast<unsigned char *>(rowmax + P);     // (ii, jj) of trailing block t, 1 <= ii <= jj
  __shared__ int s_fail;
  const long long t_begin = wall_clock64();
  if (tid == 0) s_fail = 0;
  if (tid < 48) s_zero[tid] = 0.0;
  for (int i = tid; i < P; i += FUSE_THREADS) rowmax[i] = rowmax_g[i];
  for (int i = tid; i < n; i += FUSE_THREADS) {
    const int rl = i / 6, g = 6 * nat(rl) + (i - 6 * rl);
    s_b[i] = rl < Pe ? B.bp[g] - B.bs[g] : 0.0;
  }
  if (front) {      // shared rows take no part in the elimination: identity U_kk and zero panel rows for the back substitution
    for (int e = tid; e < (P - Pe) * 36; e += FUSE_THREADS) s_ud[(size_t)Pe * 36 + e] = 0.0;
    for (int e = tid; e < (P - Pe) * FUSE_SLOTS * 36; e += FUSE_THREADS) upanel[(size_t)Pe * FUSE_SLOTS * 36 + e] = 0.0;
  }
  for (int t = tid; t < R * (R - 1) / 2; t += FUSE_THREADS) {   // t = jj (jj - 1) / 2 + (ii - 1)
    int jj = 1;
    while ((jj + 1) * jj / 2 <= t) ++jj;
    lut[2 * t] = (unsigned char)(t - jj * (jj - 1) / 2 + 1);
    lut[2 * t + 1] = (unsigned char)jj;
  }
  {   // the first R envelope rows: all loads of a lane in flight together (one global round trip, not one per row)
    constexpr int NI = (FUSE_SLOTS * FUSE_SLOTS * 36 + FUSE_THREADS - 1) / FUSE_THREADS;
    const int rows0 = min(R, P), total = rows0 * R * 36;
    double v[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int e = tid + u * FUSE_THREADS, r = e / (R * 36), rem = e - r * R * 36, c = rem / 36;
      v[u] = (e < total && r + c < P) ? hval(r, c, rem - c * 36) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NI; ++u) { const int e = tid + u * FUSE_THREADS; if (e < total) s_win[e] = v[u]; }      // row r of the ring = slot r % R = r
  }
  constexpr int LDN = (FUSE_SLOTS * 36 + 63) / 64;     // loader registers per lane and buffer
  double ra[LDN], rb_[LDN];
  auto load_row = [&](double (&reg)[LDN], int rn) {
    if (rn < P) {
#pragma unroll
      for (int i = 0; i < LDN; ++i) {
        const int e = lane + 64 * i;
        const int c = e / 36;
        reg[i] = (e < R * 36 && rn + c < P) ? hval(rn, c, e - c * 36) : 0.0;
      }
    }
  };
  auto store_row = [&](const double (&reg)[LDN], int k, int ringk) {      // ringk = k % R
    if (k + R < P) {
      double *row = s_win + (ringk * R) * 36;
#pragma unroll
      for (int i = 0; i < LDN; ++i) { const int e = lane + 64 * i; if (e < R * 36) row[e] = reg[i]; }
    }
  };
  if (wave == 4) { load_row(ra, R); load_row(rb_, R + 1); }
  __syncthreads();

  // ---- pivot wave state: lane (slot, c) <-> column c of block column j, j % 10 == slot; lane 60 <-> rhs
  const int slot = lane / 6, cc = lane - slot * 6;
  const bool col_lane = lane < 6 * FUSE_SLOTS, rhs_lane = lane == 6 * FUSE_SLOTS;
  double ycol[6] = {0, 0, 0, 0, 0, 0};          // this lane's column of Y_k (rhs lane: w_k); 0 outside panel k
  int failed = 0;
  long long t_ld = 0;
  double *const g_trash = upanel + (size_t)PN * FUSE_SLOTS * 36;     // 64 spare doubles behind this front's panel rows (write sink)
  const double *const g_zero = g_trash + 64;                         // 64 zeros (never written)
  // stage for pivot row p: apply panel p-1 (nj_prev blocks), eliminate, emit panel p
  auto pivot_stage = [&](int p, int nj_prev, int njp, int jp, int ringp, bool apply_only = false) {      // jp = p % 10, ringp = p % R (kept as counters)
    const int pl = jp * 6;
    int off = slot - jp;                                            // block column j = p + off
    off += off < 0 ? FUSE_SLOTS : 0;
    const bool in_env = col_lane && off <= njp, act = in_env || rhs_lane;
    double a[6];
    {
      const double *src = rhs_lane ? (s_b + 6 * p) : in_env ? (s_win + ((ringp * R) + off) * 36 + cc) : s_zero;
      const int stride = rhs_lane ? 1 : 6;
#pragma unroll
      for (int r = 0; r < 6; ++r) a[r] = src[stride * r];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) a[r] += (in_env && off == 0 && r == cc && !apply_only) ? B.lambda : 0.0;
    if (nj_prev > 0) {
      // block row p gets panel p-1:  a[r] -= sum_q Z_(p-1),p [q][r] * Y_(p-1),j [q][c]
      const double *z0 = s_pz + (size_t)((p - 1) & 1) * R * 36;      // block 0 of panel p-1 is block column p
      double z[36];
#pragma unroll
      for (int i = 0; i < 36; ++i) z[i] = z0[i];
      __builtin_amdgcn_sched_barrier(0);                            // all 18 LDS loads in flight before the first FMA waits
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int r = 0; r < 6; ++r) a[r] = __builtin_fma(-z[6 * q + r], ycol[q], a[r]);
    }
    if (apply_only) {      // first shared row of front 1: it only receives panel p-1 (the update waves cover rows >= p+1), then rests
      if (rhs_lane) {
#pragma unroll
        for (int r = 0; r < 6; ++r) s_b[6 * p + r] = a[r];
      } else if (in_env) {
#pragma unroll
        for (int r = 0; r < 6; ++r) s_win[((ringp * R) + off) * 36 + 6 * r + cc] = a[r];
      }
      wave_lds_fence();
      return;
    }
    t_ld = wall_clock64();
    // eliminate the 6 pivots of block (p,p) on every column of the row (multipliers broadcast from the pivot lanes)
    double rd[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const double rj = rcp_nr(readlane_f64(a[q], pl + q));
      rd[q] = rj;
#pragma unroll
      for (int r = q + 1; r < 6; ++r) {
        const double u = readlane_f64(a[q], pl + r) * rj;            // U_pp[q][r]
        a[r] = __builtin_fma(-u, a[q], a[r]);
      }
    }
    // a non-positive / non-finite pivot shows in its reciprocal; elimination continues (uniform control flow), x is zeroed at the end
#pragma unroll
    for (int q = 0; q < 6; ++q) failed |= !(rd[q] > 0.0 && rd[q] < 1.7976931348623157e308);
    // emit: Z = a, Y = D^-1 a.  Panel columns -> panel buffers + global row; pivot block columns -> U_pp = D^-1 (D U);
    // rhs -> z_p, w_p.  One exec mask for all 18 stores: every lane gets its own destinations and strides.
    const bool in_panel = in_env && off >= 1;
    double *dZ, *dY, *dG = g_trash + lane % 6;
    int sZ = 6, sY = 6;
    if (rhs_lane) { dZ = s_y + (p & 1) * 8; dY = s_b + 6 * p; sZ = 1; sY = 1; }
    else if (off == 0) { dZ = s_trash + cc; dY = s_ud + (size_t)p * 36 + cc; }
    else {
      dZ = s_pz + ((size_t)(p & 1) * R + (off - 1)) * 36 + cc;
      dY = s_py + ((size_t)(p & 1) * R + (off - 1)) * 36 + cc;
      dG = upanel + ((size_t)p * FUSE_SLOTS + (off - 1)) * 36 + cc;
    }
    if (act) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const double y = a[q] * rd[q];
        ycol[q] = y;
        dZ[sZ * q] = a[q];
        dY[sY * q] = y;
        dG[6 * q] = y;
      }
    } else if (col_lane) {      // outside the envelope of row p: zero block for the unconditional loads of the back substitution
#pragma unroll
      for (int q = 0; q < 6; ++q) dG[6 * q] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) ycol[q] = (in_panel || rhs_lane) ? ycol[q] : 0.0;
    wave_lds_fence();
  };
  auto env_len = [&](int k) { return k < P ? rowmax[k] - k : 0; };
  if (wave == 0) __builtin_amdgcn_s_setprio(3);      // the pivot wave is the critical path of every stage
  if (wave == 0) pivot_stage(0, 0, env_len(0), 0, 0);
  lds_barrier();
  const long long t_loop = wall_clock64();
  long long acc_piv = 0, acc_bar = 0, acc_ld = 0;      // pivot-wave stage breakdown (SVS_BA_DEBUG)
  // One loop per role (P stages, one lds_barrier each).  Separate loops keep the compiler's wait-count
  // bookkeeping apart: in a shared loop body the pivot wave waited every stage for vmcnt(0) -- the round
  // trip of its own panel stores -- because the loader's loads were pending on the merged path.
  // front 0, entering stage inj_k: the shared rows m0 .. P_top-1 are all inside the LDS window and none has been consumed
  // yet -- wait for front 1, add its deltas (update waves), one extra barrier for every role
  const int inj_k = (two && !front) ? m0 - 1 : -1;
  bool wait_failed = false;
  auto inject = [&]() {
    if (wave >= 1 && wave <= 3) {
      if (!fuse_wait_flag(F.flags, F.epoch)) wait_failed = true;
      const int ul = tid - 64, nsh = R - 1;
      // front 1's local block (P1 + a, P1 + a + d) is natural block (j, i), i = P_top-1-a, j = i-d, transposed
      for (int e = ul; e < nsh * R * 36; e += 192) {
        const int a = e / (R * 36), rem = e - a * R * 36, d = rem / 36, q = rem - d * 36, r = q / 6, c = q - 6 * r;
        if (a + d < nsh) {
          const int i = F.P_top - 1 - a, j = i - d;
          s_win[(((j % R) * R) + d) * 36 + 6 * c + r] += F.xfer[e];
        }
      }
      for (int e = ul; e < nsh * 6; e += 192) { const int a = e / 6; s_b[6 * (F.P_top - 1 - a) + (e - 6 * a)] += F.xfer[nsh * R * 36 + e]; }
    }
    lds_barrier();
  };
  if (wave == 0) {
    int nj_a = env_len(0), nj_b = env_len(1), nj_c = env_len(2);      // envelope lengths of rows k, k+1, k+2 (read ahead of use)
    int ring1 = 1 % R, slot1 = 1 % FUSE_SLOTS;                         // (k + 1) % R and (k + 1) % 10
    for (int k = 0; k < Pe; ++k) {
      if (k == inj_k) inject();
      const int nj = nj_a, nj_nx = nj_b;
      nj_a = nj_b; nj_b = nj_c; nj_c = env_len(k + 3);
      const long long ts0 = wall_clock64();
      if (k + 1 < Pe) pivot_stage(k + 1, nj, nj_nx, slot1, ring1);
      else if (k + 1 < P) pivot_stage(k + 1, nj, nj_nx, slot1, ring1, true);
      ring1 = ring1 + 1 == R ? 0 : ring1 + 1;
      slot1 = slot1 + 1 == FUSE_SLOTS ? 0 : slot1 + 1;
      const long long ts3 = wall_clock64();
      lds_barrier();
      acc_piv += ts3 - ts0; acc_bar += (long long)wall_clock64() - ts3; acc_ld += t_ld - ts0;
    }
  } else if (wave <= 3) {
    const int ul = tid - 64;
    int nj_a = env_len(0), nj_b = env_len(1);
    int ring1 = 1 % R;
    for (int k = 0; k < Pe; ++k) {
      if (k == inj_k) inject();
      const int nj = nj_a;
      nj_a = nj_b; nj_b = env_len(k + 2);
      const double *pz = s_pz + (size_t)(k & 1) * R * 36, *py = s_py + (size_t)(k & 1) * R * 36, *zk = s_y + (k & 1) * 8;
      // trailing blocks (ii >= 1): lane = (block, column pair)
      const int nb = nj * (nj - 1) / 2;
      for (int e = ul; e < nb * 3; e += 192) {
        const int t = e / 3, c2 = 2 * (e - t * 3);
        const int ii = lut[2 * t], jj = lut[2 * t + 1];
        double y0[6], y1[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) { y0[q] = py[(size_t)jj * 36 + 6 * q + c2]; y1[q] = py[(size_t)jj * 36 + 6 * q + c2 + 1]; }
        int ring = ring1 + ii;
        ring -= ring >= R ? R : 0;
        double *Aij = s_win + ((ring * R) + (jj - ii)) * 36 + c2;
        const double *zi = pz + (size_t)ii * 36;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double s0 = 0, s1 = 0;
#pragma unroll
          for (int q = 0; q < 6; ++q) { s0 = __builtin_fma(zi[6 * q + r], y0[q], s0); s1 = __builtin_fma(zi[6 * q + r], y1[q], s1); }
          Aij[6 * r] -= s0; Aij[6 * r + 1] -= s1;
        }
      }
      // b~_(k+1+jj) -= Y_k,jj^T z_k  for jj >= 1 (block row k+1 is the pivot wave's)
      for (int e = ul; e < (nj - 1) * 6; e += 192) {
        const int jj = 1 + e / 6, c = e % 6;
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int q = 0; q < 6; q += 2) {
          s0 = __builtin_fma(py[(size_t)jj * 36 + 6 * q + c], zk[q], s0);
          s1 = __builtin_fma(py[(size_t)jj * 36 + 6 * (q + 1) + c], zk[q + 1], s1);
        }
        s_b[6 * (k + 1 + jj) + c] -= s0 + s1;
      }
      ring1 = ring1 + 1 == R ? 0 : ring1 + 1;
      lds_barrier();
    }
  } else {
    int ringk = 0;
    for (int k = 0; k < Pe; ++k) {
      if (k == inj_k) inject();
      if (k & 1) { store_row(rb_, k, ringk); load_row(rb_, k + R + 2); }
      else { store_row(ra, k, ringk); load_row(ra, k + R + 2); }
      ringk = ringk + 1 == R ? 0 : ringk + 1;
      lds_barrier();
    }
  }
  if (wave == 0 && __any(failed)) s_fail = 1;
  if (__any(wait_failed) && lane == 0) s_fail = 1;
  __syncthreads();
  if (front) {
    // hand the deltas of the shared rows to front 0, then wait for their solution
    const int nsh = R - 1;
    for (int e = tid; e < nsh * R * 36; e += FUSE_THREADS) {
      const int a = e / (R * 36), rem = e - a * R * 36;
      F.xfer[e] = s_win[(((Pe + a) % R) * R) * 36 + rem];
    }
    for (int e = tid; e < nsh * 6; e += FUSE_THREADS) F.xfer[nsh * R * 36 + e] = s_b[6 * Pe + e];
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(F.flags, F.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
      if (!fuse_wait_flag(F.flags + 1, F.epoch)) { if (lane == 0) s_fail = 1; }
      const double *xs = F.xfer + nsh * R * 36 + nsh * 6;              // x of natural rows m0 .. P_top-1
      for (int e = lane; e < nsh * 6; e += 64) { const int a = e / 6; s_b[6 * (Pe + a) + (e - 6 * a)] = xs[6 * (nsh - 1 - a) + (e - 6 * a)]; }
    }
    __syncthreads();
  }
  const int fail = s_fail;
  const long long t_fwd = wall_clock64();
  const bool publish = two && !front;
  double *const xpub = F.xfer + (R - 1) * R * 36 + (R - 1) * 6;
  if (wave == 0) {
    if (!fail) {
      // ---- back substitution: lane (slot, r) carries component r of row i, i % 10 == slot
      const int r = cc;
      // lane state: dist = (k - slot) mod 10 for the step being prefetched (row i = k - dist; dist == 0: pivot lane)
      auto load_row_state = [&](int i, double &w, double (&u)[6]) {      // w_i[r], U_ii[r][r+1..5]
        const bool ok = i >= 0 && col_lane;
        w = ok ? s_b[6 * i + r] : 0.0;
#pragma unroll
        for (int rr = 1; rr < 6; ++rr) u[rr] = (ok && rr > r) ? s_ud[(size_t)i * 36 + 6 * r + rr] : 0.0;
      };
      int kp = P - 1;                                              // step whose panel row is fetched next
      int distp = (kp % FUSE_SLOTS) - slot;
      distp += distp < 0 ? FUSE_SLOTS : 0;
      auto load_y = [&](double (&y)[6]) {                          // Y_(i,kp)[r][:], i = kp - distp;  0 on the pivot lanes
        const int i = kp - distp;
        const bool ok = col_lane && distp > 0 && i >= 0;
        // unconditional loads (lanes without a block read a zero block): predicated loads would make the compiler drain
        // vmcnt to 0 every step and put a full global round trip on the chain
        const double *src = ok ? upanel + ((size_t)i * FUSE_SLOTS + (distp - 1)) * 36 + 6 * r : g_zero;
#pragma unroll
        for (int c = 0; c < 6; ++c) y[c] = src[c];
        --kp;
        distp = distp == 0 ? FUSE_SLOTS - 1 : distp - 1;
      };
      double yq[FUSE_PRE][6];
#pragma unroll
      for (int d = 0; d < FUSE_PRE; ++d) load_y(yq[d]);
      double acc, wnext, ucol[6], ucolnext[6];
      int dist = ((P - 1) % FUSE_SLOTS) - slot;
      dist += dist < 0 ? FUSE_SLOTS : 0;
      load_row_state(P - 1 - dist, acc, ucol);
      load_row_state(P - 1 - dist - FUSE_SLOTS, wnext, ucolnext);
      int sk = (P - 1) % FUSE_SLOTS;
      for (int kb = P - 1; kb >= 0; kb -= FUSE_PRE) {
#pragma unroll
        for (int d = 0; d < FUSE_PRE; ++d) {
          const int k = kb - d;
          if (k >= 0) {
            const int pl = sk * 6;
            const bool piv = col_lane && dist == 0;
            double cm[6], xs[6];
#pragma unroll
            for (int rr = 1; rr < 6; ++rr) cm[rr] = piv ? ucol[rr] : 0.0;
#pragma unroll
            for (int rr = 5; rr >= 1; --rr) {
              xs[rr] = readlane_f64(acc, pl + rr);
              acc = __builtin_fma(-cm[rr], xs[rr], acc);
            }
            xs[0] = readlane_f64(acc, pl);
            if (piv) {
              s_b[6 * k + r] = acc;                                   // x_k
              if (publish && k >= m0) xpub[6 * (k - m0) + r] = acc;
              acc = wnext;
#pragma unroll
              for (int rr = 1; rr < 6; ++rr) ucol[rr] = ucolnext[rr];
              load_row_state(k - 2 * FUSE_SLOTS, wnext, ucolnext);
            }
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int c = 0; c < 6; c += 2) { s0 = __builtin_fma(yq[d][c], xs[c], s0); s1 = __builtin_fma(yq[d][c + 1], xs[c + 1], s1); }
            acc -= s0 + s1;
            if (publish && k == m0) {                                 // the shared rows are solved: release front 1
              __threadfence();
              if (lane == 0) __hip_atomic_store(F.flags + 1, F.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            load_y(yq[d]);
            sk = sk == 0 ? FUSE_SLOTS - 1 : sk - 1;
            dist = dist == 0 ? FUSE_SLOTS - 1 : dist - 1;
          }
        }
      }
    } else {
      for (int i = lane; i < n; i += 64) s_b[i] = 0;
      if (publish) {                                                   // a failed factorisation still releases front 1 (the step is rejected anyway)
        for (int i = lane; i < (R - 1) * 6; i += 64) xpub[i] = 0.0;
        __threadfence();
        if (lane == 0) __hip_atomic_store(F.flags + 1, F.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __syncthreads();
  const long long t_back = wall_clock64();
  double sc = 0;
  for (int i = tid; i < 6 * Pe; i += FUSE_THREADS) {
    const int rl = i / 6, g = 6 * nat(rl) + (i - 6 * rl);
    const double xv = s_b[i];
    x_out[g] = xv;
    sc += xv * (B.lambda * xv + B.bp[g]);
  }
  sc = wave_sum_f64(sc);
  if ((tid & 63) == 0 && sc != 0.0) atomic_add_f64(&B.scal[2], sc);
  for (int p = tid; p < Pe; p += FUSE_THREADS) {
    double Tn[12];
    const int g = nat(p);
    d_se3_exp_mul(s_b + 6 * p, B.poses + 12 * (size_t)g, Tn);
    for (int i = 0; i < 12; ++i) B.poses_trial[12 * (size_t)g + i] = Tn[i];
  }
  if (tid == 0 && fail) B.scal[3] = 1.0;                               // zeroed before every trial by the Schur kernel / the host memset
  if (tid == 0 && !front) {
    B.scal[4] = chi2_cur_sum(B);
    B.scal[5] = (double)(t_loop - t_begin) * 0.01; B.scal[6] = (double)(t_fwd - t_loop) * 0.01; B.scal[7] = (double)(t_back - t_fwd) * 0.01;
    B.scal[8] = acc_ld * 0.01; B.scal[9] = (acc_piv - acc_ld) * 0.01; B.scal[10] = 0; B.scal[11] = acc_bar * 0.01;
  }
}

// expands the packed upper blocks to a full symmetric matrix + bred (parity tests)
__global__ void ba_expand_kernel(BaDev B, double *__restrict__ Hfull, double *__restrict__ bred) {
  const int P = B.P, n = 6 * P;
  const long total = (long)n * n;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / n), col = (int)(e % n);
    int bi = row / 6, bj = col / 6, r = row % 6, c = col % 6;
    double v;
    if (bi < bj) v = B.H[blk_index(bi, bj, P) * 36 + 6 * r + c];
    else if (bi > bj) v = B.H[blk_index(bj, bi, P) * 36 + 6 * c + r];
    else v = r <= c ? B.H[blk_index(bi, bi, P) * 36 + 6 * r + c] : B.H[blk_index(bi, bi, P) * 36 + 6 * c + r];
    if (row == col) v += B.lambda;
    Hfull[e] = v;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bred[i] = B.bp[i] - B.bs[i];
}

// Accept / reject of one LM trial on the device (OptimizationAlgorithmLevenberg::solve), used by the
// speculative enqueue of svs_ba_optimize: on acceptance the next trial's lambda is published in ctl[0]; anything else
// (rejection, rho == 0, non-finite chi2) raises the abort flag, the already enqueued kernels of later trials return
// immediately and the host takes over with its ordinary loop.  Record per iteration: chi2 at the current state, trial
// chi2, scale, rho, accepted, lambda after the update, solver failure, done marker.
__global__ void ba_lm_kernel(BaDev B, int it) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || B.ctl[1] != 0.0) return;
  ba_lm_decide(B, it);
}

}  // namespace

// Small host-side worker pool for the per-call marshalling (svs_ba_set_problem): the phases are passes over the 64-byte
// edge records and are memory-bound on one core (~0.25 ms per pass at 100k edges).  Workers spin briefly between the
// back-to-back phases of one call and sleep on a condition variable between calls.
class HostPool {
 public:
  explicit HostPool(int n) : n_(std::max(1, n)) {
    for (int t = 1; t < n_; ++t) th_.emplace_back([this, t] { loop(t); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_.store(true); gen_fast_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  int size() const { return n_; }
  // fn(t, n) on every worker t in [0, n); returns when all are done
  void run(const std::function<void(int, int)> &fn) {
    if (n_ == 1) { fn(0, 1); return; }
    job_ = &fn;
    pending_.store(n_ - 1, std::memory_order_relaxed);
    { std::lock_guard<std::mutex> lk(m_); gen_fast_.fetch_add(1, std::memory_order_release); }      // under the lock: no lost wake-up
    cv_.notify_all();
    fn(0, n_);
    for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; ++spin)       // phases are sub-millisecond: spin briefly, then let go of the core
      if (spin > 4096) std::this_thread::yield();
  }
  std::mutex use_;                      // one caller at a time (the pool is shared by every optimizer of the process)

 private:
  void loop(int t) {
    unsigned seen = 0;
    for (;;) {
      bool got = false;
      for (int spin = 0; spin < 20000 && !got; ++spin) got = gen_fast_.load(std::memory_order_acquire) != seen;     // ~100 us of polling
      if (!got) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_.load() || gen_fast_.load(std::memory_order_acquire) != seen; });
      }
      if (stop_.load()) return;
      seen = gen_fast_.load(std::memory_order_acquire);
      (*job_)(t, n_);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  const std::function<void(int, int)> *job_ = nullptr;
  std::atomic<unsigned> gen_fast_{0};
  std::atomic<int> pending_{0};
  std::atomic<bool> stop_{false};
};

struct BaOptions {                      // experiment / test switches, latched at svs_ba_create (never read from the environment per call)
  int no_speculation = 0, one_front = 0, no_fused_solve = 0, no_lds_solve = 0, no_fused_cons = 0, no_grid_solve = 0, debug = 0;
  int nw = 0, nw4 = 0, p1 = -1, group = 0, host_threads = 0, grid_g = 0;
};
int svs_comm_allreduce_hook(void *d_buf, size_t count, void *user);      // comm.hip
// waves per workgroup of the Schur kernel: the smallest of 4..8 that gets the grid down to one workgroup per CU (if any does)
static inline int pick_nw(int n_chunks, int n_cu, const BaOptions &opt) {
  int nw = 4;
  if (!opt.nw4) for (int c = 5; c <= 8 && (n_chunks + nw - 1) / nw > n_cu; ++c) if ((n_chunks + c - 1) / c <= n_cu) nw = c;
  if (opt.nw >= 4) nw = opt.nw;      // 4..8 (clamped where it is set)
  return nw;
}

struct svs_ba {
  svs_ctx *ctx = nullptr;
  BaOptions opt;
  svs_comm *comm = nullptr;             // library-owned collective of sharded runs (svs_ba_set_comm)
  bool problem_valid = false;           // set by a COMPLETED svs_ba_set_problem / svs_ba_window_update; every other entry point requires it
  // ---- persistent window (svs_ba_window_*): every observation handed over since the last reset stays on the device ----
  svs_ba_edge *w_store = nullptr; size_t w_n = 0, w_cap = 0;           // observation store, ids in .point / .pose, arrival order
  int *w_pose_tab = nullptr, *w_point_tab = nullptr; size_t w_pose_tab_n = 0, w_point_tab_n = 0;      // id -> window index (-1: not in the window)
  void *w_work = nullptr; size_t w_work_bytes = 0;                      // per-call work arrays (grow-only)
  void *w_sort_tmp = nullptr; size_t w_sort_tmp_bytes = 0;
  unsigned char *w_hback = nullptr; size_t w_hback_bytes = 0;           // pinned read-back (counters, landmark lengths, pattern)
  int P = 0, L = 0, E = 0, C = 0, n_chunks = 0, n_wide = 0, add_pose_terms = 1;
  std::vector<int> w_chunk_nlm;                       // landmarks per wave chunk (set_problem work vector)
  int nw_sched = 0;                                   // waves per Schur workgroup the chunk list was laid out for (0: chosen at launch)
  svs_cam cam{};
  svs_ba_params prm{};
  double *d_poses[2] = {nullptr, nullptr}, *d_psi[2] = {nullptr, nullptr};
  int cur = 0;
  svs_ba_edge *d_edges = nullptr;
  int *d_chunk_start = nullptr, *d_chunk_len = nullptr;
  svs_ba_constraint *d_cons = nullptr;
  double *d_red = nullptr;     // [nblk*36][bp 6P][bs 6P][chi2]
  size_t red_count = 0;
  double *d_x = nullptr, *d_scal = nullptr, *d_linv = nullptr;
  int *d_rowmax = nullptr, *d_colmin = nullptr;
  double *d_upanel = nullptr;           // [P][R][36] panel rows of the LDS-window solve
  int env_R = 0;                        // max envelope row length + 1 (0 = unknown)
  bool use_lds_solve = false, use_fused_solve = false; size_t lds_solve_smem = 0;
  bool timing = false;                         // hipEvent brackets around the three dominant kernels of every trial (svs_ba_set_timing / svs_ba_kernel_times)
  int fuse_P1 = 0; unsigned fuse_epoch = 0;    // two-front fused solve: rows of the reversed front (0 = single front), launch counter for its flags
  int *d_rowmax2 = nullptr; size_t cap_rowmax2 = 0; double *d_xfer = nullptr; unsigned *d_flags = nullptr;
  unsigned *d_gridbar = nullptr; int grid_G = 0;      // multi-workgroup solve: arrival counter + failure flag, number of workgroups (0 = not used)
  double *d_ctl = nullptr, *h_ctl = nullptr;   // LM control block of the speculative path (device + pinned host mirror)
  int ctl_iters = 0;
  std::vector<hipEvent_t> spec_ev;      // 6 events per speculative trial
  double *h_scal = nullptr;             // pinned host mirror of d_scal (the per-trial read-back must not go through a pageable staging copy)
  double *d_pattern = nullptr;          // [P*P] structural indicator (all-reduced in sharded runs)
  std::vector<double> h_pattern;
  // persistent host work arrays / pinned staging of set_problem, device capacities (grow-only)
  std::vector<int> w_anchor, w_nobs, w_pos, w_off, w_aoff, w_alist, w_order, w_cs, w_cl;
  std::vector<int> w_cnt;                      // [workers][L] per-worker landmark counts -> start offsets
  std::vector<uint64_t> w_keys, w_ent;         // per edge: (point, pose) / per slot: (pose, source index)
  unsigned char *h_stage = nullptr; size_t h_stage_cap = 0, h_stage_used = 0;      // pinned staging of the small per-call uploads
  HostPool *pool = nullptr;                    // marshalling workers: ONE pool per process, shared by all optimizers (created on first use)
  std::vector<double> w_pat_local;
  svs_ba_edge *h_edges = nullptr; size_t h_edges_cap = 0;
  size_t cap_poses[2] = {0, 0}, cap_psi[2] = {0, 0}, cap_edges = 0, cap_cs = 0, cap_cl = 0, cap_cons = 0, cap_red = 0, cap_x = 0, cap_scal = 0,
         cap_linv = 0, cap_rowmax = 0, cap_colmin = 0, cap_pattern = 0, cap_upanel = 0;
  bool profile_ready = false;
  hipEvent_t ev[6] = {};
  float t_reduce = 0, t_solve = 0, t_backsub = 0;
  int n_reduce = 0;
  void free_all() {
    for (int i = 0; i < 2; ++i) { if (d_poses[i]) (void)hipFree(d_poses[i]); if (d_psi[i]) (void)hipFree(d_psi[i]); d_poses[i] = d_psi[i] = nullptr; }
    if (d_edges) (void)hipFree(d_edges); if (d_chunk_start) (void)hipFree(d_chunk_start); if (d_chunk_len) (void)hipFree(d_chunk_len);
    if (d_cons) (void)hipFree(d_cons); if (d_red) (void)hipFree(d_red); if (d_x) (void)hipFree(d_x); if (d_scal) (void)hipFree(d_scal);
    if (d_linv) (void)hipFree(d_linv);
    if (d_rowmax) (void)hipFree(d_rowmax); if (d_colmin) (void)hipFree(d_colmin); if (d_pattern) (void)hipFree(d_pattern);
    if (d_upanel) (void)hipFree(d_upanel); d_upanel = nullptr; env_R = 0; use_lds_solve = false;
    d_rowmax = d_colmin = nullptr; d_pattern = nullptr; profile_ready = false;
    d_edges = nullptr; d_chunk_start = d_chunk_len = nullptr; d_cons = nullptr; d_red = d_x = d_scal = d_linv = nullptr;
    cap_poses[0] = cap_poses[1] = cap_psi[0] = cap_psi[1] = cap_edges = cap_cs = cap_cl = cap_cons = cap_red = cap_x = cap_scal = cap_linv = cap_rowmax =
        cap_colmin = cap_pattern = cap_upanel = 0;
    if (h_edges) { (void)hipHostFree(h_edges); h_edges = nullptr; h_edges_cap = 0; }
  }
};

static BaDev make_dev(const svs_ba *ba, double lambda, int cur = -1, double *ctl = nullptr) {
  BaDev B{};
  if (cur < 0) cur = ba->cur;
  B.ctl = ctl;
  B.P = ba->P; B.L = ba->L; B.E = ba->E; B.C = ba->add_pose_terms ? ba->C : 0; B.n_chunks = ba->n_chunks;
  B.poses = ba->d_poses[cur]; B.psi = ba->d_psi[cur];
  B.poses_trial = ba->d_poses[1 - cur]; B.psi_trial = ba->d_psi[1 - cur];
  B.edges = ba->d_edges; B.chunk_start = ba->d_chunk_start; B.chunk_len = ba->d_chunk_len; B.cons = ba->d_cons;
  const size_t nblk = (size_t)ba->P * (ba->P + 1) / 2;
  B.H = ba->d_red; B.bp = ba->d_red + nblk * 36; B.bs = B.bp + 6 * (size_t)ba->P; B.chi2_cur = B.bs + 6 * (size_t)ba->P;
  B.x = ba->d_x; B.scal = ba->d_scal;
  B.cam = ba->cam; B.delta = ba->prm.huber_delta; B.lambda = lambda; B.robust = ba->prm.use_robust; B.self_mode = ba->prm.self_edge_mode;
  B.fuse_cons = (B.C > 0 && B.n_chunks > 0 && !ba->opt.no_fused_cons) ? 1 : 0;
  B.n_wide = ba->n_wide; B.wide_start = ba->d_chunk_start + ba->n_chunks; B.wide_len = ba->d_chunk_len + ba->n_chunks;
  B.wide_split = 16;
  return B;
}

// Small uploads go through one pinned staging area: an asynchronous copy from pageable memory makes the runtime wait for
// the stream (it has to reuse its own bounce buffer), which would serialise the host behind the big edge DMA.  The area is
// reset by svs_ba_set_problem after its initial stream synchronisation.
static int stage_reserve(svs_ba *ba, size_t bytes) {
  svs_ctx *ctx = ba->ctx;
  if (bytes <= ba->h_stage_cap) return SVS_OK;
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ba->h_stage) (void)hipHostFree(ba->h_stage);
  ba->h_stage = nullptr; ba->h_stage_cap = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  SVS_HIP(ctx, hipHostMalloc((void **)&ba->h_stage, want, hipHostMallocDefault));
  ba->h_stage_cap = want;
  return SVS_OK;
}
static int stage_upload(svs_ba *ba, void *d_dst, const void *h_src, size_t bytes) {
  svs_ctx *ctx = ba->ctx;
  if (bytes == 0) return SVS_OK;
  const size_t off = (ba->h_stage_used + 63) & ~(size_t)63;
  if (off + bytes > ba->h_stage_cap) {      // should not happen (reserved up front): fall back to the pageable copy
    SVS_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SVS_OK;
  }
  std::memcpy(ba->h_stage + off, h_src, bytes);
  ba->h_stage_used = off + bytes;
  SVS_HIP(ctx, hipMemcpyAsync(d_dst, ba->h_stage + off, bytes, hipMemcpyHostToDevice, ctx->stream));
  return SVS_OK;
}

extern "C" int svs_ba_create(svs_ctx *ctx, svs_ba **out) {
  SVS_REQUIRE(ctx, ctx && out);
  svs_ba *ba = new svs_ba();
  ba->ctx = ctx;
  SVS_DEVICE(ctx);
  {
    auto flag = [](const char *n) { return getenv(n) ? 1 : 0; };
    auto num = [](const char *n, int lo, int hi, int dflt) { const char *e = getenv(n); if (!e) return dflt; const int v = atoi(e); return v < lo ? lo : (v > hi ? hi : v); };
    BaOptions &o = ba->opt;
    o.no_speculation = flag("SVS_BA_NO_SPECULATION"); o.one_front = flag("SVS_BA_ONE_FRONT"); o.no_fused_solve = flag("SVS_BA_NO_FUSED_SOLVE");
    o.no_lds_solve = flag("SVS_BA_NO_LDS_SOLVE"); o.no_fused_cons = flag("SVS_BA_NO_FUSED_CONS"); o.nw4 = flag("SVS_BA_NW4");
    o.debug = num("SVS_BA_DEBUG", 0, 2, 0); o.nw = num("SVS_BA_NW", 4, 8, 0); o.p1 = num("SVS_BA_P1", 0, SOLVE_MAX_P, -1);
    o.group = num("SVS_BA_GROUP", 1, WIN, 0); o.host_threads = num("SVS_HOST_THREADS", 1, 64, 0);
  }
  for (auto &e : ba->ev) SVS_HIP(ctx, hipEventCreate(&e));
  *out = ba;
  return SVS_OK;
}
extern "C" int svs_ba_destroy(svs_ba *ba) {
  if (!ba) return SVS_OK;
  (void)hipStreamSynchronize(ba->ctx->stream);
  if (ba->h_scal) { (void)hipHostFree(ba->h_scal); ba->h_scal = nullptr; }
  if (ba->h_ctl) { (void)hipHostFree(ba->h_ctl); ba->h_ctl = nullptr; }
  if (ba->h_stage) { (void)hipHostFree(ba->h_stage); ba->h_stage = nullptr; ba->h_stage_cap = 0; }
  if (ba->d_ctl) { (void)hipFree(ba->d_ctl); ba->d_ctl = nullptr; }
  if (ba->d_rowmax2) (void)hipFree(ba->d_rowmax2);
  if (ba->d_xfer) (void)hipFree(ba->d_xfer);
  if (ba->d_flags) (void)hipFree(ba->d_flags);
  if (ba->d_gridbar) (void)hipFree(ba->d_gridbar);
  if (ba->w_store) (void)hipFree(ba->w_store);
  if (ba->w_pose_tab) (void)hipFree(ba->w_pose_tab);
  if (ba->w_point_tab) (void)hipFree(ba->w_point_tab);
  if (ba->w_work) (void)hipFree(ba->w_work);
  if (ba->w_sort_tmp) (void)hipFree(ba->w_sort_tmp);
  if (ba->w_hback) (void)hipHostFree(ba->w_hback);
  for (auto &e : ba->spec_ev) if (e) (void)hipEventDestroy(e);
  ba->spec_ev.clear();
  ba->free_all();
  ba->pool = nullptr;                   // shared, not owned
  for (auto &e : ba->ev) if (e) (void)hipEventDestroy(e);
  delete ba;
  return SVS_OK;
}

extern "C" int svs_ba_set_problem(svs_ba *ba, int P, const double *h_poses, int L, const double *h_psi, int E,
                                  const svs_ba_edge *h_edges, int C, const svs_ba_constraint *h_cons, const svs_cam *cam,
                                  const svs_ba_params *prm, int add_pose_terms) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && h_poses && (L == 0 || h_psi) && (E == 0 || h_edges) && (C == 0 || h_cons) && cam && prm);
  SVS_REQUIRE(ctx, P >= 1 && L >= 0 && E >= 0 && C >= 0);
  SVS_DEVICE(ctx);
  // a call that fails half-way must not leave new sizes next to old buffers behind: the handle is unusable until a call completes
  ba->problem_valid = false;
  if (P > SOLVE_MAX_P) { ctx->err = "svs_ba: P > 256 poses not supported by the single-workgroup solve yet"; return SVS_ERR_UNSUPPORTED; }
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ba->h_stage_used = 0;
  { int rc = stage_reserve(ba, sizeof(double) * (12 * (size_t)P + 3 * (size_t)L) + sizeof(svs_ba_constraint) * (size_t)C + sizeof(int) * ((size_t)E / 8 + 6 * (size_t)P) + 8192);
    if (rc) return rc; }
  const bool dbg_t = ba->opt.debug != 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t_0 = now(), t_1 = t_0, t_2 = t_0, t_3 = t_0, t_4 = t_0, t_5 = t_0;
  ba->P = P; ba->L = L; ba->E = E; ba->C = C; ba->cam = *cam; ba->prm = *prm; ba->add_pose_terms = add_pose_terms; ba->cur = 0;
  ba->profile_ready = false; ba->env_R = 0; ba->use_lds_solve = ba->use_fused_solve = false;
  // Edge order (copyDataToG2o iterates hash sets, so the reference has no meaningful edge order to
  // preserve): landmarks are grouped by anchor so that the poses a workgroup touches stay inside its
  // LDS accumulation window, but inside a group of G consecutive anchors the landmarks are dealt
  // round-robin over the anchors -- neighbouring lanes then hit different pose blocks and the LDS
  // atomics of one instruction rarely collide on an address.  Whole landmarks are packed into
  // <=64-edge wave chunks, observers ascending inside a landmark.
  // All of it is linear time (bucket passes, no comparison sort of the edge list): this marshalling is what a caller
  // pays per optimize() once the kernels take a fraction of a millisecond.
  std::vector<int> &anchor_of = ba->w_anchor, &n_obs = ba->w_nobs, &lm_pos = ba->w_pos, &lm_off = ba->w_off;
  anchor_of.assign(L, -1); n_obs.assign(L, 0);
  int span = 1;
  // worker pool: one pass over the edge records is memory-bound on one core
  if (!ba->pool) {
    static std::mutex pool_mutex;
    static HostPool *shared_pool = nullptr;
    std::lock_guard<std::mutex> lk(pool_mutex);
    if (!shared_pool) {
      const unsigned hc = std::max(1u, std::thread::hardware_concurrency());
      int nt = (int)std::min(8u, std::max(1u, hc / 2));
      if (ba->opt.host_threads > 0) nt = std::min(ba->opt.host_threads, (int)hc);      // never more workers than cores
      shared_pool = new HostPool(nt);                                                  // lives for the process
    }
    ba->pool = shared_pool;
  }
  HostPool &pool = *ba->pool;
  std::lock_guard<std::mutex> pool_lock(pool.use_);
  const bool par = pool.size() > 1 && E >= 16384 && (size_t)L * pool.size() <= ((size_t)1 << 24);
  auto for_range = [&](int total, const std::function<void(int, int, int)> &body) {      // body(t, begin, end)
    if (!par) { body(0, 0, total); return; }
    pool.run([&](int t, int n) { const long long a = (long long)total * t / n, b = (long long)total * (t + 1) / n; body(t, (int)a, (int)b); });
  };
  std::vector<uint64_t> &keys = ba->w_keys, &ent = ba->w_ent;
  keys.resize((size_t)std::max(E, 1)); ent.resize((size_t)std::max(E, 1));
  // Pass 1 over the edge records, no shared writes: every worker keeps, per landmark, the count of its own edges and the
  // anchor it saw ((anchor + 1) << 8 | count in one word), and leaves a compact (point, pose) key per edge.  The merge
  // turns the per-worker counts into per-worker start offsets inside the landmark (a counting sort without atomics whose
  // result -- input order inside a landmark -- does not depend on the number of workers).
  const int T = par ? pool.size() : 1;
  std::vector<int> &cnt = ba->w_cnt;
  cnt.assign((size_t)T * std::max(L, 1), 0);
  {
    std::atomic<int> err{0};
    std::vector<int> span_t(T, 1);
    for_range(E, [&](int t, int i0, int i1) {
      int sp = 1;
      int *c = cnt.data() + (size_t)t * L;
      for (int i = i0; i < i1; ++i) {
        const svs_ba_edge &e = h_edges[i];
        if (!(e.point >= 0 && e.point < L && e.pose >= 0 && e.pose < P && e.anchor >= 0 && e.anchor < P)) { err.store(1); return; }
        const int w = c[e.point];
        if (w != 0 && (w >> 12) != e.anchor + 1) { err.store(2); return; }        // one anchor per point (slam_graph.hpp:121-133)
        if ((w & 0xfff) >= WIDE_THREADS) { err.store(3); return; }
        c[e.point] = ((e.anchor + 1) << 12) | ((w & 0xfff) + 1);
        sp = std::max(sp, std::abs(e.pose - e.anchor) + 1);
        keys[i] = ((uint64_t)(uint32_t)e.point << 32) | (uint32_t)e.pose;           // all the later passes need of the record
      }
      span_t[t] = sp;
    });
    if (err.load() == 0)
      for_range(L, [&](int, int l0, int l1) {
        for (int l = l0; l < l1; ++l) {
          int total = 0, anc = -1;
          for (int t = 0; t < T; ++t) {
            int &w = cnt[(size_t)t * L + l];
            if (w != 0) {
              const int a = (w >> 12) - 1;
              if (anc >= 0 && a != anc) err.store(2);
              anc = a;
            }
            const int n = w & 0xfff;
            w = total;                                                              // start offset of worker t inside landmark l
            total += n;
          }
          if (total > WIDE_THREADS) err.store(3);
          anchor_of[l] = anc; n_obs[l] = total;
        }
      });
    if (err.load() == 3) { ctx->err = "svs_ba: a landmark with more than 256 observations (more than a window can hold)"; return SVS_ERR_UNSUPPORTED; }
    if (err.load() == 1) { ctx->err = "svs_ba_set_problem: edge index out of range"; return SVS_ERR_INVALID; }
    if (err.load() == 2) { ctx->err = "svs_ba_set_problem: a point is observed with two different anchors"; return SVS_ERR_INVALID; }
    for (int v : span_t) span = std::max(span, v);
  }
  t_1 = now();
  // anchors interleaved per group.  One: a workgroup's landmarks then share their anchor (or two neighbouring ones where a batch crosses
  // the group boundary), its window is as narrow as the data allows (span .. span + 1 poses) and holds the most copies; the lanes that
  // collide on one address -- the landmarks of a wave seen from the same keyframe -- are dealt to different copies by landmark ordinal.
  // (Round 1 interleaved up to 8 anchors to spread those lanes over blocks instead: that needs a 2 x wider window = a quarter of the copies.)
  int G = 1;
  if (ba->opt.group > 0) G = ba->opt.group;                                 // experiments only (clamped to 1..WIN at svs_ba_create / set_option)
  // landmark order: per anchor the landmarks in index order; per group of G anchors deal them round-robin
  std::vector<int> &by_anchor_off = ba->w_aoff, &by_anchor = ba->w_alist, &lm_order = ba->w_order;
  by_anchor_off.assign(P + 1, 0);
  for (int l = 0; l < L; ++l) if (anchor_of[l] >= 0) ++by_anchor_off[anchor_of[l] + 1];
  for (int a = 0; a < P; ++a) by_anchor_off[a + 1] += by_anchor_off[a];
  by_anchor.resize(by_anchor_off[P]);
  {
    std::vector<int> fill(by_anchor_off.begin(), by_anchor_off.end() - 1);
    for (int l = 0; l < L; ++l) if (anchor_of[l] >= 0) by_anchor[fill[anchor_of[l]]++] = l;
  }
  lm_order.clear(); lm_order.reserve(by_anchor.size());
  for (int g0 = 0; g0 < P; g0 += G) {
    const int g1 = std::min(P, g0 + G);
    int longest = 0;
    for (int a = g0; a < g1; ++a) longest = std::max(longest, by_anchor_off[a + 1] - by_anchor_off[a]);
    for (int r = 0; r < longest; ++r)
      for (int a = g0; a < g1; ++a)
        if (r < by_anchor_off[a + 1] - by_anchor_off[a]) lm_order.push_back(by_anchor[by_anchor_off[a] + r]);
  }
  // landmarks with more than 64 observations go to the end of the order: they do not fit a wave chunk and get a workgroup each
  const size_t n_lm_all = lm_order.size();
  std::stable_partition(lm_order.begin(), lm_order.end(), [&](int l) { return n_obs[l] <= 64; });
  size_t n_lm_reg = n_lm_all;
  while (n_lm_reg > 0 && n_obs[lm_order[n_lm_reg - 1]] > 64) --n_lm_reg;
  // Workgroup composition.  A workgroup of the Schur kernel = nw consecutive wave chunks (whole landmarks, <= 64 edges each).
  //  * The pair phase of a wave runs as many rounds as its LARGEST landmark needs, and the waves of a workgroup share one LDS
  //    pipeline: inside each workgroup's batch the landmarks are sorted by observation count, so most waves hold landmarks of
  //    one size (1..2 rounds) and only the wave(s) with the large ones run 3 (every wave ran 3..4 before: max of a random mix).
  //    Every workgroup keeps the same mix, so the CUs stay evenly loaded.
  //  * A workgroup accumulates into an LDS window of WIN poses from the smallest pose it touches; one that held landmarks of TWO
  //    anchor groups would reach past it (2G + span - 1 poses) and send those blocks to global atomics -- 6 such workgroups of 231
  //    set the kernel's span at 50 KF (39 vs 32 us).  So a group of anchors starts a new workgroup (empty chunks pad the last one).
  std::vector<int> &chunk_nlm = ba->w_chunk_nlm;
  chunk_nlm.clear();
  {
    long e_reg = 0;
    for (size_t k = 0; k < n_lm_reg; ++k) e_reg += n_obs[lm_order[k]];
    const int nw = pick_nw((int)((e_reg + 59) / 60), ctx->n_cu, ba->opt);
    ba->nw_sched = nw;
    const bool align_groups = 2 * G + span - 1 > WIN;
    std::vector<int> out, cand, carry;
    out.reserve(n_lm_reg);
    size_t k = 0;
    while (k < n_lm_reg || !carry.empty()) {
      const int grp = !carry.empty() ? anchor_of[carry[0]] / G : anchor_of[lm_order[k]] / G;
      cand.clear();
      int sum = 0;
      for (int l : carry) { cand.push_back(l); sum += n_obs[l]; }
      carry.clear();
      while (k < n_lm_reg && sum + n_obs[lm_order[k]] <= nw * 64 && (!align_groups || anchor_of[lm_order[k]] / G == grp)) { sum += n_obs[lm_order[k]]; cand.push_back(lm_order[k]); ++k; }
      std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return n_obs[a] > n_obs[b]; });
      int n_ch = 0, len = 0, cnt = 0;
      size_t i = 0;
      for (; i < cand.size(); ++i) {
        const int n = n_obs[cand[i]];
        if (cnt > 0 && len + n > 64) { chunk_nlm.push_back(cnt); ++n_ch; len = 0; cnt = 0; if (n_ch == nw) break; }
        len += n; ++cnt; out.push_back(cand[i]);
      }
      if (cnt > 0) { chunk_nlm.push_back(cnt); ++n_ch; }
      for (; i < cand.size(); ++i) carry.push_back(cand[i]);          // did not fit the nw chunks after sorting: first in line for the next workgroup
      const bool group_ends = carry.empty() && (k >= n_lm_reg || anchor_of[lm_order[k]] / G != grp);
      if (align_groups && group_ends && k < n_lm_reg) while (n_ch < nw) { chunk_nlm.push_back(0); ++n_ch; }
    }
    std::copy(out.begin(), out.end(), lm_order.begin());
  }
  // edge slots: landmarks in that order, each with its observers ascending (insertion sort inside the landmark's slots)
  lm_pos.assign(L, -1); lm_off.assign(lm_order.size() + 1, 0);
  for (size_t k = 0; k < lm_order.size(); ++k) { lm_pos[lm_order[k]] = (int)k; lm_off[k + 1] = lm_off[k] + n_obs[lm_order[k]]; }
  if ((size_t)E > ba->h_edges_cap) {
    if (ba->h_edges) (void)hipHostFree(ba->h_edges);
    ba->h_edges_cap = (size_t)E + (size_t)E / 4 + 64;
    SVS_HIP(ctx, hipHostMalloc((void **)&ba->h_edges, sizeof(svs_ba_edge) * ba->h_edges_cap, hipHostMallocDefault));
  }
  // device buffers persist across calls and only grow (the window changes by about one keyframe per call)
  auto ensure = [&](void **ptr, size_t *cap, size_t bytes) -> hipError_t {
    if (bytes <= *cap && *ptr) return hipSuccess;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr; *cap = 0;
    const size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(ptr, want);
    if (e == hipSuccess) *cap = want;
    return e;
  };
  t_2 = now();
  svs_ba_edge *sorted = ba->h_edges;      // pinned: the upload is a plain DMA, no pageable staging copy
  for_range(E, [&](int t, int i0, int i1) {            // pass 2, compact arrays only: every edge takes its slot (same worker ranges as pass 1)
    int *c = cnt.data() + (size_t)t * L;
    for (int i = i0; i < i1; ++i) {
      const int pt = (int)(keys[i] >> 32);
      ent[lm_off[lm_pos[pt]] + c[pt]++] = ((keys[i] & 0xffffffffull) << 32) | (uint32_t)i;      // (pose, source index)
    }
  });
  // ... then each landmark is put in observer order (<= 64 entries, insertion sort on the (pose, index) words; the result
  // does not depend on the slot order above because (point, keyframe) pairs are unique), and the structural pattern of
  // the reduced camera system is marked: poses sharing a landmark, and constraints
  const size_t n_lm = lm_order.size();
  ba->h_pattern.assign((size_t)P * P, 0.0);
  std::vector<double> &pat_local = ba->w_pat_local;
  pat_local.assign((size_t)P * P * (par ? pool.size() : 0), 0.0);
  {
    std::atomic<int> dup{0};
    for_range((int)n_lm, [&](int t, int k0, int k1) {
      double *pat = par ? pat_local.data() + (size_t)t * P * P : ba->h_pattern.data();
      for (int k = k0; k < k1; ++k) {
        const int a = lm_off[k], b = lm_off[k + 1];
        for (int i = a + 1; i < b; ++i) {
          const uint64_t e = ent[i];
          int j = i;
          while (j > a && ent[j - 1] > e) { ent[j] = ent[j - 1]; --j; }
          ent[j] = e;
        }
        for (int i = a + 1; i < b; ++i) if ((ent[i] >> 32) == (ent[i - 1] >> 32)) dup.store(1);      // one observation per (point, keyframe)
        const int anc = anchor_of[lm_order[k]], p_first = (int)(ent[a] >> 32), p_last = (int)(ent[b - 1] >> 32);
        const int lo = std::min(p_first, anc), hi = std::max(p_last, anc);
        // envelope only needs, per pose, the farthest co-visible pose: mark (p, hi) for every pose of the landmark
        for (int e = a; e < b; ++e) pat[(size_t)(ent[e] >> 32) * P + hi] = 1.0;
        pat[(size_t)anc * P + hi] = 1.0;
        pat[(size_t)lo * P + hi] = 1.0;
        // ... and, for the reversed front of the two-front solve, per pose the FIRST co-visible pose: mark (lo, p)
        for (int e = a; e < b; ++e) pat[(size_t)lo * P + (ent[e] >> 32)] = 1.0;
        pat[(size_t)lo * P + anc] = 1.0;
      }
    });
    if (dup.load()) { ctx->err = "svs_ba_set_problem: two observations of one point in one keyframe"; return SVS_ERR_INVALID; }
    if (par)
      for (int t = 0; t < pool.size(); ++t)
        for (size_t i = 0; i < (size_t)P * P; ++i) if (pat_local[(size_t)t * P * P + i] != 0.0) ba->h_pattern[i] = 1.0;
  }
  t_3 = now();
  // gather the records into slot order (sequential writes into the pinned buffer, each record read once) in a few
  // sub-ranges, each uploaded as soon as it is complete: the DMA of one overlaps the gather of the next
  SVS_HIP(ctx, ensure((void **)&ba->d_edges, &ba->cap_edges, sizeof(svs_ba_edge) * (size_t)std::max(E, 1)));
  {
    const int n_sub = par ? 4 : 1;
    for (int sb = 0; sb < n_sub; ++sb) {
      const int s0 = (int)((long long)E * sb / n_sub), s1 = (int)((long long)E * (sb + 1) / n_sub);
      for_range(s1 - s0, [&](int, int i0, int i1) {
        for (int i = s0 + i0; i < s0 + i1; ++i) sorted[i] = h_edges[(uint32_t)ent[i]];
      });
      if (s1 > s0) SVS_HIP(ctx, hipMemcpyAsync(ba->d_edges + s0, sorted + s0, sizeof(svs_ba_edge) * (size_t)(s1 - s0), hipMemcpyHostToDevice, ctx->stream));
    }
  }
  std::vector<int> &cs = ba->w_cs, &cl = ba->w_cl;
  cs.clear(); cl.clear();
  {
    size_t k = 0;
    for (int cnt : chunk_nlm) { cs.push_back(lm_off[k]); cl.push_back(lm_off[k + cnt] - lm_off[k]); k += cnt; }
  }
  ba->n_chunks = (int)cs.size();
  ba->n_wide = (int)(n_lm - n_lm_reg);
  for (size_t k = n_lm_reg; k < n_lm; ++k) { cs.push_back(lm_off[k]); cl.push_back(lm_off[k + 1] - lm_off[k]); }      // wide landmarks: entries behind the chunks
  if (add_pose_terms)
    for (int c = 0; c < C; ++c) {
      SVS_REQUIRE(ctx, h_cons[c].pose1 >= 0 && h_cons[c].pose1 < P && h_cons[c].pose2 >= 0 && h_cons[c].pose2 < P);
      ba->h_pattern[(size_t)std::min(h_cons[c].pose1, h_cons[c].pose2) * P + std::max(h_cons[c].pose1, h_cons[c].pose2)] = 1.0;
    }
  t_4 = now();
  const size_t nblk = (size_t)P * (P + 1) / 2;
  ba->red_count = nblk * 36 + 12 * (size_t)P + SC_SLOTS;
  for (int k = 0; k < 2; ++k) {
    SVS_HIP(ctx, ensure((void **)&ba->d_poses[k], &ba->cap_poses[k], sizeof(double) * 12 * (size_t)P));
    SVS_HIP(ctx, ensure((void **)&ba->d_psi[k], &ba->cap_psi[k], sizeof(double) * 3 * (size_t)std::max(L, 1)));
  }
  { int rc = stage_upload(ba, ba->d_poses[0], h_poses, sizeof(double) * 12 * (size_t)P); if (rc) return rc; }
  SVS_HIP(ctx, hipMemcpyAsync(ba->d_poses[1], ba->d_poses[0], sizeof(double) * 12 * (size_t)P, hipMemcpyDeviceToDevice, ctx->stream));
  if (L) {
    int rc = stage_upload(ba, ba->d_psi[0], h_psi, sizeof(double) * 3 * (size_t)L); if (rc) return rc;
    SVS_HIP(ctx, hipMemcpyAsync(ba->d_psi[1], ba->d_psi[0], sizeof(double) * 3 * (size_t)L, hipMemcpyDeviceToDevice, ctx->stream));
  }
  SVS_HIP(ctx, ensure((void **)&ba->d_chunk_start, &ba->cap_cs, sizeof(int) * (size_t)std::max(ba->n_chunks + ba->n_wide, 1)));
  SVS_HIP(ctx, ensure((void **)&ba->d_chunk_len, &ba->cap_cl, sizeof(int) * (size_t)std::max(ba->n_chunks + ba->n_wide, 1)));
  SVS_HIP(ctx, ensure((void **)&ba->d_cons, &ba->cap_cons, sizeof(svs_ba_constraint) * (size_t)std::max(C, 1)));
  SVS_HIP(ctx, ensure((void **)&ba->d_red, &ba->cap_red, sizeof(double) * ba->red_count));
  SVS_HIP(ctx, ensure((void **)&ba->d_x, &ba->cap_x, sizeof(double) * 6 * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_scal, &ba->cap_scal, sizeof(double) * SC_N));
  SVS_HIP(ctx, ensure((void **)&ba->d_linv, &ba->cap_linv, sizeof(double) * 36 * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_rowmax, &ba->cap_rowmax, sizeof(int) * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_colmin, &ba->cap_colmin, sizeof(int) * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_pattern, &ba->cap_pattern, sizeof(double) * (size_t)P * P));
  if (!cs.empty()) {
    int rc = stage_upload(ba, ba->d_chunk_start, cs.data(), sizeof(int) * cs.size()); if (rc) return rc;
    rc = stage_upload(ba, ba->d_chunk_len, cl.data(), sizeof(int) * cl.size()); if (rc) return rc;
  }
  if (C) { int rc = stage_upload(ba, ba->d_cons, h_cons, sizeof(svs_ba_constraint) * (size_t)C); if (rc) return rc; }
  SVS_HIP(ctx, hipMemsetAsync(ba->d_x, 0, sizeof(double) * 6 * (size_t)P, ctx->stream));
  t_5 = now();
  // No synchronisation here: everything above is ordered on the ctx stream in front of whatever the caller enqueues next;
  // the caller's (pageable) arrays have been staged by the runtime when hipMemcpyAsync returns, the pinned edge buffer and
  // the work vectors are only touched again by the next set_problem, which starts with a stream synchronisation.
  if (dbg_t) SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (dbg_t) {
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "[svs_ba] set_problem: validate+count %.0f us, landmark order %.0f us, slots+sort+pattern %.0f us, gather+upload+chunks %.0f us, enqueue copies %.0f us, wait %.0f us\n",
            us(t_0, t_1), us(t_1, t_2), us(t_2, t_3), us(t_3, t_4), us(t_4, t_5), us(t_5, now()));
  }
  ba->problem_valid = true;
  return SVS_OK;
}

// ---- persistent window ----------------------------------------------------------------------------
// The reference rebuilds a whole optimizer from its graph for every optimize() (slam_graph.cpp:324, copyDataToG2o :983-1032), and
// svs_ba_set_problem mirrors that: all edge records cross PCIe and are re-ordered on every call although the window changes by about
// one keyframe.  Here the library KEEPS every observation it has been given (device-resident, 64 B each, ids of point and keyframe as
// the reference's graph names them) and a call only brings what is new: the window's poses and point values (they are the caller's
// state), the ids that define the window, the observations made since the last call, and the few hundred pose-pose constraints.
// The edge list of the window is then assembled ON THE DEVICE:
//   id -> window-index tables (direct addressed)  ->  filter the store (observation is in the window iff its point and its keyframe
//   are)  ->  64-bit keys (wide | anchor | point | pose)  ->  radix sort (rocPRIM through hipCUB: a plain library sort)  ->  gather
//   into slot order  ->  landmark segments (scan)  ->  co-visibility pattern;
// one small read-back (landmark lengths + pattern) lets the host pack landmarks into wave chunks and build the block envelope exactly
// as svs_ba_set_problem does.  Host work per call is O(points + new observations), PCIe traffic likewise.
namespace {
struct WinCounters { int n_edges, n_lm, dup, pad; };
__global__ void win_scatter_kernel(const int *__restrict__ ids, int n, int *__restrict__ tab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tab[ids[i]] = i;
}
__global__ void win_anchor_kernel(const int *__restrict__ anchor_ids, int n, const int *__restrict__ pose_tab, int *__restrict__ anchor_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) anchor_idx[i] = pose_tab[anchor_ids[i]];
}
__global__ void win_filter_kernel(const svs_ba_edge *__restrict__ store, size_t n, const int *__restrict__ pose_tab, size_t pose_tab_n,
                                  const int *__restrict__ point_tab, size_t point_tab_n, const int *__restrict__ anchor_idx,
                                  svs_ba_edge *__restrict__ tmp, int *__restrict__ count, WinCounters *__restrict__ ctr) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  svs_ba_edge e = store[r];
  if ((size_t)e.point >= point_tab_n || (size_t)e.pose >= pose_tab_n) return;
  const int li = point_tab[e.point], pi = pose_tab[e.pose];
  if (li < 0 || pi < 0) return;
  const int ai = anchor_idx[li];
  if (ai < 0) return;
  e.point = li; e.pose = pi; e.anchor = ai;
  const int slot = atomicAdd(&ctr->n_edges, 1);
  tmp[slot] = e;
  atomicAdd(&count[li], 1);
}
__global__ void win_keys_kernel(const svs_ba_edge *__restrict__ tmp, const WinCounters *__restrict__ ctr, const int *__restrict__ count,
                                unsigned long long *__restrict__ keys, int *__restrict__ vals, int cap, int pb, int lb) {
  // key = padding | wide | anchor (pb bits) | point (lb bits) | pose (pb bits): only as many bits as this window's indices need, so the
  // radix sort runs ceil((2 pb + lb + 2) / 8) passes (4 at 50 poses / 20 000 points; the fixed 42-bit layout took 6)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  if (i >= ctr->n_edges) { keys[i] = 1ull << (2 * pb + lb + 1); vals[i] = 0; return; }      // padding sorts to the end
  const svs_ba_edge &e = tmp[i];
  const unsigned long long wide = count[e.point] > 64 ? 1ull : 0ull;
  keys[i] = (wide << (2 * pb + lb)) | ((unsigned long long)e.anchor << (pb + lb)) | ((unsigned long long)e.point << pb) | (unsigned long long)e.pose;
  vals[i] = i;
}
__global__ void win_gather_kernel(const svs_ba_edge *__restrict__ tmp, const int *__restrict__ order, const unsigned long long *__restrict__ keys,
                                  WinCounters *__restrict__ ctr, svs_ba_edge *__restrict__ edges, int *__restrict__ head, int pb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ctr->n_edges) return;
  edges[i] = tmp[order[i]];
  const bool h = i == 0 || (keys[i] >> pb) != (keys[i - 1] >> pb);
  head[i] = h ? 1 : 0;
  if (i > 0 && keys[i] == keys[i - 1]) ctr->dup = 1;                    // two observations of one point in one keyframe
}
__global__ void win_segments_kernel(const int *__restrict__ head, const int *__restrict__ rank, WinCounters *__restrict__ ctr, int *__restrict__ lm_start) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = ctr->n_edges;
  if (i >= n) return;
  if (head[i]) lm_start[rank[i]] = i;
  if (i == n - 1) { const int n_lm = rank[i] + head[i]; ctr->n_lm = n_lm; lm_start[n_lm] = n; }      // rank = heads BEFORE i
}
__global__ void win_pattern_kernel(const svs_ba_edge *__restrict__ edges, const int *__restrict__ lm_start, const WinCounters *__restrict__ ctr, int P,
                                   unsigned char *__restrict__ pat, int *__restrict__ lm_len) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ctr->n_lm) return;
  const int a = lm_start[k], b = lm_start[k + 1];
  lm_len[k] = b - a;
  const int anc = edges[a].anchor, p_first = edges[a].pose, p_last = edges[b - 1].pose;
  const int lo = min(p_first, anc), hi = max(p_last, anc);
  for (int e = a; e < b; ++e) { const int p = edges[e].pose; pat[(size_t)p * P + hi] = 1; pat[(size_t)lo * P + p] = 1; }
  pat[(size_t)anc * P + hi] = 1; pat[(size_t)lo * P + hi] = 1; pat[(size_t)lo * P + anc] = 1;
}
}  // namespace

extern "C" int svs_ba_window_reset(svs_ba *ba) {
  if (!ba) return SVS_ERR_INVALID;
  ba->w_n = 0;
  ba->problem_valid = false;
  return SVS_OK;
}

extern "C" int svs_ba_window_update(svs_ba *ba, int P, const int32_t *h_pose_ids, const double *h_poses, int L, const int32_t *h_point_ids,
                                    const double *h_psi, const int32_t *h_anchor_pose_ids, int n_new, const svs_ba_edge *h_new_obs, int C,
                                    const svs_ba_constraint *h_cons, const svs_cam *cam, const svs_ba_params *prm) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && h_pose_ids && h_poses && (L == 0 || (h_point_ids && h_psi && h_anchor_pose_ids)) && (n_new == 0 || h_new_obs) && (C == 0 || h_cons) && cam && prm);
  SVS_REQUIRE(ctx, P >= 1 && L >= 0 && n_new >= 0 && C >= 0);
  SVS_REQUIRE(ctx, !ba->comm);                        // single-GPU path (a sharded window goes through svs_ba_set_problem)
  SVS_DEVICE(ctx);
  ba->problem_valid = false;
  if (P > SOLVE_MAX_P) { ctx->err = "svs_ba: P > 256 poses not supported by the single-workgroup solve yet"; return SVS_ERR_UNSUPPORTED; }
  auto t_0 = std::chrono::steady_clock::now();
  // ---- host: ranges of the ids, constraints by index -------------------------------------------------------------------------
  int max_pose_id = -1, max_point_id = -1;
  for (int i = 0; i < P; ++i) { SVS_REQUIRE(ctx, h_pose_ids[i] >= 0); max_pose_id = std::max(max_pose_id, h_pose_ids[i]); }
  for (int i = 0; i < L; ++i) { SVS_REQUIRE(ctx, h_point_ids[i] >= 0 && h_anchor_pose_ids[i] >= 0); max_point_id = std::max(max_point_id, h_point_ids[i]); max_pose_id = std::max(max_pose_id, h_anchor_pose_ids[i]); }
  for (int i = 0; i < n_new; ++i) {
    SVS_REQUIRE(ctx, h_new_obs[i].point >= 0 && h_new_obs[i].pose >= 0);
    max_point_id = std::max(max_point_id, h_new_obs[i].point); max_pose_id = std::max(max_pose_id, h_new_obs[i].pose);
  }
  SVS_REQUIRE(ctx, L < (1 << 23));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ba->h_stage_used = 0;
  { int rc = stage_reserve(ba, sizeof(double) * (12 * (size_t)P + 3 * (size_t)L) + sizeof(int) * (4 * (size_t)P + 2 * (size_t)L) + sizeof(svs_ba_constraint) * (size_t)C +
                                  sizeof(svs_ba_edge) * (size_t)n_new + sizeof(int) * (((size_t)ba->w_n + n_new) / 8 + 8 * (size_t)P) + 16384);
    if (rc) return rc; }
  auto grow = [&](void **ptr, size_t *cap, size_t bytes, bool keep, size_t keep_bytes) -> int {
    if (bytes <= *cap && *ptr) return SVS_OK;
    void *np = nullptr;
    const size_t want = bytes + bytes / 2 + 4096;
    SVS_HIP(ctx, hipMalloc(&np, want));
    if (keep && *ptr && keep_bytes) SVS_HIP(ctx, hipMemcpy(np, *ptr, keep_bytes, hipMemcpyDeviceToDevice));
    if (*ptr) (void)hipFree(*ptr);
    *ptr = np; *cap = want;
    return SVS_OK;
  };
  // ---- the store: append the new observations --------------------------------------------------------------------------------
  {
    size_t cap_b = ba->w_cap * sizeof(svs_ba_edge);
    int rc = grow((void **)&ba->w_store, &cap_b, (ba->w_n + (size_t)n_new) * sizeof(svs_ba_edge), true, ba->w_n * sizeof(svs_ba_edge));
    if (rc) return rc;
    ba->w_cap = cap_b / sizeof(svs_ba_edge);
    if (n_new) { rc = stage_upload(ba, ba->w_store + ba->w_n, h_new_obs, sizeof(svs_ba_edge) * (size_t)n_new); if (rc) return rc; }
    ba->w_n += (size_t)n_new;
  }
  const size_t N = ba->w_n;
  // ---- id tables ---------------------------------------------------------------------------------------------------------------
  {
    size_t cb = ba->w_pose_tab_n * sizeof(int);
    int rc = grow((void **)&ba->w_pose_tab, &cb, ((size_t)max_pose_id + 1) * sizeof(int), false, 0); if (rc) return rc;
    ba->w_pose_tab_n = cb / sizeof(int);
    cb = ba->w_point_tab_n * sizeof(int);
    rc = grow((void **)&ba->w_point_tab, &cb, ((size_t)max_point_id + 2) * sizeof(int), false, 0); if (rc) return rc;
    ba->w_point_tab_n = cb / sizeof(int);
  }
  // ---- work arrays (one allocation): ids | anchor_idx | count | tmp edges | keys x2 | vals x2 | head | rank | lm_start | lm_len | pattern | counters
  const size_t capE = N + 1;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
  const size_t o_pids = take(sizeof(int) * P), o_lids = take(sizeof(int) * (size_t)std::max(L, 1)), o_aids = take(sizeof(int) * (size_t)std::max(L, 1)),
               o_aidx = take(sizeof(int) * (size_t)std::max(L, 1)), o_count = take(sizeof(int) * (size_t)std::max(L, 1)), o_tmp = take(sizeof(svs_ba_edge) * capE),
               o_k0 = take(8 * capE), o_k1 = take(8 * capE), o_v0 = take(4 * capE), o_v1 = take(4 * capE), o_head = take(4 * capE), o_rank = take(4 * capE),
               o_lms = take(4 * ((size_t)L + 2)), o_lml = take(4 * ((size_t)L + 2)), o_pat = take((size_t)P * P), o_ctr = take(sizeof(WinCounters));
  { int rc = grow(&ba->w_work, &ba->w_work_bytes, off, false, 0); if (rc) return rc; }
  char *W = static_cast<char *>(ba->w_work);
  int *d_pids = (int *)(W + o_pids), *d_lids = (int *)(W + o_lids), *d_aids = (int *)(W + o_aids), *d_aidx = (int *)(W + o_aidx), *d_count = (int *)(W + o_count);
  svs_ba_edge *d_tmp = (svs_ba_edge *)(W + o_tmp);
  unsigned long long *d_k0 = (unsigned long long *)(W + o_k0), *d_k1 = (unsigned long long *)(W + o_k1);
  int *d_v0 = (int *)(W + o_v0), *d_v1 = (int *)(W + o_v1), *d_head = (int *)(W + o_head), *d_rank = (int *)(W + o_rank), *d_lms = (int *)(W + o_lms),
      *d_lml = (int *)(W + o_lml);
  unsigned char *d_pat = (unsigned char *)(W + o_pat);
  WinCounters *d_ctr = (WinCounters *)(W + o_ctr);
  // device buffers of the optimizer proper (grow-only, as in svs_ba_set_problem)
  auto ensure = [&](void **ptr, size_t *cap, size_t bytes) -> hipError_t {
    if (bytes <= *cap && *ptr) return hipSuccess;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr; *cap = 0;
    const size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(ptr, want);
    if (e == hipSuccess) *cap = want;
    return e;
  };
  SVS_HIP(ctx, ensure((void **)&ba->d_edges, &ba->cap_edges, sizeof(svs_ba_edge) * capE));
  for (int k = 0; k < 2; ++k) {
    SVS_HIP(ctx, ensure((void **)&ba->d_poses[k], &ba->cap_poses[k], sizeof(double) * 12 * (size_t)P));
    SVS_HIP(ctx, ensure((void **)&ba->d_psi[k], &ba->cap_psi[k], sizeof(double) * 3 * (size_t)std::max(L, 1)));
  }
  // ---- enqueue ---------------------------------------------------------------------------------------------------------------
  const int TB = 256;
  int rc = stage_upload(ba, d_pids, h_pose_ids, sizeof(int) * (size_t)P); if (rc) return rc;
  if (L) {
    if ((rc = stage_upload(ba, d_lids, h_point_ids, sizeof(int) * (size_t)L))) return rc;
    if ((rc = stage_upload(ba, d_aids, h_anchor_pose_ids, sizeof(int) * (size_t)L))) return rc;
  }
  if ((rc = stage_upload(ba, ba->d_poses[0], h_poses, sizeof(double) * 12 * (size_t)P))) return rc;
  SVS_HIP(ctx, hipMemcpyAsync(ba->d_poses[1], ba->d_poses[0], sizeof(double) * 12 * (size_t)P, hipMemcpyDeviceToDevice, ctx->stream));
  if (L) {
    if ((rc = stage_upload(ba, ba->d_psi[0], h_psi, sizeof(double) * 3 * (size_t)L))) return rc;
    SVS_HIP(ctx, hipMemcpyAsync(ba->d_psi[1], ba->d_psi[0], sizeof(double) * 3 * (size_t)L, hipMemcpyDeviceToDevice, ctx->stream));
  }
  SVS_HIP(ctx, hipMemsetAsync(ba->w_pose_tab, 0xff, ba->w_pose_tab_n * sizeof(int), ctx->stream));
  SVS_HIP(ctx, hipMemsetAsync(ba->w_point_tab, 0xff, ba->w_point_tab_n * sizeof(int), ctx->stream));
  SVS_HIP(ctx, hipMemsetAsync(W + o_count, 0, (o_tmp - o_count), ctx->stream));                     // count
  SVS_HIP(ctx, hipMemsetAsync(W + o_pat, 0, (o_ctr - o_pat) + sizeof(WinCounters), ctx->stream));   // pattern + counters
  hipLaunchKernelGGL(win_scatter_kernel, dim3(div_up(P, TB)), dim3(TB), 0, ctx->stream, d_pids, P, ba->w_pose_tab);
  if (L) {
    hipLaunchKernelGGL(win_scatter_kernel, dim3(div_up(L, TB)), dim3(TB), 0, ctx->stream, d_lids, L, ba->w_point_tab);
    hipLaunchKernelGGL(win_anchor_kernel, dim3(div_up(L, TB)), dim3(TB), 0, ctx->stream, d_aids, L, ba->w_pose_tab, d_aidx);
  }
  SVS_LAUNCH_CHECK(ctx);
  auto stage = [&](const char *what) { if (ba->opt.debug >= 2) { const hipError_t e = hipStreamSynchronize(ctx->stream); fprintf(stderr, "[svs_ba] window stage %s: %s\n", what, hipGetErrorString(e)); } };
  stage("tables");
  if (N && L) {
    hipLaunchKernelGGL(win_filter_kernel, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, ctx->stream, ba->w_store, N, ba->w_pose_tab, ba->w_pose_tab_n,
                       ba->w_point_tab, ba->w_point_tab_n, d_aidx, d_tmp, d_count, d_ctr);
    stage("filter");
    auto bits_for = [](int n) { int b = 1; while ((1 << b) < n) ++b; return b; };      // indices 0 .. n-1
    const int key_pb = bits_for(P), key_lb = bits_for(L), key_bits = 2 * key_pb + key_lb + 2;
    hipLaunchKernelGGL(win_keys_kernel, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, ctx->stream, d_tmp, d_ctr, d_count, d_k0, d_v0, (int)N, key_pb, key_lb);
    SVS_LAUNCH_CHECK(ctx);
    stage("keys");
    size_t tmp_bytes = 0;
    SVS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_k0, d_k1, d_v0, d_v1, (int)N, 0, key_bits, ctx->stream));
    size_t scan_bytes = 0;
    SVS_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_head, d_rank, (int)N, ctx->stream));
    { int rc2 = grow(&ba->w_sort_tmp, &ba->w_sort_tmp_bytes, std::max(tmp_bytes, scan_bytes), false, 0); if (rc2) return rc2; }
    tmp_bytes = ba->w_sort_tmp_bytes;
    SVS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ba->w_sort_tmp, tmp_bytes, d_k0, d_k1, d_v0, d_v1, (int)N, 0, key_bits, ctx->stream));
    stage("sort");
    SVS_HIP(ctx, hipMemsetAsync(d_head, 0, 4 * capE, ctx->stream));
    hipLaunchKernelGGL(win_gather_kernel, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, ctx->stream, d_tmp, d_v1, d_k1, d_ctr, ba->d_edges, d_head, key_pb);
    SVS_LAUNCH_CHECK(ctx);
    stage("gather");
    scan_bytes = ba->w_sort_tmp_bytes;
    SVS_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(ba->w_sort_tmp, scan_bytes, d_head, d_rank, (int)N, ctx->stream));
    stage("scan");
    hipLaunchKernelGGL(win_segments_kernel, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, ctx->stream, d_head, d_rank, d_ctr, d_lms);
    stage("segments");
    if (ba->opt.debug >= 2) { WinCounters hc; (void)hipMemcpy(&hc, d_ctr, sizeof hc, hipMemcpyDeviceToHost); fprintf(stderr, "[svs_ba] window counters: n_edges %d n_lm %d dup %d (N %zu L %d P %d)\n", hc.n_edges, hc.n_lm, hc.dup, N, L, P); }
    hipLaunchKernelGGL(win_pattern_kernel, dim3(div_up(std::max(L, 1), TB)), dim3(TB), 0, ctx->stream, ba->d_edges, d_lms, d_ctr, P, d_pat, d_lml);
    SVS_LAUNCH_CHECK(ctx);
  }
  // ---- one read-back: counters, landmark lengths, pattern ----------------------------------------------------------------------------
  const size_t back_bytes = sizeof(WinCounters) + 4 * ((size_t)L + 2) + (size_t)P * P + 64;
  if (back_bytes > ba->w_hback_bytes) {
    if (ba->w_hback) (void)hipHostFree(ba->w_hback);
    ba->w_hback = nullptr; ba->w_hback_bytes = 0;
    SVS_HIP(ctx, hipHostMalloc((void **)&ba->w_hback, back_bytes + back_bytes / 2, hipHostMallocDefault));
    ba->w_hback_bytes = back_bytes + back_bytes / 2;
  }
  WinCounters *h_ctr = reinterpret_cast<WinCounters *>(ba->w_hback);
  int *h_lml = reinterpret_cast<int *>(ba->w_hback + sizeof(WinCounters));
  unsigned char *h_pat = ba->w_hback + sizeof(WinCounters) + 4 * ((size_t)L + 2);
  SVS_HIP(ctx, hipMemcpyAsync(h_ctr, d_ctr, sizeof(WinCounters), hipMemcpyDeviceToHost, ctx->stream));
  if (L) SVS_HIP(ctx, hipMemcpyAsync(h_lml, d_lml, 4 * (size_t)L, hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipMemcpyAsync(h_pat, d_pat, (size_t)P * P, hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ba->h_stage_used = 0;                                 // everything staged so far has been consumed
  auto t_1 = std::chrono::steady_clock::now();
  if (h_ctr->dup) { ctx->err = "svs_ba_window_update: two observations of one point in one keyframe"; return SVS_ERR_INVALID; }
  const int E = (N && L) ? h_ctr->n_edges : 0, n_lm = (N && L) ? h_ctr->n_lm : 0;
  // ---- host: wave chunks, wide landmarks, constraints by index, pattern -> envelope -------------------------------------------------
  std::vector<int> &cs = ba->w_cs, &cl = ba->w_cl;
  cs.clear(); cl.clear();
  int n_reg = n_lm;
  while (n_reg > 0 && h_lml[n_reg - 1] > 64) --n_reg;               // wide landmarks sort to the end (key bit 41)
  {
    int start = 0;
    for (int k = 0; k < n_reg;) {
      const int s0 = start;
      int len = 0;
      while (k < n_reg && len + h_lml[k] <= 64) { len += h_lml[k]; ++k; }
      cs.push_back(s0); cl.push_back(len);
      start += len;
    }
    ba->n_chunks = (int)cs.size();
    ba->nw_sched = 0;
    ba->n_wide = n_lm - n_reg;
    for (int k = n_reg; k < n_lm; ++k) {
      if (h_lml[k] > WIDE_THREADS) { ctx->err = "svs_ba: a landmark with more than 256 observations"; return SVS_ERR_UNSUPPORTED; }
      cs.push_back(start); cl.push_back(h_lml[k]); start += h_lml[k];
    }
  }
  ba->h_pattern.assign((size_t)P * P, 0.0);
  for (size_t i = 0; i < (size_t)P * P; ++i) if (h_pat[i]) ba->h_pattern[i] = 1.0;
  std::vector<svs_ba_constraint> cons_idx((size_t)C);
  for (int c = 0; c < C; ++c) {
    cons_idx[c] = h_cons[c];
    int i1 = -1, i2 = -1;
    for (int i = 0; i < P; ++i) { if (h_pose_ids[i] == h_cons[c].pose1) i1 = i; if (h_pose_ids[i] == h_cons[c].pose2) i2 = i; }
    SVS_REQUIRE(ctx, i1 >= 0 && i2 >= 0);                   // a constraint of this window names two window poses
    cons_idx[c].pose1 = i1; cons_idx[c].pose2 = i2;
    ba->h_pattern[(size_t)std::min(i1, i2) * P + std::max(i1, i2)] = 1.0;
  }
  ba->P = P; ba->L = L; ba->E = E; ba->C = C; ba->cam = *cam; ba->prm = *prm; ba->add_pose_terms = 1; ba->cur = 0;
  ba->profile_ready = false; ba->env_R = 0; ba->use_lds_solve = ba->use_fused_solve = false;
  const size_t nblk = (size_t)P * (P + 1) / 2;
  ba->red_count = nblk * 36 + 12 * (size_t)P + SC_SLOTS;
  SVS_HIP(ctx, ensure((void **)&ba->d_chunk_start, &ba->cap_cs, sizeof(int) * (size_t)std::max(ba->n_chunks + ba->n_wide, 1)));
  SVS_HIP(ctx, ensure((void **)&ba->d_chunk_len, &ba->cap_cl, sizeof(int) * (size_t)std::max(ba->n_chunks + ba->n_wide, 1)));
  SVS_HIP(ctx, ensure((void **)&ba->d_cons, &ba->cap_cons, sizeof(svs_ba_constraint) * (size_t)std::max(C, 1)));
  SVS_HIP(ctx, ensure((void **)&ba->d_red, &ba->cap_red, sizeof(double) * ba->red_count));
  SVS_HIP(ctx, ensure((void **)&ba->d_x, &ba->cap_x, sizeof(double) * 6 * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_scal, &ba->cap_scal, sizeof(double) * SC_N));
  SVS_HIP(ctx, ensure((void **)&ba->d_linv, &ba->cap_linv, sizeof(double) * 36 * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_rowmax, &ba->cap_rowmax, sizeof(int) * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_colmin, &ba->cap_colmin, sizeof(int) * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_pattern, &ba->cap_pattern, sizeof(double) * (size_t)P * P));
  if (!cs.empty()) {
    if ((rc = stage_upload(ba, ba->d_chunk_start, cs.data(), sizeof(int) * cs.size()))) return rc;
    if ((rc = stage_upload(ba, ba->d_chunk_len, cl.data(), sizeof(int) * cl.size()))) return rc;
  }
  if (C) { if ((rc = stage_upload(ba, ba->d_cons, cons_idx.data(), sizeof(svs_ba_constraint) * (size_t)C))) return rc; }
  SVS_HIP(ctx, hipMemsetAsync(ba->d_x, 0, sizeof(double) * 6 * (size_t)P, ctx->stream));
  if (ba->opt.debug) {
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "[svs_ba] window_update: store %zu obs (+%d), window %d edges / %d landmarks / %d chunks / %d wide; enqueue + device pipeline + read-back %.0f us, host tail %.0f us\n",
            N, n_new, E, n_lm, ba->n_chunks, ba->n_wide, us(t_0, t_1), us(t_1, std::chrono::steady_clock::now()));
  }
  ba->problem_valid = true;
  return SVS_OK;
}

extern "C" int svs_ba_reset_state(svs_ba *ba, const double *h_poses, const double *h_psi) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && h_poses && ba->d_poses[0] && ba->problem_valid);
  SVS_DEVICE(ctx);
  ba->cur = 0;
  for (int k = 0; k < 2; ++k) {
    SVS_HIP(ctx, hipMemcpyAsync(ba->d_poses[k], h_poses, sizeof(double) * 12 * (size_t)ba->P, hipMemcpyHostToDevice, ctx->stream));
    if (ba->L && h_psi) SVS_HIP(ctx, hipMemcpyAsync(ba->d_psi[k], h_psi, sizeof(double) * 3 * (size_t)ba->L, hipMemcpyHostToDevice, ctx->stream));
  }
  SVS_HIP(ctx, hipMemsetAsync(ba->d_x, 0, sizeof(double) * 6 * (size_t)ba->P, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVS_OK;
}

// Block envelope of the reduced system (with Cholesky fill).  In sharded runs the structural
// pattern is summed over ranks first (each rank only knows its own landmarks' co-visibility).
static int ensure_profile(svs_ba *ba, svs_allreduce_fn allreduce, void *user) {
  if (ba->profile_ready) return SVS_OK;
  svs_ctx *ctx = ba->ctx;
  const int P = ba->P;
  std::vector<double> pat = ba->h_pattern;
  if (allreduce) {
    SVS_HIP(ctx, hipMemcpyAsync(ba->d_pattern, pat.data(), sizeof(double) * pat.size(), hipMemcpyHostToDevice, ctx->stream));
    if (allreduce(ba->d_pattern, pat.size(), user)) { ctx->err = "allreduce callback failed"; return SVS_ERR_INVALID; }
    SVS_HIP(ctx, hipMemcpyAsync(pat.data(), ba->d_pattern, sizeof(double) * pat.size(), hipMemcpyDeviceToHost, ctx->stream));
    SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  std::vector<int> rowmax(P), colmin(P);
  for (int i = 0; i < P; ++i) {
    rowmax[i] = i;
    for (int j = P - 1; j > i; --j) if (pat[(size_t)i * P + j] != 0.0) { rowmax[i] = j; break; }
  }
  for (int k = 0; k < P; ++k)                       // fill: row k spreads its reach to rows k+1..rowmax[k]
    for (int i = k + 1; i <= rowmax[k]; ++i) rowmax[i] = std::max(rowmax[i], rowmax[k]);
  for (int k = 0; k < P; ++k) { colmin[k] = k; for (int i = 0; i < k; ++i) if (rowmax[i] >= k) { colmin[k] = i; break; } }
  { int rc = stage_upload(ba, ba->d_rowmax, rowmax.data(), sizeof(int) * P); if (rc) return rc; }
  { int rc = stage_upload(ba, ba->d_colmin, colmin.data(), sizeof(int) * P); if (rc) return rc; }
  int R = 1;
  for (int k = 0; k < P; ++k) R = std::max(R, rowmax[k] - k + 1);
  ba->env_R = R;
  // LDS budget of the window solve: rhs + R*R window + 2 x (Z, Y) panels + U_kk + 1/diag + z + scratch + envelope + block LUT
  const size_t need = sizeof(double) * ((size_t)6 * P + (size_t)R * R * 36 + 4 * (size_t)R * 36 + (size_t)P * 42 + 16 + 128 + (size_t)R * 6) +
                      sizeof(int) * (size_t)P + (size_t)R * (R - 1) + 16;
  ba->use_lds_solve = need <= 150 * 1024 && R * 36 <= 64 * PIPE_LD && !ba->opt.no_lds_solve;
  ba->lds_solve_smem = need;
  ba->use_fused_solve = ba->use_lds_solve && R <= FUSE_SLOTS && !ba->opt.no_fused_solve;
  // two-front elimination (fused kernel only): front 1 takes the last P1 block rows in reversed order.  Balance: front 0
  // needs front 1's deltas when it reaches row P_top-(R-1), i.e. after P - P1 - (R-1) stages; front 1 needs P1 stages + the hand-over.
  ba->fuse_P1 = 0;
  if (ba->use_fused_solve && !ba->opt.one_front) {
    int P1 = (P - (R - 1) - 2) / 2;
    if (ba->opt.p1 >= 0) P1 = ba->opt.p1;                                // experiments only; goes through the same validity checks
    if (P1 >= 4 && P - P1 - (R - 1) >= 2) ba->fuse_P1 = P1;
  }
  std::vector<int> rm2(2 * (size_t)P, 0);
  if (ba->fuse_P1 > 0) {
    const int P1 = ba->fuse_P1, P_top = P - P1, Pv = P1 + R - 1;
    for (int k = 0; k < P_top; ++k) rm2[k] = std::min(rowmax[k], P_top - 1);
    // reversed profile: local row k' <-> natural P-1-k'; its reach = the topmost natural row coupled to it, with the same fill closure
    std::vector<int> rr(P);
    for (int kk = 0; kk < P; ++kk) {
      const int i = P - 1 - kk;
      int top = i;
      for (int j = 0; j < i; ++j) if (pat[(size_t)j * P + i] != 0.0) { top = j; break; }
      rr[kk] = P - 1 - top;
    }
    for (int kk = 0; kk < P; ++kk) for (int i = kk + 1; i <= rr[kk]; ++i) rr[i] = std::max(rr[i], rr[kk]);
    bool ok = true;
    for (int kk = 0; kk < P; ++kk) if (rr[kk] - kk + 1 > R) ok = false;      // (a symmetric band: same width both ways)
    for (int kk = 0; kk < Pv; ++kk) rm2[P + kk] = std::min(rr[kk], Pv - 1);
    if (!ok) ba->fuse_P1 = 0;
  }
  if (ba->fuse_P1 == 0) for (int k = 0; k < P; ++k) rm2[k] = rowmax[k];
  if (ba->use_fused_solve) {
    if (!ba->d_rowmax2 || ba->cap_rowmax2 < sizeof(int) * rm2.size()) {
      if (ba->d_rowmax2) (void)hipFree(ba->d_rowmax2);
      ba->cap_rowmax2 = sizeof(int) * rm2.size() + 256;
      SVS_HIP(ctx, hipMalloc(&ba->d_rowmax2, ba->cap_rowmax2));
    }
    { int rc = stage_upload(ba, ba->d_rowmax2, rm2.data(), sizeof(int) * rm2.size()); if (rc) return rc; }
    if (!ba->d_xfer) {
      SVS_HIP(ctx, hipMalloc(&ba->d_xfer, sizeof(double) * (FUSE_SLOTS * FUSE_SLOTS * 36 + 2 * FUSE_SLOTS * 6)));
      SVS_HIP(ctx, hipMalloc(&ba->d_flags, sizeof(unsigned) * 4));
      SVS_HIP(ctx, hipMemsetAsync(ba->d_flags, 0, sizeof(unsigned) * 4, ctx->stream));
    }
  }
  if (ba->use_lds_solve) {
    const size_t up_count = 2 * (36 * (size_t)P * std::max(R, FUSE_SLOTS) + 128);      // per front: panel rows + write sink + zero block of the fused kernel
    if (sizeof(double) * up_count > ba->cap_upanel || !ba->d_upanel) {
      if (ba->d_upanel) { (void)hipFree(ba->d_upanel); ba->d_upanel = nullptr; ba->cap_upanel = 0; }
      const size_t want = sizeof(double) * (up_count + up_count / 4);
      SVS_HIP(ctx, hipMalloc(&ba->d_upanel, want));
      ba->cap_upanel = want;
    }
    // the fused kernel's zero block sits right behind the P x 10 panel rows of each front (position depends on P): clear sink + zero block
    SVS_HIP(ctx, hipMemsetAsync(ba->d_upanel + (up_count / 2 - 128), 0, sizeof(double) * 128, ctx->stream));
    SVS_HIP(ctx, hipMemsetAsync(ba->d_upanel + (up_count - 128), 0, sizeof(double) * 128, ctx->stream));
    if (need > 64 * 1024) {
      SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    }
  }
  // wide envelopes (neither LDS variant fits): the trailing updates are spread over several workgroups once they are worth it
  ba->grid_G = 0;
  // (measured at P = 230: one workgroup 0.98 ms at R = 19, 29.9 ms at R = 224; several 1.59 / 3.67 ms -- the per-step arrival wait costs ~6 us)
  if (!ba->use_lds_solve && !ba->opt.no_grid_solve && R >= 40) {
    const long tile_rows = (long)R * (R + 1) / 2 * 6;
    ba->grid_G = (int)std::max(8l, std::min((long)ctx->n_cu, (tile_rows + 4 * SOLVE_THREADS - 1) / (4 * SOLVE_THREADS)));
    if (ba->opt.grid_g > 0) ba->grid_G = std::min(ba->opt.grid_g, ctx->n_cu);      // experiments only
    if (!ba->d_gridbar) {
      SVS_HIP(ctx, hipMalloc(&ba->d_gridbar, sizeof(unsigned) * (32 * GRID_NBAR + 4)));
    }
    const size_t smem_g = sizeof(double) * ((size_t)P * 36 + 36 + 6 * (size_t)P);
    if (smem_g > 64 * 1024) SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g));
  }
  ba->profile_ready = true;
  return SVS_OK;
}

// buildSystem + Schur reduction at the current state (MODE 0 kernels)
static int launch_reduce(svs_ba *ba, double lambda, int cur = -1, double *ctl = nullptr, hipEvent_t *ev = nullptr) {
  svs_ctx *ctx = ba->ctx;
  BaDev B = make_dev(ba, lambda, cur, ctl);
  if (!ev) ev = ba->ev;
  SVS_HIP(ctx, hipMemsetAsync(ba->d_red, 0, sizeof(double) * ba->red_count, ctx->stream));
  if (B.C > 0 && !B.fuse_cons) { hipLaunchKernelGGL(ba_constraint_kernel<0>, dim3(B.C), dim3(64), 0, ctx->stream, B); SVS_LAUNCH_CHECK(ctx); }
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[0], ctx->stream));     // brackets the landmark (Schur) kernel alone
  const bool timeline = ba->opt.debug >= 2 && B.n_chunks > 0;
  if (timeline) SVS_HIP(ctx, hipMalloc(&B.dbg, sizeof(long long) * DBG_W * (size_t)B.n_chunks));
  int dbg_nw = 1;
  if (B.n_chunks > 0) {
    const int nw = (ba->nw_sched > 0 && ba->opt.nw < 4 && !ba->opt.nw4) ? ba->nw_sched : pick_nw(B.n_chunks, ctx->n_cu, ba->opt);
    const int xc = B.fuse_cons ? B.C : 0;      // pose-pose constraints in extra workgroups of the same launch
    const int n_wg = div_up(B.n_chunks, nw);
    dbg_nw = nw;
    switch (nw) {
      case 5: hipLaunchKernelGGL((ba_landmark_kernel<0, 5, 2>), dim3(n_wg + xc), dim3(320), 0, ctx->stream, B); break;
      case 6: hipLaunchKernelGGL((ba_landmark_kernel<0, 6, 2>), dim3(n_wg + xc), dim3(384), 0, ctx->stream, B); break;
      case 7: hipLaunchKernelGGL((ba_landmark_kernel<0, 7, 2>), dim3(n_wg + xc), dim3(448), 0, ctx->stream, B); break;
      case 8: hipLaunchKernelGGL((ba_landmark_kernel<0, 8, 2>), dim3(n_wg + xc), dim3(512), 0, ctx->stream, B); break;
      default:
        if (n_wg <= ctx->n_cu) hipLaunchKernelGGL((ba_landmark_kernel<0, 4, 2>), dim3(n_wg + xc), dim3(256), 0, ctx->stream, B);      // one workgroup per CU: LDS to spare
        else hipLaunchKernelGGL((ba_landmark_kernel<0, 4, 1>), dim3(n_wg + xc), dim3(256), 0, ctx->stream, B);                         // two per CU must fit
        break;
    }
    SVS_LAUNCH_CHECK(ctx);
  }
  if (B.n_wide > 0) { hipLaunchKernelGGL(ba_wide_landmark_kernel<0>, dim3(B.n_wide * B.wide_split), dim3(WIDE_THREADS), 0, ctx->stream, B); SVS_LAUNCH_CHECK(ctx); }
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[1], ctx->stream));
  if (timeline) {   // per-wave timeline of the Schur kernel (debug only; synchronises)
    std::vector<long long> h(DBG_W * (size_t)B.n_chunks);
    SVS_HIP(ctx, hipMemcpyAsync(h.data(), B.dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(B.dbg);
    long long t0 = h[0], t1 = h[DBG_N - 1];
    for (int c = 0; c < B.n_chunks; ++c) { t0 = std::min(t0, h[(size_t)DBG_W * c]); t1 = std::max(t1, h[(size_t)DBG_W * c + DBG_N - 1]); }
    double ph[DBG_N] = {}, phmax[DBG_N] = {}, s_start = 0, mx_start = 0, mx_dur = 0;
    int slow = 0;
    for (int c = 0; c < B.n_chunks; ++c) {
      const long long *d = &h[(size_t)DBG_W * c];
      const double st = (d[0] - t0) * 0.01, dur = (d[DBG_N - 1] - d[0]) * 0.01;
      s_start += st; mx_start = std::max(mx_start, st);
      if (dur > mx_dur) { mx_dur = dur; slow = c; }
      for (int k = 1; k < DBG_N; ++k) { ph[k] += (d[k] - d[k - 1]) * 0.01; phmax[k] = std::max(phmax[k], (d[k] - d[k - 1]) * 0.01); }
    }
    const double n = B.n_chunks;
    static const char *names[DBG_N] = {"", "load", "linearize + landmark sums", "segment reduce", "D^-1, W", "anchor block", "observer blocks", "pairs", "barrier", "flush"};
    fprintf(stderr, "[svs_ba] schur kernel timeline: span %.1f us, %d waves, start avg %.1f max %.1f us, wave duration max %.1f us; phases (us avg / max / slowest wave):",
            (t1 - t0) * 0.01, B.n_chunks, s_start / n, mx_start, mx_dur);
    for (int k = 1; k < DBG_N; ++k) fprintf(stderr, " %s %.2f / %.2f / %.2f |", names[k], ph[k] / n, phmax[k], (h[(size_t)DBG_W * slow + k] - h[(size_t)DBG_W * slow + k - 1]) * 0.01);
    {   // structure of the waves: landmarks per wave, largest same-address multiplicity of one observer-block add / anchor-block add, pair rounds
      static const char *xn[DBG_X] = {"landmarks", "observer multiplicity", "anchor multiplicity", "pair rounds", "", "", "", ""};
      {
        double a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
        for (int c = 0; c < B.n_chunks; ++c) { const long long *d = &h[(size_t)DBG_W * c]; a1 += (d[DBG_N + 4] - d[0]) * 0.01; a2 += (d[DBG_N + 5] - d[DBG_N + 4]) * 0.01; a3 += (d[DBG_N + 6] - d[DBG_N + 5]) * 0.01; a4 += (d[DBG_N + 7] - d[DBG_N + 6]) * 0.01; a5 += (d[1] - d[DBG_N + 7]) * 0.01; }
        fprintf(stderr, "\n[svs_ba]   load phase (avg us, debug waits serialise it): edge record %.2f, state gathers %.2f, window extent + barrier %.2f, zeroing + barrier %.2f, linearize %.2f", a1 / n, a2 / n, a3 / n, a4 / n, a5 / n);
      }
      fprintf(stderr, "\n[svs_ba]   wave structure (avg / max / slowest wave):");
      for (int k = 0; k < 4; ++k) {
        double sum = 0; long long mx = 0;
        for (int c = 0; c < B.n_chunks; ++c) { const long long v = h[(size_t)DBG_W * c + DBG_N + k]; sum += (double)v; mx = std::max(mx, v); }
        fprintf(stderr, " %s %.1f / %lld / %lld |", xn[k], sum / n, mx, h[(size_t)DBG_W * slow + DBG_N + k]);
      }
      // workgroup view: duration of the slowest wave of each workgroup
      const int nwg = (B.n_chunks + dbg_nw - 1) / dbg_nw;
      std::vector<double> wg((size_t)nwg, 0.0);
      for (int c = 0; c < B.n_chunks; ++c) wg[c / dbg_nw] = std::max(wg[c / dbg_nw], (h[(size_t)DBG_W * c + DBG_N - 2] - h[(size_t)DBG_W * c]) * 0.01);
      std::vector<double> srt(wg); std::sort(srt.begin(), srt.end());
      fprintf(stderr, "\n[svs_ba]   workgroups (%d of %d waves): time to the barrier of the slowest wave: min %.1f median %.1f p90 %.1f max %.1f us", nwg, dbg_nw,
              srt.front(), srt[srt.size() / 2], srt[srt.size() * 9 / 10], srt.back());
    }
    fprintf(stderr, "\n");
  }
  return SVS_OK;
}

extern "C" int svs_ba_reduced_system(svs_ba *ba, double lambda, double *h_Hred, double *h_bred, double *h_chi2) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && ba->d_red && ba->problem_valid);
  SVS_DEVICE(ctx);
  int rc = launch_reduce(ba, lambda);
  if (rc) return rc;
  BaDev B = make_dev(ba, lambda);
  const int n = 6 * ba->P;
  double *d_full = nullptr, *d_b = nullptr;
  SVS_HIP(ctx, hipMalloc(&d_full, sizeof(double) * (size_t)n * n));
  SVS_HIP(ctx, hipMalloc(&d_b, sizeof(double) * n));
  hipLaunchKernelGGL(ba_expand_kernel, dim3(256), dim3(256), 0, ctx->stream, B, d_full, d_b);
  SVS_LAUNCH_CHECK(ctx);
  if (h_Hred) SVS_HIP(ctx, hipMemcpyAsync(h_Hred, d_full, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
  if (h_bred) SVS_HIP(ctx, hipMemcpyAsync(h_bred, d_b, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  double chi_slots[SC_SLOTS] = {};
  if (h_chi2) SVS_HIP(ctx, hipMemcpyAsync(chi_slots, B.chi2_cur, sizeof(chi_slots), hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h_chi2) { double t = 0; for (int i = 0; i < SC_SLOTS; ++i) t += chi_slots[i]; *h_chi2 = t; }
  (void)hipFree(d_full); (void)hipFree(d_b);
  return SVS_OK;
}

// enqueue one LM trial on the ctx stream: system at the current state, solve, trial state, trial chi2 + scale
static int enqueue_trial(svs_ba *ba, double lambda, int cur, double *ctl, hipEvent_t *ev, size_t smem_fallback, svs_allreduce_fn allreduce, void *user) {
  svs_ctx *ctx = ba->ctx;
  // (re)build at the current state with this lambda; the state only changes on accept, so a
  // rebuilt system equals g2o's "restore diagonal + add new lambda"
  int rc = launch_reduce(ba, lambda, cur, ctl, ev);
  if (rc) return rc;
  if (allreduce) { rc = allreduce(ba->d_red, ba->red_count, user); if (rc) { ctx->err = "allreduce callback failed"; return SVS_ERR_INVALID; } }
  BaDev B = make_dev(ba, lambda, cur, ctl);
  if (B.n_chunks == 0) SVS_HIP(ctx, hipMemsetAsync(ba->d_scal, 0, sizeof(double) * SC_N, ctx->stream));      // else zeroed by the Schur kernel
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[2], ctx->stream));
  if (ba->use_fused_solve)
  {
    FuseFronts F{ba->P - ba->fuse_P1, ba->fuse_P1, ba->d_xfer, ba->d_flags, ++ba->fuse_epoch};
    hipLaunchKernelGGL(ba_solve_fused_kernel, dim3(ba->fuse_P1 > 0 ? 2 : 1), dim3(FUSE_THREADS), ba->lds_solve_smem, ctx->stream, B, ba->d_x, ba->d_upanel,
                       ba->d_rowmax2, ba->env_R, F);
  }
  else if (ba->use_lds_solve)
    hipLaunchKernelGGL(ba_solve_lds_kernel, dim3(1), dim3(PIPE_THREADS), ba->lds_solve_smem, ctx->stream, B, ba->d_x, ba->d_upanel, ba->d_rowmax, ba->env_R);
  else if (ba->grid_G > 0) {
    const size_t smem_g = sizeof(double) * ((size_t)ba->P * 36 + 36 + 6 * (size_t)ba->P);
    SVS_HIP(ctx, hipMemsetAsync(ba->d_gridbar, 0, sizeof(unsigned) * (32 * GRID_NBAR + 4), ctx->stream));      // arrival counter + failure flag of this launch (a speculative
    hipLaunchKernelGGL(ba_solve_grid_kernel, dim3(ba->grid_G), dim3(SOLVE_THREADS), smem_g, ctx->stream, B, ba->d_x, ba->d_linv, ba->d_rowmax, ba->d_gridbar, 0u);      // launch may be skipped, so the counter starts at 0 every time)
  }
  else
    hipLaunchKernelGGL(ba_solve_kernel, dim3(1), dim3(SOLVE_THREADS), smem_fallback, ctx->stream, B, ba->d_x, ba->d_linv, ba->d_rowmax, ba->d_colmin);
  SVS_LAUNCH_CHECK(ctx);
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[3], ctx->stream));
  if (B.C > 0 && !B.fuse_cons) { hipLaunchKernelGGL(ba_constraint_kernel<1>, dim3(B.C), dim3(64), 0, ctx->stream, B); SVS_LAUNCH_CHECK(ctx); }
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[5], ctx->stream));
  if (B.n_chunks > 0) {
    // the same waves per workgroup as the Schur pass: one workgroup per CU where that is possible (403 four-wave workgroups left some CUs
    // with two and others with one: 19.6 us; 231 seven-wave ones: 17.0 us at 50 KF / 20k)
    const int nw = (ba->nw_sched > 0 && ba->opt.nw < 4 && !ba->opt.nw4) ? ba->nw_sched : pick_nw(B.n_chunks, ctx->n_cu, ba->opt);
    const int xc = B.fuse_cons ? B.C : 0;
    switch (nw) {
      case 5: hipLaunchKernelGGL((ba_landmark_kernel<1, 5>), dim3(div_up(B.n_chunks, 5) + xc), dim3(320), 0, ctx->stream, B); break;
      case 6: hipLaunchKernelGGL((ba_landmark_kernel<1, 6>), dim3(div_up(B.n_chunks, 6) + xc), dim3(384), 0, ctx->stream, B); break;
      case 7: hipLaunchKernelGGL((ba_landmark_kernel<1, 7>), dim3(div_up(B.n_chunks, 7) + xc), dim3(448), 0, ctx->stream, B); break;
      case 8: hipLaunchKernelGGL((ba_landmark_kernel<1, 8>), dim3(div_up(B.n_chunks, 8) + xc), dim3(512), 0, ctx->stream, B); break;
      default: hipLaunchKernelGGL((ba_landmark_kernel<1, 4>), dim3(div_up(B.n_chunks, 4) + xc), dim3(256), 0, ctx->stream, B); break;
    }
    SVS_LAUNCH_CHECK(ctx);
  }
  if (B.n_wide > 0) { hipLaunchKernelGGL(ba_wide_landmark_kernel<1>, dim3(B.n_wide), dim3(WIDE_THREADS), 0, ctx->stream, B); SVS_LAUNCH_CHECK(ctx); }
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[4], ctx->stream));
  if (allreduce) { rc = allreduce(ba->d_scal + SC_CHI, 2 * SC_SLOTS, user); if (rc) { ctx->err = "allreduce callback failed"; return SVS_ERR_INVALID; } }
  return SVS_OK;
}
static int add_trial_times(svs_ba *ba, hipEvent_t *ev) {
  svs_ctx *ctx = ba->ctx;
  float ms;
  if (!ba->timing) return SVS_OK;
  SVS_HIP(ctx, hipEventElapsedTime(&ms, ev[0], ev[1])); ba->t_reduce += ms; ba->n_reduce++;
  SVS_HIP(ctx, hipEventElapsedTime(&ms, ev[2], ev[3])); ba->t_solve += ms;
  SVS_HIP(ctx, hipEventElapsedTime(&ms, ev[5], ev[4])); ba->t_backsub += ms;
  return SVS_OK;
}

// OptimizationAlgorithmLevenberg::solve x num_iters, device math.  Control flow: the common path -- the
// first trial of every iteration is accepted -- is enqueued speculatively for all iterations at once with the
// accept/reject decision and the lambda update on the device (ba_lm_kernel), one host synchronisation in total; the first
// rejection (or terminate condition) turns the remaining enqueued kernels into no-ops and the host loop below resumes
// from exactly that point.
// An optimize() in two halves, so that several windows can be in flight at once (svs_ba_optimize_batch): optimize_begin enqueues the
// speculative trials of one window on ITS context's stream and returns without waiting; optimize_finish waits, replays the device's
// decisions and runs the host-driven remainder (rejected trials) if there is one.
struct OptRun {
  svs_allreduce_fn allreduce = nullptr; void *user = nullptr;
  double lambda = 0, ni = 2;
  svs_ba_stats st{};
  size_t smem = 0;
  bool speculate = false;
};
static int optimize_begin(svs_ba *ba, svs_allreduce_fn allreduce, void *user, OptRun &R) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && ba->d_red && ba->problem_valid);
  SVS_DEVICE(ctx);
  if (!allreduce && ba->comm) { allreduce = svs_comm_allreduce_hook; user = ba->comm; }      // library-owned collective (svs_ba_set_comm)
  R.allreduce = allreduce; R.user = user;
  const svs_ba_params &prm = ba->prm;
  double lambda = prm.lambda_init;
  R.lambda = lambda; R.ni = 2; R.st = svs_ba_stats{};
  ba->t_reduce = ba->t_solve = ba->t_backsub = 0; ba->n_reduce = 0;
  { int rc = ensure_profile(ba, allreduce, user); if (rc) return rc; }
  const size_t smem = sizeof(double) * ((size_t)6 * ba->P + (size_t)ba->P * 36 + 36);
  R.smem = smem;
  if (smem > 64 * 1024) SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const bool speculate = prm.num_iters >= 1 && prm.max_trials > 1 && !ba->opt.debug && !ba->opt.no_speculation;
  R.speculate = speculate;
  if (speculate) {
    const int n_it = prm.num_iters;
    const size_t n_ctl = 8 + 8 * (size_t)n_it;
    if (ba->ctl_iters < n_it) {
      if (ba->d_ctl) (void)hipFree(ba->d_ctl);
      if (ba->h_ctl) (void)hipHostFree(ba->h_ctl);
      ba->d_ctl = ba->h_ctl = nullptr;
      SVS_HIP(ctx, hipMalloc(&ba->d_ctl, sizeof(double) * n_ctl));
      SVS_HIP(ctx, hipHostMalloc((void **)&ba->h_ctl, sizeof(double) * n_ctl, hipHostMallocDefault));
      ba->ctl_iters = n_it;
    }
    while (ba->spec_ev.size() < 6 * (size_t)n_it) { hipEvent_t e; SVS_HIP(ctx, hipEventCreate(&e)); ba->spec_ev.push_back(e); }
    for (size_t i = 0; i < n_ctl; ++i) ba->h_ctl[i] = 0.0;
    ba->h_ctl[0] = lambda;
    SVS_HIP(ctx, hipMemcpyAsync(ba->d_ctl, ba->h_ctl, sizeof(double) * n_ctl, hipMemcpyHostToDevice, ctx->stream));
    int cur = ba->cur;
    for (int j = 0; j < n_it; ++j) {
      int rc = enqueue_trial(ba, lambda, cur, ba->d_ctl, &ba->spec_ev[6 * j], smem, allreduce, user);
      if (rc) return rc;
      BaDev B = make_dev(ba, lambda, cur, ba->d_ctl);
      hipLaunchKernelGGL(ba_lm_kernel, dim3(1), dim3(64), 0, ctx->stream, B, j);      // (taking this decision in the last workgroup of the
      SVS_LAUNCH_CHECK(ctx);                                                           //  previous kernel costs more than the launch: 400 same-address tickets)
      cur = 1 - cur;                   // as if accepted
    }
    SVS_HIP(ctx, hipMemcpyAsync(ba->h_ctl, ba->d_ctl, sizeof(double) * n_ctl, hipMemcpyDeviceToHost, ctx->stream));
  }
  return SVS_OK;
}
static int optimize_finish(svs_ba *ba, OptRun &R, svs_ba_stats *stats) {
  svs_ctx *ctx = ba->ctx;
  SVS_DEVICE(ctx);
  const svs_ba_params &prm = ba->prm;
  svs_allreduce_fn allreduce = R.allreduce; void *user = R.user;
  double lambda = R.lambda, ni = R.ni;
  svs_ba_stats st = R.st;
  const size_t smem = R.smem;
  bool ok = true;
  int it = 0;
  bool resume = false;                 // the first trial of iteration `it` was already run (and rejected) by the speculative phase
  double r_rho = 0, r_chi = 0;
  if (R.speculate) {
    const int n_it = prm.num_iters;
    SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < n_it; ++j) {
      const double *rec = ba->h_ctl + 8 + 8 * j;
      if (rec[7] == 0.0) break;        // not executed: an earlier trial raised the abort flag
      { int rc = add_trial_times(ba, &ba->spec_ev[6 * j]); if (rc) return rc; }
      ++st.trials;
      if (j == 0) st.chi2_init = rec[0];
      if (rec[4] != 0.0) {             // accepted on the device: this iteration is complete
        lambda = rec[5]; ni = 2; ++st.accepted;
        ba->cur = 1 - ba->cur;
        ++st.iterations;
        st.chi2_final = rec[1];
        it = j + 1;
      } else {                         // rejected (or rho == 0 / not finite): same bookkeeping as the host path, then resume there
        lambda *= ni; ni *= 2;
        resume = true; r_rho = rec[3]; r_chi = rec[0];
        it = j;
        break;
      }
    }
  }
  for (; it < prm.num_iters && ok; ++it) {
    if (it == 0 && !resume) { lambda = prm.lambda_init; ni = 2; }
    double rho = 0, currentChi = 0;
    int qmax = 0;
    bool skip = resume;
    if (resume) { rho = r_rho; currentChi = r_chi; qmax = 1; resume = false; }
    do {
      if (skip) { skip = false; continue; }      // `continue` in a do-while jumps to the condition: the speculative phase ran this trial
      int rc = enqueue_trial(ba, lambda, -1, nullptr, ba->ev, smem, allreduce, user);
      if (rc) return rc;
      if (!ba->h_scal) SVS_HIP(ctx, hipHostMalloc((void **)&ba->h_scal, sizeof(double) * SC_N, hipHostMallocDefault));
      double *h = ba->h_scal;
      SVS_HIP(ctx, hipMemcpyAsync(h, ba->d_scal, sizeof(double) * SC_N, hipMemcpyDeviceToHost, ctx->stream));
      SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      { int rc2 = add_trial_times(ba, ba->ev); if (rc2) return rc2; }
      if (ba->opt.debug)
        fprintf(stderr, "[svs_ba] solve phases: init %.1f us, forward %.1f us (pivot wave: load+row update %.1f, eliminate+emit %.1f, - %.1f, barrier wait %.1f), backward %.1f us\n",
                h[5], h[6], h[8], h[9], h[10], h[11], h[7]);
      const bool fail = h[3] != 0.0;
      if (qmax == 0) { currentChi = h[4]; if (it == 0) st.chi2_init = currentChi; }
      double chi_t = 0, scale_l = 0;
      for (int i = 0; i < SC_SLOTS; ++i) { chi_t += h[SC_CHI + i]; scale_l += h[SC_SCL + i]; }
      double tempChi = fail ? 1.7976931348623157e308 : chi_t;
      rho = currentChi - tempChi;
      double scale = scale_l + h[2] + 1e-3;
      rho /= scale;
      ++st.trials;
      if (rho > 0 && std::isfinite(tempChi)) {
        const double q = 2 * rho - 1;
        double alpha = 1. - q * q * q;                // same expression as ba_lm_kernel (g2o: pow(2 rho - 1, 3), <= 1 ulp apart)
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2; currentChi = tempChi; ++st.accepted;
        ba->cur = 1 - ba->cur;                      // discardTop: trial state becomes current
        // keep the other buffer's untouched landmarks coherent: psi_trial of landmarks without
        // edges on this rank was never written, but both buffers started equal and only owned
        // landmarks ever change, so nothing to copy.
      } else {
        lambda *= ni; ni *= 2;                      // pop: current state stays
      }
      ++qmax;
    } while (rho < 0 && qmax < prm.max_trials);
    ++st.iterations;
    st.chi2_final = currentChi;
    if (qmax == prm.max_trials || rho == 0) { ok = false; st.terminated = 1; }
  }
  st.lambda_final = lambda;
  if (stats) *stats = st;
  return SVS_OK;
}

extern "C" int svs_ba_optimize(svs_ba *ba, svs_allreduce_fn allreduce, void *user, svs_ba_stats *stats) {
  OptRun R;
  int rc = optimize_begin(ba, allreduce, user, R);
  if (rc) return rc;
  return optimize_finish(ba, R, stats);
}

// Throughput mode: n independent windows, each on its own context (= its own stream), all enqueued before the first wait -- the
// kernels of different windows overlap on the device (a 50-keyframe window keeps 2 of 256 CUs busy during its solve).
extern "C" int svs_ba_optimize_batch(svs_ba *const *bas, int n, svs_ba_stats *stats) {
  if (!bas || n < 1) return SVS_ERR_INVALID;
  std::vector<OptRun> runs((size_t)n);
  for (int i = 0; i < n; ++i) {
    if (!bas[i]) return SVS_ERR_INVALID;
    for (int j = 0; j < i; ++j) if (bas[j]->ctx == bas[i]->ctx) { bas[i]->ctx->err = "svs_ba_optimize_batch: every window needs its own context (stream)"; return SVS_ERR_INVALID; }
    const int rc = optimize_begin(bas[i], nullptr, nullptr, runs[i]);
    if (rc) return rc;
  }
  int rc_all = SVS_OK;
  for (int i = 0; i < n; ++i) {
    const int rc = optimize_finish(bas[i], runs[i], stats ? stats + i : nullptr);
    if (rc && !rc_all) rc_all = rc;
  }
  return rc_all;
}

extern "C" int svs_ba_get_state(svs_ba *ba, double *h_poses, double *h_psi) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && ba->d_poses[0] && ba->problem_valid);
  SVS_DEVICE(ctx);
  if (h_poses) SVS_HIP(ctx, hipMemcpyAsync(h_poses, ba->d_poses[ba->cur], sizeof(double) * 12 * (size_t)ba->P, hipMemcpyDeviceToHost, ctx->stream));
  if (h_psi && ba->L) SVS_HIP(ctx, hipMemcpyAsync(h_psi, ba->d_psi[ba->cur], sizeof(double) * 3 * (size_t)ba->L, hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVS_OK;
}

extern "C" int svs_ba_set_comm(svs_ba *ba, svs_comm *comm) {
  if (!ba) return SVS_ERR_INVALID;
  ba->comm = comm;
  ba->profile_ready = false;            // the co-visibility pattern must be exchanged over the new communicator
  return SVS_OK;
}

extern "C" int svs_ba_set_option(svs_ba *ba, const char *name, int value) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && name);
  auto clamp = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
  BaOptions &o = ba->opt;
  const std::string n(name);
  if (n == "no_speculation") o.no_speculation = value != 0;
  else if (n == "one_front") o.one_front = value != 0;
  else if (n == "no_fused_solve") o.no_fused_solve = value != 0;
  else if (n == "no_lds_solve") o.no_lds_solve = value != 0;
  else if (n == "no_fused_cons") o.no_fused_cons = value != 0;
  else if (n == "no_grid_solve") o.no_grid_solve = value != 0;
  else if (n == "debug") o.debug = clamp(value, 0, 2);
  else if (n == "nw") o.nw = value == 0 ? 0 : clamp(value, 4, 8);
  else if (n == "p1") o.p1 = value < 0 ? -1 : clamp(value, 0, SOLVE_MAX_P);
  else if (n == "group") o.group = value == 0 ? 0 : clamp(value, 1, WIN);
  else if (n == "grid_g") o.grid_g = clamp(value, 0, 1024);
  else if (n == "host_threads") o.host_threads = clamp(value, 0, 64);
  else SVS_REQUIRE(ctx, !"unknown option");
  ba->profile_ready = false;            // solve-kernel choice depends on the switches
  return SVS_OK;
}

extern "C" int svs_ba_info(svs_ba *ba, int32_t *solve_kind, int32_t *envelope_rows, int32_t *n_chunks, int32_t *n_wide) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && ba->problem_valid);
  if (!ba->profile_ready) { const int rc = ensure_profile(ba, ba->comm ? svs_comm_allreduce_hook : nullptr, ba->comm); if (rc) return rc; }
  if (solve_kind) *solve_kind = ba->use_fused_solve ? (ba->fuse_P1 > 0 ? 3 : 2) : (ba->use_lds_solve ? 1 : (ba->grid_G > 0 ? 4 : 0));
  if (envelope_rows) *envelope_rows = ba->env_R;
  if (n_chunks) *n_chunks = ba->n_chunks;
  if (n_wide) *n_wide = ba->n_wide;
  return SVS_OK;
}

extern "C" int svs_ba_set_timing(svs_ba *ba, int on) {
  if (!ba) return SVS_ERR_INVALID;
  ba->timing = on != 0;
  return SVS_OK;
}

extern "C" int svs_ba_kernel_times(svs_ba *ba, float *reduce_ms, float *solve_ms, float *backsub_ms, int32_t *n_launches) {
  if (!ba) return SVS_ERR_INVALID;
  if (reduce_ms) *reduce_ms = ba->t_reduce;
  if (solve_ms) *solve_ms = ba->t_solve;
  if (backsub_ms) *backsub_ms = ba->t_backsub;
  if (n_launches) *n_launches = ba->n_reduce;
  return SVS_OK;
}
