// match.hip -- guided 8x8 zero-mean-SSD matcher for gfx950.
// Replaces GuidedMatcher<StereoCamera>::match (matcher.cpp:312-398) incl. computePrediction
// (:98-142), warpAffinve (:403-458), computePatchScores/matchPatchZeroMeanSSD (:42-96),
// matchCandidates (:144-181), returnBestMatch (:183-214), createObervation
// (matcher-impl.cpp:33-51).
//
// MI355X-first design, three kernels:
//   match_pose_kernel     relative poses per (stream, keyframe)
//   match_predict_kernel  ONE LANE per candidate point: computePrediction and the local affine of warpAffinve -- the
//                         scalar f64 part (10 exact divisions) is evaluated once per point, not by 64 lanes redundantly
//   match_kernel          ONE WAVEFRONT per candidate point: lane = pixel of the 8x8 key patch (only the used centre of the
//                         reference's 10x10 warp is formed; the patch never leaves registers), texture sums by wave
//                         reduction; the quadtree of the reference is replaced by a scan of the (2R+1)^2 search window in
//                         the corner bitmap of fast.hip (a pixel is a candidate iff
//                         its score clears the emit threshold of its cell); hits are compacted in LDS and scored one lane
//                         per candidate with V_SAD_U8 / V_DOT4_U32_U8; the winner is the minimum ZNSSD, and the reference's
//                         tie-break (first hit in QuadTree::query DFS order, strict '<') is reproduced with a quadrant key
//                         computed only on ties -- bit-exact without building a tree.
// f64 geometry is compiled with -ffp-contract=off so it rounds like the host oracle.
#include "common.h"

#include "fast_view.h"
#include <algorithm>

namespace {

__device__ __forceinline__ void d_pose_act(const double *T, const double *x, double *y) {
  double a = T[0] * x[0] + T[1] * x[1] + T[2] * x[2] + T[3];
  double b = T[4] * x[0] + T[5] * x[1] + T[6] * x[2] + T[7];
  double c = T[8] * x[0] + T[9] * x[1] + T[10] * x[2] + T[11];
  y[0] = a; y[1] = b; y[2] = c;
}
__device__ __forceinline__ void d_pose_mul(const double *A, const double *B, double *C) {
  double t[12];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) t[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
    t[4 * i + 3] += A[4 * i + 3];
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) C[i] = t[i];
}
__device__ __forceinline__ void d_pose_inv(const double *A, double *B) {
  double t[12];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[4 * i + j] = A[4 * j + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[4 * i + 3] = -(t[4 * i] * A[3] + t[4 * i + 1] * A[7] + t[4 * i + 2] * A[11]);
#pragma unroll
  for (int i = 0; i < 12; ++i) B[i] = t[i];
}
__device__ __forceinline__ bool d_in_frame(const svs_cam &c, int u, int v, int border) {
  return u >= border && v >= border && u < c.w - border && v < c.h - border;
}
__device__ __forceinline__ void d_warp_f(const double *T, double depth, const svs_cam &cam, double ku, double kv, double *o) {
  double p[3] = {depth * ((ku - cam.cx) / cam.f), depth * ((kv - cam.cy) / cam.f), depth * 1.0};
  double q[3];
  d_pose_act(T, p, q);
  o[0] = cam.f * (q[0] / q[2]) + cam.cx;
  o[1] = cam.f * (q[1] / q[2]) + cam.cy;
}
// DFS-order key of QuadTree::query (quadtree.h:510-544,693-708): recursive halving of the level
// box, x-quadrant bit above y-quadrant bit.  The reference decides a quadrant with
// rel = 1 - (x0 + w - p)/w < 0.5 in doubles; for integer p and box sizes W/2^d that test is exactly
// p < x0 + w/2 (all quantities are dyadic, the quotient is never within an ulp of 1/2), so the key
// is computed in 2^-12 fixed point with integer compares (tests/test_oracle_cpu.py checks the order
// against the real tree).
__device__ __forceinline__ unsigned quad_key(int px, int py, int W, int H) {
  int bx = 0, by = 0, bw = W << 12, bh = H << 12;
  const int X = px << 12, Y = py << 12;
  unsigned key = 0;
#pragma unroll
  for (int d = 0; d < 12; ++d) {
    bw >>= 1; bh >>= 1;
    const unsigned hx = X >= bx + bw, hy = Y >= by + bh;
    key = (key << 2) | (hx << 1) | hy;
    bx += hx ? bw : 0;
    by += hy ? bh : 0;
  }
  return key;
}

// quad_key(ax, ay) < quad_key(bx, by) without forming the keys: the two points share their box as long as their digits agree, so ONE box is halved until the
// digits differ (a rolled loop, a dozen registers: match_kernel3 calls it behind wave-uniform branches -- ties of the ZNSSD are rare -- and has no registers
// to spare for two unrolled key computations there)
__device__ __forceinline__ bool quad_less(int ax, int ay, int bx, int by, int W, int H) {
  int x0 = 0, y0 = 0, bw = W << 12, bh = H << 12;
  const int AX = ax << 12, AY = ay << 12, BX = bx << 12, BY = by << 12;
  bool less = false;
#pragma unroll 1
  for (int d = 0; d < 12; ++d) {
    bw >>= 1; bh >>= 1;
    const unsigned ahx = AX >= x0 + bw, ahy = AY >= y0 + bh, bhx = BX >= x0 + bw, bhy = BY >= y0 + bh;
    const unsigned da = (ahx << 1) | ahy, db = (bhx << 1) | bhy;
    if (da != db) { less = da < db; break; }
    x0 += ahx ? bw : 0;
    y0 += ahy ? bh : 0;
  }
  return less;
}
// per-point result of computePrediction + the local affine of warpAffinve (everything that is one scalar evaluation per
// point): produced one LANE per point by match_predict_kernel, consumed one WAVE per point by match_kernel
struct alignas(16) PointPred {      // 128 bytes = one cache line, fields grouped into 16-byte loads (match_kernel3 reads it with vector loads)
  int32_t status, ui, vi, lvl;
  double inv[4];          // inverse of the local affine A (matcher.cpp:415-426)
  double key_uv[2];       // anchor_obs_pyr
  // for match_kernel2/3: the keyframe's level image (saves the dependent read of the keyframe record) and where the search window's rows start in the corner
  // bitmap of fast.hip: the padded bit position of the window's first in-image column and the one cell-column boundary the window may cross
  const uint8_t *kimg;
  int32_t kstride, xb;
  int32_t yb;
  int32_t pb_lo;          // max(ui - R, 0) + gap * (cell column of it)
  int32_t kfi, pad_;
  double xyz_actkey[3];
  double pad2_;
};
static_assert(sizeof(PointPred) == 128, "PointPred layout");
// what match_kernel3 needs of a pyramid level, as a table in memory: its 16-lane groups work on points of different levels, so the level index is a vector
// register there and the per-level kernel arguments cannot be picked with scalar indexing (written by match_pose_kernel)
struct LevelTab {
  const uint8_t *bm, *cimg;
  size_t bm_bstride, cur_bstride;
  int32_t bm_stride, cstride, w, h, xhi, yhi, gap, pad_;
};
struct MatchParams {
  svs_match_args a;
  FastView fv;
  const double *kf_T;     // [n_batch][n_kf][24]: T_cur_from_anchor, T_actkey_from_anchor (match_pose_kernel)
  PointPred *pred;        // [n_batch][n_pts]
  const int32_t *order;   // [n_batch][n_pts] or NULL: the order in which match_kernel3 takes the points of a stream (match_order_kernel)
  int32_t *keys;          // [n_batch][n_pts] or NULL: the bucket of every point (written by match_predict_kernel, read by match_order_kernel)
  int32_t ord_sx, ord_sy, ord_nbx, ord_nb;      // its buckets: level * ord_nb + (vi >> ord_sy) * ord_nbx + (ui >> ord_sx); bucket 3 * ord_nb = points that are not searched
  LevelTab *lt;           // [SVS_NUM_PYR_LEVELS]
  // optional (the one-call front end, few keyframes): the tracked pose and the active keyframe's pose per stream -- match_predict_kernel<true> then forms the two poses of
  // matcher.cpp:326-330 and the per-keyframe relative poses itself (what frontend_pose_kernel + match_pose_kernel did in two launches of their own between tracker and matcher)
  const double *src_T, *src_Ta;
  int swz;                // match_kernel3 takes its blocks in XCD-contiguous order (common.h: xcd_contiguous)
};

// The two relative poses a candidate needs depend only on (camera stream, anchor keyframe), not on
// the point: T_cur_from_anchor = T_cur_from_w * T_anchor_from_w^-1 (matcher.cpp:113) and
// (T_anchor_from_w * T_w_from_actkey)^-1 (matcher.cpp:393-394).  One lane per pair, same operation
// order as the per-point formulation => bit-identical values, ~250 f64 ops less per wavefront.
__global__ void match_pose_kernel(MatchParams M, double *__restrict__ out) {
  const svs_match_args &A = M.a;
  const int kf = blockIdx.x * blockDim.x + threadIdx.x, slot = blockIdx.y;
  if (blockIdx.x == 0 && slot == 0) {
#pragma unroll
    for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l)
      if ((int)threadIdx.x == l && l < M.fv.n_levels) {
        LevelTab t;
        t.bm = M.fv.bm[l]; t.cimg = A.d_cur_pyr[l]; t.bm_bstride = M.fv.bm_bstride[l]; t.cur_bstride = A.cur_bstride[l];
        t.bm_stride = M.fv.bm_stride[l]; t.cstride = A.cur_stride[l]; t.w = A.cam_vec[l].w; t.h = A.cam_vec[l].h;
        t.xhi = min(M.fv.gx[l] * M.fv.cell_w[l], A.cam_vec[l].w - 6); t.yhi = min(M.fv.gy[l] * M.fv.cell_h[l], A.cam_vec[l].h - 6);      // isInFrame(uv, 6) and inside the cell grid
        t.gap = M.fv.bm_gap[l]; t.pad_ = 0;
        M.lt[l] = t;
      }
  }
  if (kf >= A.n_kf) return;
  double kfT[12], Tcw[12], Twk[12], t0[12], t1[12];
  for (int i = 0; i < 12; ++i) { kfT[i] = A.d_kfs[(size_t)slot * A.kf_bstride + kf].T_anchor_from_w[i]; Tcw[i] = A.d_T_cur_from_w[(size_t)slot * 12 + i]; Twk[i] = A.d_T_w_from_actkey[(size_t)slot * 12 + i]; }
  d_pose_inv(kfT, t0);
  d_pose_mul(Tcw, t0, t1);
  double *o = out + ((size_t)slot * A.n_kf + kf) * 24;
  for (int i = 0; i < 12; ++i) o[i] = t1[i];
  d_pose_mul(kfT, Twk, t0);
  d_pose_inv(t0, t1);
  for (int i = 0; i < 12; ++i) o[12 + i] = t1[i];
}

// computePrediction (matcher.cpp:98-142) and the affine set-up of warpAffinve (:403-426), one lane per candidate point.
// The same f64 expressions as before, evaluated once per point instead of redundantly by the 64 lanes of its wave.
constexpr int PRED_FUSE_MAX_KF = 8;
template <bool FUSE>
__global__ __launch_bounds__(64) void match_predict_kernel(MatchParams M) {
  const svs_match_args &A = M.a;
  const int ip = blockIdx.x * 64 + threadIdx.x, slot = blockIdx.y;
  __shared__ double s_kfT[FUSE ? PRED_FUSE_MAX_KF : 1][24];
  if constexpr (FUSE) {
    if (blockIdx.x == 0 && slot == 0) {                  // the level table match_kernel3 reads (match_pose_kernel's other job)
#pragma unroll
      for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l)
        if ((int)threadIdx.x == l && l < M.fv.n_levels) {
          LevelTab t;
          t.bm = M.fv.bm[l]; t.cimg = A.d_cur_pyr[l]; t.bm_bstride = M.fv.bm_bstride[l]; t.cur_bstride = A.cur_bstride[l];
          t.bm_stride = M.fv.bm_stride[l]; t.cstride = A.cur_stride[l]; t.w = A.cam_vec[l].w; t.h = A.cam_vec[l].h;
          t.xhi = min(M.fv.gx[l] * M.fv.cell_w[l], A.cam_vec[l].w - 6); t.yhi = min(M.fv.gy[l] * M.fv.cell_h[l], A.cam_vec[l].h - 6);
          t.gap = M.fv.bm_gap[l]; t.pad_ = 0;
          M.lt[l] = t;
        }
    }
    const int kf = threadIdx.x;
    if (kf < A.n_kf) {
      // T_cur_from_w = T_cur_from_actkey * T_actkey_from_w, T_w_from_actkey = T_actkey_from_w^-1 (matcher.cpp:326-330; frontend_pose_kernel's operation order), then the
      // two relative poses of keyframe kf (match_pose_kernel's): the same expressions in the same order, i.e. the same bits, computed per block instead of fetched
      double Tc[12], Ta[12], Tcw[12], Twk[12], kfT[12], t0[12], t1[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) { Tc[i] = M.src_T[(size_t)slot * 12 + i]; Ta[i] = M.src_Ta[(size_t)slot * 12 + i]; kfT[i] = A.d_kfs[(size_t)slot * A.kf_bstride + kf].T_anchor_from_w[i]; }
      d_pose_mul(Tc, Ta, Tcw);
      d_pose_inv(Ta, Twk);
      d_pose_inv(kfT, t0);
      d_pose_mul(Tcw, t0, t1);
#pragma unroll
      for (int i = 0; i < 12; ++i) s_kfT[kf][i] = t1[i];
      d_pose_mul(kfT, Twk, t0);
      d_pose_inv(t0, t1);
#pragma unroll
      for (int i = 0; i < 12; ++i) s_kfT[kf][12 + i] = t1[i];
    }
    __syncthreads();
  }
  if (ip >= A.n_pts) return;
  const svs_candidate_point ap = A.d_pts[(size_t)slot * A.pts_bstride + ip];
  PointPred pr;
  pr.status = SVS_MATCH_OK; pr.ui = pr.vi = 0; pr.lvl = ap.anchor_level; pr.kfi = ap.kf_index; pr.pad_ = 0; pr.pad2_ = 0;
  pr.inv[0] = pr.inv[1] = pr.inv[2] = pr.inv[3] = 0; pr.key_uv[0] = ap.anchor_obs_pyr[0]; pr.key_uv[1] = ap.anchor_obs_pyr[1];
  pr.xyz_actkey[0] = pr.xyz_actkey[1] = pr.xyz_actkey[2] = 0;
  pr.kimg = nullptr; pr.kstride = 0; pr.xb = pr.yb = 0x7fffffff; pr.pb_lo = 0;
  if (ap.kf_index < 0 || ap.kf_index >= A.n_kf) pr.status = SVS_MATCH_NO_ANCHOR;
  else if (ap.anchor_level < 0 || ap.anchor_level >= M.fv.n_levels) pr.status = SVS_MATCH_NONE;  // no feature_tree for that level
  if (pr.status == SVS_MATCH_OK) {
    const svs_cam cam = A.cam_vec[ap.anchor_level];
    const double *kfT = FUSE ? s_kfT[ap.kf_index] : M.kf_T + ((size_t)slot * A.n_kf + ap.kf_index) * 24;
    double T_cur_from_anchor[12], xyz_cur[3];
#pragma unroll
    for (int i = 0; i < 12; ++i) T_cur_from_anchor[i] = kfT[i];
    d_pose_act(T_cur_from_anchor, ap.xyz_anchor, xyz_cur);
    const double uv0 = cam.f * (xyz_cur[0] / xyz_cur[2]) + cam.cx;
    const double uv1 = cam.f * (xyz_cur[1] / xyz_cur[2]) + cam.cy;
    const double depth_cur = 1. / xyz_cur[2], depth_anchor = 1. / ap.xyz_anchor[2];
    if (!d_in_frame(cam, (int)ap.anchor_obs_pyr[0], (int)ap.anchor_obs_pyr[1], 4)) pr.status = SVS_MATCH_BORDER;
    else if (depth_cur > depth_anchor * 3 || depth_anchor > depth_cur * 3) pr.status = SVS_MATCH_DEPTH;
    else if (!(fabs(uv0) < 1e9) || !(fabs(uv1) < 1e9)) pr.status = SVS_MATCH_NONE;
    if (pr.status == SVS_MATCH_OK) {
      pr.ui = (int)uv0; pr.vi = (int)uv1;
      // f(uv), f(uv + e_x), f(uv + e_y) (matcher.cpp:411-413); x + 0.0 is exact
      double f0[2], fu[2], fv[2];
      d_warp_f(T_cur_from_anchor, ap.xyz_anchor[2], cam, ap.anchor_obs_pyr[0] + 0.0, ap.anchor_obs_pyr[1] + 0.0, f0);
      d_warp_f(T_cur_from_anchor, ap.xyz_anchor[2], cam, ap.anchor_obs_pyr[0] + 1.0, ap.anchor_obs_pyr[1] + 0.0, fu);
      d_warp_f(T_cur_from_anchor, ap.xyz_anchor[2], cam, ap.anchor_obs_pyr[0] + 0.0, ap.anchor_obs_pyr[1] + 1.0, fv);
      const double a00 = fu[0] - f0[0], a01 = fu[1] - f0[1], a10 = fv[0] - f0[0], a11 = fv[1] - f0[1];
      const double invdet = 1.0 / (a00 * a11 - a01 * a10);
      pr.inv[0] = a11 * invdet; pr.inv[1] = -a01 * invdet; pr.inv[2] = -a10 * invdet; pr.inv[3] = a00 * invdet;
      double T_actkey_from_anchor[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) T_actkey_from_anchor[i] = kfT[12 + i];
      d_pose_act(T_actkey_from_anchor, ap.xyz_anchor, pr.xyz_actkey);
      const int lvl = ap.anchor_level, R = A.search_radius;
      const svs_keyframe &kf = A.d_kfs[(size_t)slot * A.kf_bstride + ap.kf_index];
      pr.kimg = kf.pyr[lvl]; pr.kstride = kf.stride[lvl];
      const int gx = M.fv.gx[lvl], gy = M.fv.gy[lvl], cw = M.fv.cell_w[lvl], chh = M.fv.cell_h[lvl];
      const int x0 = pr.ui - R, y0 = pr.vi - R;
      int cxa = 0, cya = 0;
      for (int q = 1; q < gx; ++q) cxa += max(x0, 0) >= q * cw;
      for (int q = 1; q < gy; ++q) cya += max(y0, 0) >= q * chh;
      pr.xb = cxa + 1 < gx ? (cxa + 1) * cw : 0x7fffffff; pr.yb = cya + 1 < gy ? (cya + 1) * chh : 0x7fffffff;
      pr.pb_lo = max(x0, 0) + M.fv.bm_gap[lvl] * cxa;
    }
  }
  M.pred[(size_t)slot * A.n_pts + ip] = pr;
  if (M.keys) {           // bucket of the point for match_order_kernel: (level, cell of the predicted position); the points that are not searched go last
    int b = 3 * M.ord_nb;
    if (pr.status == SVS_MATCH_OK) {
      const int bx = min(max(pr.ui, 0) >> M.ord_sx, M.ord_nbx - 1), by = max(pr.vi, 0) >> M.ord_sy;
      b = min(pr.lvl * M.ord_nb + by * M.ord_nbx + bx, 3 * M.ord_nb - 1);
    }
    M.keys[(size_t)slot * A.n_pts + ip] = b;
  }
}

// ---- the corner bitmap of fast.hip (fast.hip, LevelDev): pixel (x, y) of cell column ci is bit x + gap * ci of row y; rows end in >= 8 zero bytes ----
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
typedef uint32_t U2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ U4 ld_u4u(const uint8_t *p) { U4 v; __builtin_memcpy(&v, p, 16); return v; }      // 16 bytes from any address: one global_load_dwordx4
__device__ __forceinline__ U2 ld_u2u(const uint8_t *p) { U2 v; __builtin_memcpy(&v, p, 8); return v; }
// one pixel (any x inside the image), the cell column found by compares: the general matcher, whose windows may be wider than a cell
__device__ __forceinline__ bool corner_bit(const uint8_t *row, int x, int cw, int gx, int gap) {
  int ci = 0;
  for (int q = 1; q < gx; ++q) ci += x >= q * cw;
  const int pb = x + gap * ci;
  return ((row[pb >> 3] >> (pb & 7)) & 1u) != 0u;
}
// up to 25 window positions x0 .. of one row, bit k = position x0 + k, from the 8 bytes `q` at byte pb_lo >> 3 of the row: pb_lo = the padded position of max(x0, 0),
// xb = the one cell-column boundary the positions may cross (windows narrower than a cell).  Positions left of the image read 0; the bits behind the window are the
// caller's to mask.
__device__ __forceinline__ uint32_t window_bits(U2 q, int pb_lo, int x0, int xb, int gap) {
  const unsigned long long raw = (((unsigned long long)q.y << 32) | q.x) >> (pb_lo & 7);
  const unsigned long long t = raw << min(max(-x0, 0), 63);      // (a window far left of the image: everything shifted out)
  // positions in front of the boundary; no boundary (xb = INT_MAX, last cell column) or one behind the window: all 32.  (Compared before subtracting: xb - x0 must not wrap.)
  const int nb = xb >= x0 + 32 ? 32 : max(xb - x0, 0);
  const uint32_t lowmask = nb >= 32 ? 0xffffffffu : (1u << nb) - 1u;
  return ((uint32_t)t & lowmask) | ((uint32_t)(t >> gap) & ~lowmask);
}
// bits of the positions x0 + k inside [xlo, xhi), k < n
__device__ __forceinline__ uint32_t span_mask(int x0, int n, int xlo, int xhi) {
  const int a = max(xlo - x0, 0), b = min(min(xhi - x0, n), 32);
  if (b <= a) return 0u;
  const uint32_t hi = b >= 32 ? 0xffffffffu : (1u << b) - 1u;
  return hi & ~((1u << a) - 1u);
}

constexpr int WAVES_PER_BLOCK = 4;

__global__ __launch_bounds__(256) void match_kernel(MatchParams M, svs_match_result *__restrict__ out) {
  __shared__ int s_cand[WAVES_PER_BLOCK][320];      // per-wave list of window hits waiting to be scored ((y << 16) | x): < 64 carried + <= 256 per trip
  const svs_match_args &A = M.a;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ip = blockIdx.x * WAVES_PER_BLOCK + wave;
  const int slot = blockIdx.y;
  if (ip >= A.n_pts) return;                       // wave-uniform
  const PointPred pp = M.pred[(size_t)slot * A.n_pts + __builtin_amdgcn_readfirstlane(ip)];      // wave-uniform record
  svs_match_result *o = &out[(size_t)slot * A.out_bstride + ip];
  const int R = A.search_radius;
  const int init_dist = A.thr_mean * A.thr_mean * 64;
  int status = pp.status;
  double xyz_actkey[3] = {0, 0, 0}, obs[3] = {0, 0, 0};
  int best = init_dist, bu = 0, bv = 0;
  if (status == SVS_MATCH_OK) {
    {
      // all 64 lanes work on the same point: wave-uniform (scalar) indices, the keyframe record and the per-level tables
      // are read with scalar loads
      const int kfi = __builtin_amdgcn_readfirstlane(pp.kfi);
      const int lvl = __builtin_amdgcn_readfirstlane(pp.lvl);
      const svs_keyframe *kfp = A.d_kfs + (size_t)slot * A.kf_bstride + kfi;
      const svs_cam cam = A.cam_vec[lvl];
      const int ui = pp.ui, vi = pp.vi;
      // ---- warpAffinve: 10x10 patch, lanes take pixels lane and lane+64 -------------------
      const double i00 = pp.inv[0], i01 = pp.inv[1], i10 = pp.inv[2], i11 = pp.inv[3];
      const double key_u = pp.key_uv[0], key_v = pp.key_uv[1];
      const uint8_t *kimg = kfp->pyr[lvl];
      const int kstride = kfp->stride[lvl];
      // Only the centre 8x8 of the reference's 10x10 warp is ever read (KEY_PATCH, matcher.cpp:376-381) and every warped
      // pixel depends on its own coordinates alone: lane = pixel (row lane>>3, column lane&7) of that centre, one pass,
      // the value stays in a register.
      int keyv;
      {
        const int iy = (lane >> 3) + 1, ix = (lane & 7) + 1;
        const double dx = ix - 5, dy = iy - 5;
        const double r0 = (i00 * dx + i01 * dy) + key_u;
        const double r1 = (i10 * dx + i11 * dy) + key_v;
        const double x = floor(r0), y = floor(r1);
        uint8_t val;
        if (!(x >= 0) || !(y >= 0) || x + 1 >= cam.w || y + 1 >= cam.h) val = 0;
        else {
          const double sx = r0 - x, sy = r1 - y;
          const double wx0 = 1 - sx, wx1 = sx, wy0 = 1 - sy, wy1 = sy;
          const int xi = (int)x, yi = (int)y;
          const double v00 = kimg[(size_t)yi * kstride + xi], v01 = kimg[(size_t)(yi + 1) * kstride + xi];
          const double v10 = kimg[(size_t)yi * kstride + xi + 1], v11 = kimg[(size_t)(yi + 1) * kstride + xi + 1];
          const double s = (wx0 * wy0) * v00 + (wx0 * wy1) * v01 + (wx1 * wy0) * v10 + (wx1 * wy1) * v11;
          val = (uint8_t)(s < 255. ? s : 255.);
        }
        keyv = val;
      }
      const int sumA = wave_sum_i32(keyv), sumAA = wave_sum_i32(keyv * keyv);
      if (sumA * sumA - sumAA < A.thr_std * A.thr_std * 64) status = SVS_MATCH_TEXTURE;
      else {
        // ---- window scan + ZNSSD of every candidate corner ----------------------------------
        const uint8_t *bm = M.fv.bm[lvl] + (size_t)slot * M.fv.bm_bstride[lvl];
        const int bm_stride = M.fv.bm_stride[lvl], gap = M.fv.bm_gap[lvl];
        const int gx = M.fv.gx[lvl], gy = M.fv.gy[lvl], cw = M.fv.cell_w[lvl], chh = M.fv.cell_h[lvl];
        const uint8_t *cimg = A.d_cur_pyr[lvl] + (size_t)slot * A.cur_bstride[lvl];
        const int cstride = A.cur_stride[lvl];
        const int side = 2 * R + 1;
        // key patch as 16 packed dwords (8 rows x 2), wave-uniform: the four bytes of a dword sit in four adjacent lanes
        // (quad shuffles), lane 4i then holds dword i
        const uint32_t kd = (uint32_t)keyv | ((uint32_t)__shfl_down(keyv, 1, 64) << 8) | ((uint32_t)__shfl_down(keyv, 2, 64) << 16) |
                            ((uint32_t)__shfl_down(keyv, 3, 64) << 24);
        uint32_t key4[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) key4[i] = (uint32_t)__builtin_amdgcn_readlane((int)kd, 4 * i);
        // window scan: hits are compacted into a per-wave LDS list; whenever it holds >= 64 corners (and at the end) they
        // are scored one lane per candidate: 8 rows x 2 unaligned dwords of the current image, sums with V_SAD_U8 /
        // V_DOT4_U32_U8 against the wave-uniform key dwords -- no cross-lane reduction per candidate
        int gbest = 0x7fffffff, gx_ = 0, gy_ = 0;      // per-lane running best
        unsigned gkey = 0xffffffffu;
        int ncand = 0;                                 // wave-uniform fill of s_cand
        // a lane tests four horizontally adjacent window positions against the corner bitmap
        const int ngrp = (side + 3) >> 2, ntask = side * ngrp;
        const float inv_ngrp = 1.0f / (float)ngrp;
        const int xlo = 6, xhi = min(gx * cw, cam.w - 6), ylo = 6, yhi = min(gy * chh, cam.h - 6);      // isInFrame(uv, 6) and inside the cell grid
        for (int p0 = 0; p0 < ntask; p0 += 64) {
          const int task = p0 + lane;
          const bool valid = task < ntask;
          const int wy = (int)(((float)task + 0.5f) * inv_ngrp), g4 = 4 * (task - wy * ngrp);      // exact for task < 2^12
          const int cy = vi - R + wy, cx0 = ui - R + g4;
          const bool row_ok = valid && cy >= ylo && cy < yhi;
          const uint8_t *brow = bm + (size_t)(row_ok ? cy : 0) * bm_stride;
          int nh = 0;
          unsigned long long mk[4];
          bool hk[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int cx = cx0 + k;
            hk[k] = row_ok && g4 + k < side && cx >= xlo && cx < xhi && corner_bit(brow, cx, cw, gx, gap);
            mk[k] = __ballot(hk[k]);
          }
          const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (hk[k]) s_cand[wave][ncand + nh + __popcll(mk[k] & lt)] = (cy << 16) | (cx0 + k);
            nh += __popcll(mk[k]);
          }
          ncand += nh;
          const bool last = p0 + 64 >= ntask;
          if (ncand < 64 && !last) continue;                           // wave-uniform
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_s_waitcnt(0xc07f);
          for (int base = 0; base < ncand; base += 64) {
            if (base + lane < ncand) {
              const int packed = s_cand[wave][base + lane];
              const int hx = packed & 0xffff, hy = packed >> 16;
              const uint8_t *p = cimg + (size_t)(hy - 4) * cstride + (hx - 4);
              uint32_t sB = 0, sBB = 0, sAB = 0;
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                uint32_t v0, v1;
                __builtin_memcpy(&v0, p + (size_t)r * cstride, 4);
                __builtin_memcpy(&v1, p + (size_t)r * cstride + 4, 4);
                sB = __builtin_amdgcn_sad_u8(v0, 0u, sB); sB = __builtin_amdgcn_sad_u8(v1, 0u, sB);
                sBB = __builtin_amdgcn_udot4(v0, v0, sBB, false); sBB = __builtin_amdgcn_udot4(v1, v1, sBB, false);
                sAB = __builtin_amdgcn_udot4(v0, key4[2 * r], sAB, false); sAB = __builtin_amdgcn_udot4(v1, key4[2 * r + 1], sAB, false);
              }
              const int iB = (int)sB;
              const int z = sumAA - 2 * (int)sAB - (int)sBB - (sumA * sumA - 2 * sumA * iB - iB * iB) / 64;
              // strict '<' in DFS order (matcher.cpp:173)  <=>  lexicographic min of (z, key); z must also beat thr_mean.
              // The quadrant key is only needed to break ties, so it is computed lazily (ties are rare).
              if (z < init_dist) {
                if (z < gbest) { gbest = z; gx_ = hx; gy_ = hy; }
                else if (z == gbest && quad_key(hx, hy, cam.w, cam.h) < quad_key(gx_, gy_, cam.w, cam.h)) { gx_ = hx; gy_ = hy; }
              }
            }
          }
          ncand = 0;
          __builtin_amdgcn_wave_barrier();
        }
        // lexicographic min of (z, quadrant key) across the wave: minimum z first (6 shuffle steps); only if several lanes
        // tie on it -- rare -- are their keys computed and compared; the winner's corner is read with readlane
        int zmin = gbest;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) zmin = min(zmin, __shfl_xor(zmin, o, 64));
        if (zmin != 0x7fffffff) {
          unsigned long long tie = __ballot(gbest == zmin);
          if (__popcll(tie) > 1) {
            unsigned k = gbest == zmin ? quad_key(gx_, gy_, cam.w, cam.h) : 0xffffffffu;
            unsigned kmin = k;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o, 64));
            tie = __ballot(gbest == zmin && k == kmin);
          }
          const int win = __ffsll((long long)tie) - 1;
          gbest = zmin;
          gx_ = __builtin_amdgcn_readlane(gx_, win);
          gy_ = __builtin_amdgcn_readlane(gy_, win);
          gkey = 0;
        }
        unsigned bestkey = gkey;
        if (bestkey != 0xffffffffu) { best = gbest; bu = gx_; bv = gy_; }
        xyz_actkey[0] = pp.xyz_actkey[0]; xyz_actkey[1] = pp.xyz_actkey[1]; xyz_actkey[2] = pp.xyz_actkey[2];      // point in the active keyframe (match_predict_kernel)
        if (bestkey == 0xffffffffu) { status = SVS_MATCH_NONE; best = init_dist; }
        else {
          const double inv_factor = 1.0 / (double)(1 << lvl);
          const float *disp = A.d_disp + (size_t)slot * A.disp_bstride;
          const double d = disp[(size_t)(bv << lvl) * A.disp_stride + (bu << lvl)] * inv_factor;
          if (d > 0) {
            const double sc = (double)(1 << lvl);
            const float fu_ = (float)bu, fv_ = (float)bv;
            obs[0] = fu_ * sc; obs[1] = fv_ * sc; obs[2] = (fu_ - d) * sc;
          } else status = SVS_MATCH_NO_DISP;
        }
      }
    }
  }
  if (lane == 0) {
    svs_match_result r;
    r.status = status; r.u = bu; r.v = bv; r.znssd = best;
    r.obs[0] = obs[0]; r.obs[1] = obs[1]; r.obs[2] = obs[2];
    r.xyz_actkey[0] = xyz_actkey[0]; r.xyz_actkey[1] = xyz_actkey[1]; r.xyz_actkey[2] = xyz_actkey[2];
    *o = r;
  }
}


// ---- round 3: the same matcher with a leaner window scan ---------------------------------------------------------------------------
// match_kernel above spends ~45 % of its instructions in the scan of the (2R+1)^2 window: two trips of 64 tasks at four positions each, four
// ballots per trip, per-lane cell look-ups with loops over the grid and vector loads of the emit thresholds.  Here the window's cell geometry is
// resolved ONCE per point on the scalar unit -- a window is narrower than a cell, so it meets at most one vertical and one horizontal cell
// boundary: four thresholds in scalar registers --, a lane tests EIGHT adjacent positions of a row from one 8-byte read (17 x 3 tasks = one trip
// at R = 8), and the sparse hits (~10 of 289) go to the per-wave list through an LDS counter (their order is irrelevant: the winner is the
// lexicographic minimum of (ZNSSD, quadrant key)).  Everything else -- warp, texture gate, scoring, tie-break, observation -- is the code above.
constexpr int CAND_CAP2 = 64 * 8;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void match_kernel2(MatchParams M, svs_match_result *__restrict__ out) {
  __shared__ int s_cand[WAVES_PER_BLOCK][CAND_CAP2];
  __shared__ int s_ncand[WAVES_PER_BLOCK];
  const svs_match_args &A = M.a;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ip = blockIdx.x * WAVES_PER_BLOCK + wave;
  const int slot = blockIdx.y;
  if (ip >= A.n_pts) return;                       // wave-uniform
  const PointPred pp = M.pred[(size_t)slot * A.n_pts + __builtin_amdgcn_readfirstlane(ip)];      // wave-uniform record
  svs_match_result *o = &out[(size_t)slot * A.out_bstride + ip];
  const int R = A.search_radius;
  const int init_dist = A.thr_mean * A.thr_mean * 64;
  int status = pp.status;
  double xyz_actkey[3] = {0, 0, 0}, obs[3] = {0, 0, 0};
  int best = init_dist, bu = 0, bv = 0;
  if (status == SVS_MATCH_OK) {
    const int lvl = __builtin_amdgcn_readfirstlane(pp.lvl);
    const svs_cam cam = A.cam_vec[lvl];
    const int ui = __builtin_amdgcn_readfirstlane(pp.ui), vi = __builtin_amdgcn_readfirstlane(pp.vi);
    // ---- the first (at R <= 10: the only) trip of the window scan is requested NOW: its addresses depend on the prediction alone, so the bitmap
    // read travels together with the keyframe pixels of the warp below instead of behind the texture gate
    const uint8_t *bm = M.fv.bm[lvl] + (size_t)slot * M.fv.bm_bstride[lvl];
    const int bm_stride = M.fv.bm_stride[lvl], gap = M.fv.bm_gap[lvl];
    const int gx = M.fv.gx[lvl], gy = M.fv.gy[lvl], cw = M.fv.cell_w[lvl], chh = M.fv.cell_h[lvl];
    const int side = 2 * R + 1, x0 = ui - R, y0 = vi - R;
    const int xlo = 6, xhi = min(gx * cw, cam.w - 6), ylo = 6, yhi = min(gy * chh, cam.h - 6);      // isInFrame(uv, 6) and inside the cell grid
    const int nseg = (side + 7) >> 3, ntask = side * nseg;
    const float inv_nseg = 1.0f / (float)nseg;
    // a task = eight adjacent positions of a window row: the 8 bytes of the row's bitmap at the segment's first in-image column (`lo`: the bits, `hi`: the padded bit
    // position of that column; the segment crosses at most one cell-column boundary)
    auto read8 = [&](int task, int &cy, int &cx0, int &g8, bool &row_ok, uint32_t &lo, uint32_t &hi) {
      const int wy = (int)(((float)task + 0.5f) * inv_nseg);      // exact for task < 2^12
      g8 = 8 * (task - wy * nseg);
      cy = y0 + wy; cx0 = x0 + g8;
      row_ok = task < ntask && cy >= ylo && cy < yhi && cx0 < cam.w && cx0 + 8 > 0;
      lo = 0; hi = 0;
      if (row_ok) {
        const int xl = max(cx0, 0);
        int ci = 0;
        for (int q = 1; q < gx; ++q) ci += xl >= q * cw;
        const int pb = xl + gap * ci;
        const U2 w = ld_u2u(bm + (size_t)cy * bm_stride + (pb >> 3));
        lo = window_bits(w, pb, cx0, ci + 1 < gx ? (ci + 1) * cw : 0x7fffffff, gap) & span_mask(cx0, min(8, side - g8), xlo, xhi);
      }
    };
    int cy_f, cx0_f, g8_f; bool row_ok_f; uint32_t lo_f, hi_f;
    read8(lane, cy_f, cx0_f, g8_f, row_ok_f, lo_f, hi_f);
    // ---- warpAffinve: lane = pixel of the centre 8x8 of the 10x10 patch (as in match_kernel)
    const double i00 = pp.inv[0], i01 = pp.inv[1], i10 = pp.inv[2], i11 = pp.inv[3];
    const double key_u = pp.key_uv[0], key_v = pp.key_uv[1];
    const uint8_t *kimg = pp.kimg;
    const int kstride = pp.kstride;
    int keyv;
    {
      const int iy = (lane >> 3) + 1, ix = (lane & 7) + 1;
      const double dx = ix - 5, dy = iy - 5;
      const double r0 = (i00 * dx + i01 * dy) + key_u;
      const double r1 = (i10 * dx + i11 * dy) + key_v;
      const double x = floor(r0), y = floor(r1);
      uint8_t val;
      if (!(x >= 0) || !(y >= 0) || x + 1 >= cam.w || y + 1 >= cam.h) val = 0;
      else {
        const double sx = r0 - x, sy = r1 - y;
        const double wx0 = 1 - sx, wx1 = sx, wy0 = 1 - sy, wy1 = sy;
        const int xi = (int)x, yi = (int)y;
        const double v00 = kimg[(size_t)yi * kstride + xi], v01 = kimg[(size_t)(yi + 1) * kstride + xi];
        const double v10 = kimg[(size_t)yi * kstride + xi + 1], v11 = kimg[(size_t)(yi + 1) * kstride + xi + 1];
        const double s = (wx0 * wy0) * v00 + (wx0 * wy1) * v01 + (wx1 * wy0) * v10 + (wx1 * wy1) * v11;
        val = (uint8_t)(s < 255. ? s : 255.);
      }
      keyv = val;
    }
    const int sumA = wave_sum_i32(keyv), sumAA = wave_sum_i32(keyv * keyv);
    if (sumA * sumA - sumAA < A.thr_std * A.thr_std * 64) status = SVS_MATCH_TEXTURE;
    else {
      const uint8_t *cimg = A.d_cur_pyr[lvl] + (size_t)slot * A.cur_bstride[lvl];
      const int cstride = A.cur_stride[lvl];
      const uint32_t kd = (uint32_t)keyv | ((uint32_t)__shfl_down(keyv, 1, 64) << 8) | ((uint32_t)__shfl_down(keyv, 2, 64) << 16) |
                          ((uint32_t)__shfl_down(keyv, 3, 64) << 24);
      uint32_t key4[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) key4[i] = (uint32_t)__builtin_amdgcn_readlane((int)kd, 4 * i);
      int gbest = 0x7fffffff, gx_ = 0, gy_ = 0;      // per-lane running best
      unsigned gkey = 0xffffffffu;
      for (int p0 = 0; p0 < ntask; p0 += 64) {
        if (lane == 0) s_ncand[wave] = 0;
        int cy = cy_f, cx0 = cx0_f, g8 = g8_f; bool row_ok = row_ok_f; uint32_t lo = lo_f, hi = hi_f;
        if (p0) read8(p0 + lane, cy, cx0, g8, row_ok, lo, hi);
        unsigned hits = lo;
        (void)hi; (void)g8;
        hits = row_ok ? hits : 0u;
        __builtin_amdgcn_wave_barrier();
        while (hits) {                                 // sparse: ~10 hits per window
          const int k = __ffs((int)hits) - 1;
          hits &= hits - 1;
          const int pos = atomicAdd(&s_ncand[wave], 1);
          s_cand[wave][pos] = (cy << 16) | (cx0 + k);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int ncand = s_ncand[wave];
        for (int base = 0; base < ncand; base += 64) {
          if (base + lane < ncand) {
            const int packed = s_cand[wave][base + lane];
            const int hx = packed & 0xffff, hy = packed >> 16;
            const uint8_t *p = cimg + (size_t)(hy - 4) * cstride + (hx - 4);
            uint32_t sB = 0, sBB = 0, sAB = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              uint32_t v0, v1;
              __builtin_memcpy(&v0, p + (size_t)r * cstride, 4);
              __builtin_memcpy(&v1, p + (size_t)r * cstride + 4, 4);
              sB = __builtin_amdgcn_sad_u8(v0, 0u, sB); sB = __builtin_amdgcn_sad_u8(v1, 0u, sB);
              sBB = __builtin_amdgcn_udot4(v0, v0, sBB, false); sBB = __builtin_amdgcn_udot4(v1, v1, sBB, false);
              sAB = __builtin_amdgcn_udot4(v0, key4[2 * r], sAB, false); sAB = __builtin_amdgcn_udot4(v1, key4[2 * r + 1], sAB, false);
            }
            const int iB = (int)sB;
            const int z = sumAA - 2 * (int)sAB - (int)sBB - (sumA * sumA - 2 * sumA * iB - iB * iB) / 64;
            // strict '<' in DFS order (matcher.cpp:173)  <=>  lexicographic min of (z, key); z must also beat thr_mean.
            if (z < init_dist) {
              if (z < gbest) { gbest = z; gx_ = hx; gy_ = hy; }
              else if (z == gbest && quad_key(hx, hy, cam.w, cam.h) < quad_key(gx_, gy_, cam.w, cam.h)) { gx_ = hx; gy_ = hy; }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      int zmin = gbest;
#pragma unroll
      for (int o2 = 1; o2 < 64; o2 <<= 1) zmin = min(zmin, __shfl_xor(zmin, o2, 64));
      if (zmin != 0x7fffffff) {
        unsigned long long tie = __ballot(gbest == zmin);
        if (__popcll(tie) > 1) {
          unsigned k = gbest == zmin ? quad_key(gx_, gy_, cam.w, cam.h) : 0xffffffffu;
          unsigned kmin = k;
#pragma unroll
          for (int o2 = 1; o2 < 64; o2 <<= 1) kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o2, 64));
          tie = __ballot(gbest == zmin && k == kmin);
        }
        const int win = __ffsll((long long)tie) - 1;
        gbest = zmin;
        gx_ = __builtin_amdgcn_readlane(gx_, win);
        gy_ = __builtin_amdgcn_readlane(gy_, win);
        gkey = 0;
      }
      xyz_actkey[0] = pp.xyz_actkey[0]; xyz_actkey[1] = pp.xyz_actkey[1]; xyz_actkey[2] = pp.xyz_actkey[2];
      if (gkey == 0xffffffffu) { status = SVS_MATCH_NONE; best = init_dist; }
      else {
        best = gbest; bu = gx_; bv = gy_;
        const double inv_factor = 1.0 / (double)(1 << lvl);
        const float *disp = A.d_disp + (size_t)slot * A.disp_bstride;
        const double d = disp[(size_t)(bv << lvl) * A.disp_stride + (bu << lvl)] * inv_factor;
        if (d > 0) {
          const double sc = (double)(1 << lvl);
          const float fu_ = (float)bu, fv_ = (float)bv;
          obs[0] = fu_ * sc; obs[1] = fv_ * sc; obs[2] = (fu_ - d) * sc;
        } else status = SVS_MATCH_NO_DISP;
      }
    }
  }
  if (lane == 0) {
    svs_match_result r;
    r.status = status; r.u = bu; r.v = bv; r.znssd = best;
    r.obs[0] = obs[0]; r.obs[1] = obs[1]; r.obs[2] = obs[2];
    r.xyz_actkey[0] = xyz_actkey[0]; r.xyz_actkey[1] = xyz_actkey[1]; r.xyz_actkey[2] = xyz_actkey[2];
    *o = r;
  }
}

// ---- round 3, second step: FOUR points per wave ------------------------------------------------------------------------------------
// With one wave per point (both kernels above) ~600 wave instructions go into a point, most of them with a handful of useful lanes: the scoring has
// one lane per candidate corner (2-10 of 64), the window scan two hits per 64 lanes, the 64-lane sums and the 16 readlanes of the key patch serve one
// point.  Here a point has a DPP row: 16 lanes = the 8 x 8 key patch as 16 dwords (lane = row, half), which is the layout V_SAD_U8 / V_DOT4_U32_U8
// want: a candidate costs one dword load and three dot products per lane and a 16-lane butterfly (quad_perm, row_half_mirror, row_mirror: DPP
// modifiers, no LDS); the key patch never moves; the (2R+1)^2 window is one row of the corner bitmap per lane (17 rows = 16 lanes + a shared last row: 17 bits of an
// 8-byte read).  Four points
// share every instruction.  Needs 2R+1 <= 17 (the reference's radii: 8 on the CPU build, 4 on the CUDA build); wider windows take match_kernel2.
// The order in which match_kernel3 takes the points of a stream.  The caller's lists come in hash order (the reference iterates tr1::unordered_maps): four
// points that share a wave then sit anywhere in the image, every wave touches its own bitmap, key-patch and image lines, and the kernel runs at the miss
// rate of the vector L1 / L2.  A counting sort by (level, 16 x 16 pixel cell of the predicted position) makes neighbours in the order neighbours in the image
// (match 0.83 -> 0.69 ms per 512 x 2000 points) and collects the points that are not searched (behind the camera, outside the frame) in waves of their own.
// One workgroup per stream; the order inside a cell is whatever the atomics make it -- every point is matched on its own and written to its own record, so
// the results do not depend on it.
constexpr int ORD_MAX_BUCKETS = 6144;
__global__ __launch_bounds__(256) void match_order_kernel(MatchParams M, int32_t *__restrict__ order) {
  __shared__ int s_cnt[ORD_MAX_BUCKETS + 1];
  __shared__ int s_part[256];
  const int slot = blockIdx.x, n = M.a.n_pts, tid = threadIdx.x, nb_tot = 3 * M.ord_nb + 1;
  const int32_t *keys = M.keys + (size_t)slot * n;
  for (int b = tid; b < nb_tot; b += 256) s_cnt[b] = 0;
  __syncthreads();
  auto bucket = [&](int ip) { return keys[ip]; };
  for (int ip = tid; ip < n; ip += 256) atomicAdd(&s_cnt[bucket(ip)], 1);
  __syncthreads();
  // exclusive scan of the bucket counts: every thread a contiguous run, then the runs' totals
  const int per = (nb_tot + 255) / 256, b0 = tid * per, b1 = min(nb_tot, b0 + per);
  int run = 0;
  for (int b = b0; b < b1; ++b) run += s_cnt[b];
  s_part[tid] = run;
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int t = 0; t < 256; ++t) { const int c = s_part[t]; s_part[t] = acc; acc += c; } }
  __syncthreads();
  int acc = s_part[tid];
  for (int b = b0; b < b1; ++b) { const int c = s_cnt[b]; s_cnt[b] = acc; acc += c; }
  __syncthreads();
  for (int ip = tid; ip < n; ip += 256) order[(size_t)slot * n + atomicAdd(&s_cnt[bucket(ip)], 1)] = ip;
}

constexpr int M3_GROUPS = 16;                      // points per 256-lane block
constexpr int M3_CAND_CAP = 17 * 17;
__device__ __forceinline__ uint32_t row16_sum(uint32_t v) {      // all-reduce over the 16 lanes of a DPP row
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);       // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);      // row_half_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);      // row_mirror
  return v;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void match_kernel3(MatchParams M, svs_match_result *__restrict__ out) {
  __shared__ uint16_t s_cand[M3_GROUPS][M3_CAND_CAP + 7];      // per point: window positions (wy << 5 | wx) waiting to be scored
  __shared__ int s_ncand[M3_GROUPS];
  __shared__ uint32_t s_key[M3_GROUPS][16];                   // per point: the warped 8 x 8 key patch, 16 packed dwords
  const svs_match_args &A = M.a;
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  // the points of a stream come in image order (match_order_kernel): neighbouring blocks read neighbouring bitmap rows, key patches and image lines.  In XCD-contiguous
  // order an XCD works through whole streams and those lines meet in ONE L2
  unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
  if (M.swz) wg = xcd_contiguous(wg, gridDim.x * gridDim.y);
  const int ipos = (int)(wg % gridDim.x) * M3_GROUPS + grp, slot = wg / gridDim.x;
  const bool live = ipos < A.n_pts;
  const int ip = live && M.order ? M.order[(size_t)slot * A.n_pts + ipos] : ipos;
  const PointPred *pp = M.pred + (size_t)slot * A.n_pts + (live ? ip : A.n_pts - 1);
  int status = pp->status;
  bool go = live && status == SVS_MATCH_OK;
  const int R = A.search_radius, side = 2 * R + 1;
  const int init_dist = A.thr_mean * A.thr_mean * 64;
  if (sub == 0) s_ncand[grp] = 0;
  const int lvl = go ? pp->lvl : 0;
  const LevelTab *Lp = M.lt + lvl;
  const int x0 = pp->ui - R, y0 = pp->vi - R;
  const int Lw = Lp->w, Lh = Lp->h;
  // The kernel is bound by the L1 (it works per instruction and cache line touched) and short of registers at 8 waves per SIMD: wide requests, and
  // values re-read from the (cached) point record / level table where they are needed rather than carried.  (Round 6, measured: carrying the bilinear
  // fractions from (2) to (4) instead of recomputing them -- 60 f64 instructions less, 80 VGPRs, 6 waves -- costs 0.05 ms per 512 x 2000 points: occupancy, not VALU.)
  // ---- (1) the window's corner-bitmap row of this lane and the 17th row (the same 8 bytes for the 16 lanes of the point: one request): 8-byte requests from the byte of the
  // window's first in-image column (fast.hip: a bit per pixel, cell columns on dword boundaries, rows padded with zeros)
  U2 w0 = {0u, 0u}, w1 = {0u, 0u};
  const int cy = y0 + sub, cyx = y0 + 16;
  {
    const uint8_t *bm = Lp->bm + (size_t)slot * Lp->bm_bstride + (pp->pb_lo >> 3);
    const int bm_stride = Lp->bm_stride, yhi = Lp->yhi;
    const bool col_ok = go && x0 < Lw;
    if (col_ok && sub < side && cy >= 6 && cy < yhi) w0 = ld_u2u(bm + (size_t)cy * bm_stride);
    if (col_ok && side > 16 && cyx >= 6 && cyx < yhi) w1 = ld_u2u(bm + (size_t)cyx * bm_stride);
  }
  // ---- (2) warpAffinve, requests: the centre 8x8 of the 10x10 patch (KEY_PATCH, matcher.cpp:376-381), lane = (row, half): four pixels = one packed
  // dword.  The common case -- a warp close to a translation -- has the lane's four samples on one source row pair within 8 columns: two 8-byte
  // requests instead of sixteen single bytes; anything else takes the byte path.  px[]: the two 8-byte rows, or per pixel v00 | v10 << 8 | v01 << 16
  // | v11 << 24; flags: bit j = pixel j inside the image, bit 4 = row-pair case; offs: byte j = column offset of pixel j in the rows.
  const int prow = sub >> 1, pc0 = 4 * (sub & 1);
  uint32_t px0 = 0, px1 = 0, px2 = 0, px3 = 0, px4 = 0, px5 = 0, flags = 0, offs = 0;
  if (go) {
    const double i00 = pp->inv[0], i01 = pp->inv[1], i10 = pp->inv[2], i11 = pp->inv[3];
    const double key_u = pp->key_uv[0], key_v = pp->key_uv[1];
    const uint8_t *kimg = pp->kimg;
    const int kstride = pp->kstride;
    const double dy = prow - 4;                    // iy - 5, iy = row + 1
    int xis[4], yis[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double dx = pc0 + j - 4;
      const double r0 = (i00 * dx + i01 * dy) + key_u;
      const double r1 = (i10 * dx + i11 * dy) + key_v;
      const double x = floor(r0), y = floor(r1);
      const bool in = x >= 0 && y >= 0 && !(x + 1 >= Lw) && !(y + 1 >= Lh);
      flags |= in ? 1u << j : 0u;
      xis[j] = in ? (int)x : 0; yis[j] = in ? (int)y : 0;
    }
    // (round 4) the four samples may also straddle TWO source row pairs (a slightly rotated or scaled warp crosses a row boundary inside most waves: with the
    // row-pair case alone, one lane in 64 on the byte path made the whole wave issue its sixteen byte loads): three 8-byte rows y, y+1, y+2, and a bit per
    // pixel that says which pair is its own (flags bit 5 + j)
    const int ymin = min(min(yis[0], yis[1]), min(yis[2], yis[3]));
    const bool oneRow = flags == 15u && (unsigned)(yis[0] - ymin) <= 1u && (unsigned)(yis[1] - ymin) <= 1u && (unsigned)(yis[2] - ymin) <= 1u &&
                        (unsigned)(yis[3] - ymin) <= 1u && (unsigned)(xis[1] - xis[0]) <= 6u && (unsigned)(xis[2] - xis[0]) <= 6u &&
                        (unsigned)(xis[3] - xis[0]) <= 6u && xis[0] + 8 <= Lw;
    if (oneRow) {
      const uint8_t *q = kimg + (size_t)ymin * kstride + xis[0];
      const U2 ra = ld_u2u(q), rb = ld_u2u(q + kstride), rc = ld_u2u(q + (ymin + 2 < Lh ? 2 : 1) * (ptrdiff_t)kstride);      // (row y+2 is only used by pixels on row y+1, whose y+2 is inside)
      px0 = ra.x; px1 = ra.y; px2 = rb.x; px3 = rb.y; px4 = rc.x; px5 = rc.y;
      flags |= 16u | (uint32_t)(yis[0] - ymin) << 5 | (uint32_t)(yis[1] - ymin) << 6 | (uint32_t)(yis[2] - ymin) << 7 | (uint32_t)(yis[3] - ymin) << 8;
      offs = (uint32_t)(xis[1] - xis[0]) << 8 | (uint32_t)(xis[2] - xis[0]) << 16 | (uint32_t)(xis[3] - xis[0]) << 24;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (flags >> j & 1u) {
          const uint8_t *q = kimg + (size_t)yis[j] * kstride + xis[j];
          const uint32_t t = (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[kstride] << 16 | (uint32_t)q[kstride + 1] << 24;
          if (j == 0) px0 = t; else if (j == 1) px1 = t; else if (j == 2) px2 = t; else px3 = t;
        }
    }
  }
  // ---- (3) hits of the window: the set bits of the lane's row inside [6, xhi) (isInFrame(uv, 6) and inside the cell grid).  Before the texture gate, whose verdict
  // only decides whether they are scored.
  {
    const int xb = pp->xb, pb_lo = pp->pb_lo, gap = Lp->gap;
    const uint32_t span = span_mask(x0, side, 6, Lp->xhi);
    uint32_t m = window_bits(w0, pb_lo, x0, xb, gap) & span;
    const uint32_t m17 = window_bits(w1, pb_lo, x0, xb, gap) & span;
    uint32_t ex = (m17 >> sub) & (sub == 15 ? 3u : 1u);      // the 17th row: lane = column, the last lane also the 17th column
    __builtin_amdgcn_wave_barrier();
    while (m) {
      const int k = __ffs((int)m) - 1;
      m &= m - 1;
      s_cand[grp][atomicAdd(&s_ncand[grp], 1)] = (uint16_t)((sub << 5) | k);
    }
    while (ex) {
      const int k = sub + ((ex & 1u) ? 0 : 1);
      ex &= ex - 1;
      s_cand[grp][atomicAdd(&s_ncand[grp], 1)] = (uint16_t)((16 << 5) | k);
    }
  }
  // ---- (3b) the pixels of the point's first 16 hits are requested NOW: their round trip runs under the bilinear arithmetic of (4) (a lane = a hit, eight 8-byte rows from
  // ONE running address -- the barrier keeps the compiler from forming eight addresses first: sixteen registers the kernel does not have at 8 waves per SIMD)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  const int cstride = Lp->cstride;
  const uint8_t *pbase = Lp->cimg + (size_t)slot * Lp->cur_bstride + (ptrdiff_t)(y0 - 4) * cstride + (x0 - 4);
  // (FIRST rows only: all eight would cost sixteen registers across (4) that the kernel does not have)
  auto fetch16 = [&](int c0, int n_, int r_lo, int r_hi, U2 (&v)[8], int &wx, int &wy) __attribute__((always_inline)) {
    const bool act = c0 + sub < n_;
    const int code = act ? s_cand[grp][c0 + sub] : 0;
    wx = code & 31; wy = code >> 5;
    const uint8_t *q = pbase + (ptrdiff_t)(wy + r_lo) * cstride + wx;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r >= r_lo && r < r_hi) {
        v[r] = act ? ld_u2u(q) : U2{0u, 0u};
        q += cstride;
        asm volatile("" : "+v"(q));
      }
  };
  constexpr int M3_PRE = 4;
  U2 v[8];
  int wx, wy;
  fetch16(0, go ? s_ncand[grp] : 0, 0, M3_PRE, v, wx, wy);
  // ---- (4) warpAffinve, arithmetic: the sample coordinates again (from the point record: 12 doubles not held across (3)), bilinear weights as the
  // reference forms them
  uint32_t keyd = 0;
  if (go) {
    const PointPred *pq = pp;
    asm volatile("" : "+v"(pq));
    const double i00 = pq->inv[0], i01 = pq->inv[1], i10 = pq->inv[2], i11 = pq->inv[3];
    const double key_u = pq->key_uv[0], key_v = pq->key_uv[1];
    const double dy = prow - 4;
    const bool oneRow = (flags & 16u) != 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double dx = pc0 + j - 4;
      const double r0 = (i00 * dx + i01 * dy) + key_u;
      const double r1 = (i10 * dx + i11 * dy) + key_v;
      const double sx = r0 - floor(r0), sy = r1 - floor(r1);
      uint32_t t = j == 0 ? px0 : j == 1 ? px1 : j == 2 ? px2 : px3;
      if (oneRow) {
        const uint32_t o = (offs >> (8 * j)) & 0xffu, sel = o | ((o + 1u) << 8) | 0x0c0c0000u;      // V_PERM: bytes o, o + 1 of an 8-byte row, zeros above
        const bool lower = (flags >> (5 + j) & 1u) != 0u;
        t = __builtin_amdgcn_perm(lower ? px3 : px1, lower ? px2 : px0, sel) | __builtin_amdgcn_perm(lower ? px5 : px3, lower ? px4 : px2, sel) << 16;
      }
      const double v00 = (double)(t & 0xffu), v10 = (double)(t >> 8 & 0xffu), v01 = (double)(t >> 16 & 0xffu), v11 = (double)(t >> 24);
      const double wx0 = 1 - sx, wx1 = sx, wy0 = 1 - sy, wy1 = sy;
      const double sv = (wx0 * wy0) * v00 + (wx0 * wy1) * v01 + (wx1 * wy0) * v10 + (wx1 * wy1) * v11;
      const uint32_t val = (flags >> j & 1u) ? (uint32_t)(uint8_t)(sv < 255. ? sv : 255.) : 0u;
      keyd |= val << (8 * j);
    }
  }
  const int sumA = (int)row16_sum(__builtin_amdgcn_sad_u8(keyd, 0u, 0u)), sumAA = (int)row16_sum(__builtin_amdgcn_udot4(keyd, keyd, 0u, false));
  if (go && sumA * sumA - sumAA < A.thr_std * A.thr_std * 64) { status = SVS_MATCH_TEXTURE; go = false; }
  const bool textured = go;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  // ---- (5) ZNSSD of every hit, ONE LANE PER HIT (round 6): the 16 lanes of a point take 16 of its hits at a time, a lane sums its hit's whole 8 x 8 patch -- eight 8-byte
  // rows against the key rows, which all lanes of the point read from LDS (a broadcast) -- with six V_SAD_U8 / V_DOT4_U32_U8 per row and NO cross-lane step per hit; the
  // point's best is one 16-lane butterfly at the end.  (Rounds 3-5: two hits per step, lane = patch row, an 8-lane butterfly per pair of hits: ~55 wave instructions per
  // pair and as many steps as the busiest of the wave's four points has pairs -- 0.18 of the kernel's 0.49 ms at 512 x 2000 points.)  The first 16 hits' pixels were
  // requested in (3b).
  const int n = go ? s_ncand[grp] : 0;
  s_key[grp][sub] = keyd;                          // key rows: dwords 2 r, 2 r + 1 = row r
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  int best = 0x7fffffff, bu = 0, bv = 0;
  for (int c0 = 0; __ballot(c0 < n) != 0ull; c0 += 16) {
    fetch16(c0, n, c0 ? 0 : M3_PRE, 8, v, wx, wy);      // the rest of the rows (all of them behind the first 16 hits: rare)
    const bool act = c0 + sub < n;
    uint32_t iBl = 0, sBB = 0, sAB = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t k0 = s_key[grp][2 * r], k1 = s_key[grp][2 * r + 1];
      iBl = __builtin_amdgcn_sad_u8(v[r].y, 0u, __builtin_amdgcn_sad_u8(v[r].x, 0u, iBl));
      sBB = __builtin_amdgcn_udot4(v[r].y, v[r].y, __builtin_amdgcn_udot4(v[r].x, v[r].x, sBB, false), false);
      sAB = __builtin_amdgcn_udot4(v[r].y, k1, __builtin_amdgcn_udot4(v[r].x, k0, sAB, false), false);
    }
    const int iB = (int)iBl;
    const int z = sumAA - (int)(sBB + 2u * sAB) - (sumA * sumA - 2 * sumA * iB - iB * iB) / 64;      // sBB + 2 sAB <= 3 * 64 * 255^2
    // strict '<' in DFS order (matcher.cpp:173)  <=>  lexicographic min of (z, quadrant key); z must also beat thr_mean.
    // The quadrant keys are only needed to break ties, which are rare: behind a wave-uniform branch, so that they are not evaluated (predicated) per hit.
    const int hx = x0 + wx, hy = y0 + wy;
    const bool good = act && z < init_dist;
    if (__ballot(good && z == best) != 0ull) {
      int hx_ = hx, bu_ = bu;
      asm volatile("" : "+v"(hx_), "+v"(bu_));      // (keeps the key arithmetic inside the branch: it is cheap enough for the compiler to speculate it)
      if (good && z == best && quad_less(hx_, hy, bu_, bv, Lw, Lh)) { bu = hx; bv = hy; }
    }
    if (good && z < best) { best = z; bu = hx; bv = hy; }
  }
  // the 16 lanes of the point meet: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- after every step both partners hold the same winner
#define M3_MEET(CTRL)                                                                                                                                        \
  {                                                                                                                                                          \
    const int oz = __builtin_amdgcn_update_dpp(0, best, CTRL, 0xf, 0xf, true), ou = __builtin_amdgcn_update_dpp(0, bu, CTRL, 0xf, 0xf, true),                \
              ov = __builtin_amdgcn_update_dpp(0, bv, CTRL, 0xf, 0xf, true);                                                                                 \
    if (__ballot(oz == best && best != 0x7fffffff && (ou != bu || ov != bv)) != 0ull) {                                                                      \
      int ou_ = ou, bu_ = bu;                                                                                                                                \
      asm volatile("" : "+v"(ou_), "+v"(bu_));                                                                                                               \
      if (oz == best && best != 0x7fffffff && quad_less(ou_, ov, bu_, bv, Lw, Lh)) { bu = ou; bv = ov; }                                                     \
    }                                                                                                                                                        \
    if (oz < best) { best = oz; bu = ou; bv = ov; }                                                                                                          \
  }
  M3_MEET(0xB1) M3_MEET(0x4E) M3_MEET(0x141) M3_MEET(0x140)
#undef M3_MEET
  double obs0 = 0, obs1 = 0, obs2 = 0;
  if (go) {
    if (best == 0x7fffffff) { status = SVS_MATCH_NONE; bu = bv = 0; }
    else {
      const double inv_factor = 1.0 / (double)(1 << lvl);
      const float *disp = A.d_disp + (size_t)slot * A.disp_bstride;
      const double dd = disp[(size_t)(bv << lvl) * A.disp_stride + (bu << lvl)] * inv_factor;
      if (dd > 0) {
        const double sc = (double)(1 << lvl);
        const float fu_ = (float)bu, fv_ = (float)bv;
        obs0 = fu_ * sc; obs1 = fv_ * sc; obs2 = (fu_ - dd) * sc;
      } else status = SVS_MATCH_NO_DISP;
    }
  }
  if (live && sub == 0) {
    svs_match_result r;
    r.status = status; r.u = bu; r.v = bv; r.znssd = best == 0x7fffffff ? init_dist : best;
    r.obs[0] = obs0; r.obs[1] = obs1; r.obs[2] = obs2;
    const PointPred *pe = pp;
    asm volatile("" : "+v"(pe));      // (read here, not carried from the top of the kernel in registers the loop above needs)
    r.xyz_actkey[0] = textured ? pe->xyz_actkey[0] : 0.0; r.xyz_actkey[1] = textured ? pe->xyz_actkey[1] : 0.0; r.xyz_actkey[2] = textured ? pe->xyz_actkey[2] : 0.0;
    out[(size_t)slot * A.out_bstride + ip] = r;
  }
}

}  // namespace

extern "C" int svs_match(svs_ctx *ctx, const svs_match_args *a, svs_fast *f, svs_match_result *d_out) {
  SVS_REQUIRE(ctx, ctx && a && f);
  SVS_DEVICE(ctx);
  SVS_REQUIRE(ctx, a->n_pts >= 0 && a->n_batch >= 1 && a->search_radius >= 0 && a->search_radius <= 31);
  if (a->n_pts == 0) return SVS_OK;                 // empty ap_map: nothing to append (matcher.cpp:332)
  SVS_REQUIRE(ctx, d_out && a->d_pts && a->d_kfs);
  MatchParams M;
  M.a = *a;
  if (!M.a.pts_bstride) M.a.pts_bstride = (size_t)a->n_pts;
  if (!M.a.out_bstride) M.a.out_bstride = (size_t)a->n_pts;
  SVS_REQUIRE(ctx, M.a.pts_bstride >= (size_t)a->n_pts && M.a.out_bstride >= (size_t)a->n_pts);
  M.fv = svs_fast_view_internal(f);
  SVS_REQUIRE(ctx, M.fv.n_levels >= 1 && a->n_kf >= 1);
  // the per-call tables live in the context (one buffer: relative poses per (stream, keyframe), then the predictions per point)
  const size_t kf_bytes = (((size_t)a->n_batch * a->n_kf * 24 * sizeof(double)) + 255) & ~(size_t)255;
  const size_t pred_bytes = (size_t)a->n_batch * a->n_pts * sizeof(PointPred);
  const size_t lt_bytes = 256;
  static_assert(sizeof(LevelTab) * SVS_NUM_PYR_LEVELS <= 256, "level table");
  const size_t ord_bytes = (((size_t)a->n_batch * a->n_pts * sizeof(int32_t)) + 255) & ~(size_t)255;
  void *buf = nullptr;
  { const int rc = svs_ctx_match_scratch(ctx, lt_bytes + kf_bytes + 2 * ord_bytes + pred_bytes, &buf); if (rc) return rc; }
  M.lt = static_cast<LevelTab *>(buf);
  double *kf_T = reinterpret_cast<double *>(static_cast<char *>(buf) + lt_bytes);
  M.kf_T = kf_T;
  int32_t *order = reinterpret_cast<int32_t *>(static_cast<char *>(buf) + lt_bytes + kf_bytes);
  M.pred = reinterpret_cast<PointPred *>(static_cast<char *>(buf) + lt_bytes + kf_bytes + 2 * ord_bytes);
  M.order = nullptr; M.keys = nullptr;
  // (option "match_legacy": 0 = the fastest kernel that applies, 1 = match_kernel, 2 = match_kernel2 where it applies)
  bool lean = ctx->match_legacy != 1;
  for (int l = 0; l < M.fv.n_levels; ++l) lean = lean && 2 * a->search_radius + 1 <= std::min(M.fv.cell_w[l], M.fv.cell_h[l]);
  const bool k3 = lean && ctx->match_legacy == 0 && 2 * a->search_radius + 1 <= 17;
  const bool ordered = k3 && ctx->match_order && a->n_pts >= 64 && (size_t)a->n_batch * a->n_pts >= 32768;      // a batch: one more launch (~10 us) is not worth it for a stream or two
  if (ordered) {      // cells of 16 x 16 pixels of level 0 (coarser for frames that would need more than ORD_MAX_BUCKETS buckets)
    int sx = 4, sy = 4;      // (16 x 16 ... 64 x 32 pixel cells measure the same within 2 %: 0.69 - 0.71 ms)
    auto nb = [&]() { return ((M.fv.w[0] + (1 << sx) - 1) >> sx) * ((M.fv.h[0] + (1 << sy) - 1) >> sy); };
    while (3 * nb() + 1 > ORD_MAX_BUCKETS) { if (sx <= sy + 1) ++sx; else ++sy; }
    M.ord_sx = sx; M.ord_sy = sy; M.ord_nbx = (M.fv.w[0] + (1 << sx) - 1) >> sx; M.ord_nb = nb();
    M.keys = order + ord_bytes / sizeof(int32_t);
  }
  M.src_T = ctx->match_src_T; M.src_Ta = ctx->match_src_Ta;
  M.swz = ctx->xcd_swizzle;
  if (M.src_T && M.src_Ta && a->n_kf <= PRED_FUSE_MAX_KF) {      // one launch instead of three (frontend_pose_kernel, match_pose_kernel, match_predict_kernel): the one-call front end
    hipLaunchKernelGGL(match_predict_kernel<true>, dim3(div_up(a->n_pts, 64), a->n_batch), dim3(64), 0, ctx->stream, M);
    SVS_LAUNCH_CHECK(ctx);
  } else {
    SVS_REQUIRE(ctx, a->d_T_cur_from_w && a->d_T_w_from_actkey);
    hipLaunchKernelGGL(match_pose_kernel, dim3(div_up(a->n_kf, 64), a->n_batch), dim3(64), 0, ctx->stream, M, kf_T);
    SVS_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(match_predict_kernel<false>, dim3(div_up(a->n_pts, 64), a->n_batch), dim3(64), 0, ctx->stream, M);
    SVS_LAUNCH_CHECK(ctx);
  }
  dim3 grid(div_up(a->n_pts, WAVES_PER_BLOCK), a->n_batch), block(64 * WAVES_PER_BLOCK);
  // the lean scan resolves a window's cells once per point: it needs windows narrower than a cell (always so for the reference's grids and radii)
  if (k3) {
    if (ordered) {
      hipLaunchKernelGGL(match_order_kernel, dim3(a->n_batch), dim3(256), 0, ctx->stream, M, order);
      SVS_LAUNCH_CHECK(ctx);
      M.order = order;
    }
    hipLaunchKernelGGL(match_kernel3, dim3(div_up(a->n_pts, M3_GROUPS), a->n_batch), dim3(256), 0, ctx->stream, M, d_out);
  }
  else if (lean) hipLaunchKernelGGL(match_kernel2, grid, block, 0, ctx->stream, M, d_out);
  else hipLaunchKernelGGL(match_kernel, grid, block, 0, ctx->stream, M, d_out);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
