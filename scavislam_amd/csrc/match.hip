// match.hip -- guided 8x8 zero-mean-SSD matcher for gfx950.
// Replaces GuidedMatcher<StereoCamera>::match (matcher.cpp:312-398) incl. computePrediction
// (:98-142), warpAffinve (:403-458), computePatchScores/matchPatchZeroMeanSSD (:42-96),
// matchCandidates (:144-181), returnBestMatch (:183-214), createObervation
// (matcher-impl.cpp:33-51).
//
// MI355X-first design: ONE WAVEFRONT PER CANDIDATE POINT, 64 lanes = the 64 pixels of the 8x8
// patch.  The affine key-patch warp (100 f64 bilinear taps) is spread over the lanes, the three
// ZNSSD sums are wave reductions, and the quadtree of the reference is replaced by a direct scan
// of the (2R+1)^2 search window in the FAST score map produced by fast.hip: a pixel is a
// candidate iff its score clears the emit threshold of its cell.  The reference's tie-break
// (first hit in QuadTree::query DFS order wins, strict '<') is reproduced with a per-candidate
// quadrant key, so results are bit-exact without building a tree.
// f64 geometry is compiled with -ffp-contract=off so it rounds like the host oracle.
#include "common.h"

#include "fast_view.h"

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
// box with the reference's own double arithmetic; x-quadrant bit above y-quadrant bit.
__device__ __forceinline__ unsigned quad_key(double px, double py, double W, double H) {
  double bx = 0, by = 0, bw = W, bh = H;
  unsigned key = 0;
#pragma unroll 1
  for (int d = 0; d < 12; ++d) {
    double rel_x = 1 - (bx + bw - px) / bw;
    double rel_y = 1 - (by + bh - py) / bh;
    unsigned hx = !(rel_x < 0.5), hy = !(rel_y < 0.5);
    key = (key << 2) | (hx << 1) | hy;
    if (hx) bx = bx + bw * 0.5;
    if (hy) by = by + bh * 0.5;
    bw = bw * 0.5; bh = bh * 0.5;
  }
  return key;
}

struct MatchParams {
  svs_match_args a;
  FastView fv;
};

constexpr int WAVES_PER_BLOCK = 4;

__global__ __launch_bounds__(256) void match_kernel(MatchParams M, svs_match_result *__restrict__ out) {
  __shared__ uint8_t s_patch[WAVES_PER_BLOCK][104];
  const svs_match_args &A = M.a;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ip = blockIdx.x * WAVES_PER_BLOCK + wave;
  const int slot = blockIdx.y;
  if (ip >= A.n_pts) return;                       // wave-uniform
  const svs_candidate_point ap = A.d_pts[(size_t)slot * A.n_pts + ip];
  svs_match_result *o = &out[(size_t)slot * A.n_pts + ip];
  const int R = A.search_radius;
  const int init_dist = A.thr_mean * A.thr_mean * 64;
  int status = SVS_MATCH_OK;
  double xyz_actkey[3] = {0, 0, 0}, obs[3] = {0, 0, 0};
  int best = init_dist, bu = 0, bv = 0;

  if (ap.kf_index < 0 || ap.kf_index >= A.n_kf) status = SVS_MATCH_NO_ANCHOR;
  else if (ap.anchor_level < 0 || ap.anchor_level >= M.fv.n_levels) status = SVS_MATCH_NONE;  // no feature_tree for that level
  if (status == SVS_MATCH_OK) {
    // all 64 lanes work on the same point: make the indices wave-uniform (scalar) so the keyframe
    // record and the per-level tables are read with scalar loads instead of a per-lane scratch copy
    const int kfi = __builtin_amdgcn_readfirstlane(ap.kf_index);
    const int lvl = __builtin_amdgcn_readfirstlane(ap.anchor_level);
    const svs_keyframe *kfp = A.d_kfs + kfi;
    const svs_cam cam = A.cam_vec[lvl];
    double kfT[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) kfT[i] = kfp->T_anchor_from_w[i];
    double Tcw[12], Twk[12], T_w_from_anchor[12], T_cur_from_anchor[12], xyz_cur[3];
#pragma unroll
    for (int i = 0; i < 12; ++i) { Tcw[i] = A.d_T_cur_from_w[(size_t)slot * 12 + i]; Twk[i] = A.d_T_w_from_actkey[(size_t)slot * 12 + i]; }
    d_pose_inv(kfT, T_w_from_anchor);
    d_pose_mul(Tcw, T_w_from_anchor, T_cur_from_anchor);
    d_pose_act(T_cur_from_anchor, ap.xyz_anchor, xyz_cur);
    const double uv0 = cam.f * (xyz_cur[0] / xyz_cur[2]) + cam.cx;
    const double uv1 = cam.f * (xyz_cur[1] / xyz_cur[2]) + cam.cy;
    const double depth_cur = 1. / xyz_cur[2], depth_anchor = 1. / ap.xyz_anchor[2];
    if (!d_in_frame(cam, (int)ap.anchor_obs_pyr[0], (int)ap.anchor_obs_pyr[1], 4)) status = SVS_MATCH_BORDER;
    else if (depth_cur > depth_anchor * 3 || depth_anchor > depth_cur * 3) status = SVS_MATCH_DEPTH;
    else if (!(fabs(uv0) < 1e9) || !(fabs(uv1) < 1e9)) status = SVS_MATCH_NONE;
    if (status == SVS_MATCH_OK) {
      const int ui = (int)uv0, vi = (int)uv1;
      // ---- warpAffinve: 10x10 patch, lanes take pixels lane and lane+64 -------------------
      double f0[2], fu[2], fv[2];
      d_warp_f(T_cur_from_anchor, ap.xyz_anchor[2], cam, ap.anchor_obs_pyr[0], ap.anchor_obs_pyr[1], f0);
      d_warp_f(T_cur_from_anchor, ap.xyz_anchor[2], cam, ap.anchor_obs_pyr[0] + 1, ap.anchor_obs_pyr[1], fu);
      d_warp_f(T_cur_from_anchor, ap.xyz_anchor[2], cam, ap.anchor_obs_pyr[0], ap.anchor_obs_pyr[1] + 1, fv);
      const double a00 = fu[0] - f0[0], a01 = fu[1] - f0[1], a10 = fv[0] - f0[0], a11 = fv[1] - f0[1];
      const double invdet = 1.0 / (a00 * a11 - a01 * a10);
      const double i00 = a11 * invdet, i01 = -a01 * invdet, i10 = -a10 * invdet, i11 = a00 * invdet;
      const uint8_t *kimg = kfp->pyr[lvl];
      const int kstride = kfp->stride[lvl];
      for (int q = lane; q < 100; q += 64) {
        const int iy = q / 10, ix = q - iy * 10;
        const double dx = ix - 5, dy = iy - 5;
        const double r0 = (i00 * dx + i01 * dy) + ap.anchor_obs_pyr[0];
        const double r1 = (i10 * dx + i11 * dy) + ap.anchor_obs_pyr[1];
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
        s_patch[wave][q] = val;
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes of this wave landed
      const int pr = lane >> 3, pc = lane & 7;
      const int keyv = s_patch[wave][(pr + 1) * 10 + pc + 1];
      const int sumA = wave_sum_i32(keyv), sumAA = wave_sum_i32(keyv * keyv);
      if (sumA * sumA - sumAA < A.thr_std * A.thr_std * 64) status = SVS_MATCH_TEXTURE;
      else {
        // ---- window scan + ZNSSD of every candidate corner ----------------------------------
        const uint8_t *score = M.fv.score[lvl] + (size_t)slot * M.fv.score_bstride[lvl];
        const int sstride = M.fv.score_stride[lvl];
        const int *emit = M.fv.emit + (size_t)slot * M.fv.ncell_total + M.fv.cell_base[lvl];
        const int gx = M.fv.gx[lvl], gy = M.fv.gy[lvl], cw = M.fv.cell_w[lvl], chh = M.fv.cell_h[lvl];
        const uint8_t *cimg = A.d_cur_pyr[lvl] + (size_t)slot * A.cur_bstride[lvl];
        const int cstride = A.cur_stride[lvl];
        const int side = 2 * R + 1, npos = side * side;
        unsigned bestkey = 0xffffffffu;
        for (int p0 = 0; p0 < npos; p0 += 64) {
          const int pos = p0 + lane;
          int cx = 0, cy = 0;
          bool hit = false;
          if (pos < npos) {
            const int wy = pos / side, wx = pos - wy * side;
            cx = ui - R + wx; cy = vi - R + wy;
            if (cx >= 0 && cy >= 0 && cx < gx * cw && cy < gy * chh && d_in_frame(cam, cx, cy, 6)) {
              const int cell = (cy / chh) * gx + cx / cw;
              const int thr1 = min(max(emit[cell], 0), 255) + 1;
              hit = score[(size_t)cy * sstride + cx] >= thr1;
            }
          }
          unsigned key = hit ? quad_key((double)cx, (double)cy, (double)cam.w, (double)cam.h) : 0u;
          unsigned long long m = __ballot(hit);
          while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int hx = __shfl(cx, src, 64), hy = __shfl(cy, src, 64);
            const unsigned hk = __shfl(key, src, 64);
            const int b = cimg[(size_t)(hy - 4 + pr) * cstride + (hx - 4 + pc)];
            const int sumB = wave_sum_i32(b), sumBB = wave_sum_i32(b * b), sumAB = wave_sum_i32(b * keyv);
            const int z = sumAA - 2 * sumAB - sumBB - (sumA * sumA - 2 * sumA * sumB - sumB * sumB) / 64;
            // strict '<' in DFS order (matcher.cpp:173)  <=>  lexicographic min of (z, key)
            if (z < best || (bestkey != 0xffffffffu && z == best && hk < bestkey)) {
              best = z; bestkey = hk; bu = hx; bv = hy;
            }
          }
        }
        double T_anchor_from_actkey[12], T_actkey_from_anchor[12];
        d_pose_mul(kfT, Twk, T_anchor_from_actkey);
        d_pose_inv(T_anchor_from_actkey, T_actkey_from_anchor);
        d_pose_act(T_actkey_from_anchor, ap.xyz_anchor, xyz_actkey);
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

}  // namespace

extern "C" int svs_match(svs_ctx *ctx, const svs_match_args *a, svs_fast *f, svs_match_result *d_out) {
  SVS_REQUIRE(ctx, ctx && a && f && d_out);
  SVS_REQUIRE(ctx, a->n_pts >= 0 && a->n_batch >= 1 && a->search_radius >= 0 && a->search_radius <= 31);
  if (a->n_pts == 0) return SVS_OK;
  MatchParams M;
  M.a = *a;
  M.fv = svs_fast_view_internal(f);
  SVS_REQUIRE(ctx, M.fv.n_levels >= 1);
  dim3 grid(div_up(a->n_pts, WAVES_PER_BLOCK), a->n_batch), block(64 * WAVES_PER_BLOCK);
  hipLaunchKernelGGL(match_kernel, grid, block, 0, ctx->stream, M, d_out);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
