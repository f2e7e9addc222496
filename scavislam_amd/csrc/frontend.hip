// frontend.hip -- one call per camera frame: the data-parallel part of StereoFrontend::processFrame
// (stereo_frontend.cpp:183-306) for ONE stream with host buffers in and host buffers out, the way stereo_slam's main loop
// calls it (stereo_slam.cpp:705).  Chains the kernels of image.hip / dense.hip / stereo.hip / fast.hip / match.hip on the
// context's stream without a host round trip in between:
//   upload (pinned staging)  ->  FrameGrabber::preprocessing (u8 pyramid)            frame_grabber.cpp:285-336
//   -> DenseTracker::denseTrackingCpu (device-resident LM, f32 + Sobel taps fused)    stereo_frontend.cpp:191-197
//   -> calcDisparityCpu (cv::StereoBM) unless a disparity image is given              :199-225
//   -> computeFastCorners (FastGrid::detectAdaptively, 6 trials)                      :228-233
//   -> matchAndTrack: GuidedMatcher::match + calcFastMotionOnly                       :235-241, :976-1069
//   -> processMatchedPoints (reprojection gate + PointStatistics)                     :245-262, :834-974
//   -> computeDensePointCloudCpu at the refined pose                                  :298-302
//   -> one download of {pose, pass count, match records, gate records, statistics}.
// What stays with the caller is the reference's bookkeeping: keyframe switching / dropping (:265-296), list building, new
// point seeding.  The object owns the device copies of the previous frame (pyramid + reference cloud), the keyframes it was
// told to keep (Frame::clone, keyframes.h:72-83), the candidate points (ap_map) and the FAST threshold state.
#include "common.h"
#include <string.h>
#include <algorithm>
#include <vector>

namespace {
// T_cur_from_w = T_cur_from_actkey * T_actkey_from_w and T_w_from_actkey = T_actkey_from_w^-1 (matcher.cpp:326-330), formed on the
// device from the tracked pose so that the matcher needs no host round trip.  Operation order of the oracle / the host mirror:
// (a0 b0 + a1 b1) + a2 b2, then + t; this file is built without contraction.
__global__ void frontend_pose_kernel(const double *__restrict__ T_cur_from_actkey, const double *__restrict__ T_actkey_from_w,
                                     double *__restrict__ T_cur_from_w, double *__restrict__ T_w_from_actkey) {
  if (threadIdx.x != 0) return;
  const double *A = T_cur_from_actkey, *Bm = T_actkey_from_w;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) T_cur_from_w[4 * i + j] = A[4 * i] * Bm[j] + A[4 * i + 1] * Bm[4 + j] + A[4 * i + 2] * Bm[8 + j];
    T_cur_from_w[4 * i + 3] += A[4 * i + 3];
  }
  double o[12];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[4 * i + j] = Bm[4 * j + i];
  for (int i = 0; i < 3; ++i) o[4 * i + 3] = -(o[4 * i] * Bm[3] + o[4 * i + 1] * Bm[7] + o[4 * i + 2] * Bm[11]);
  for (int i = 0; i < 12; ++i) T_w_from_actkey[i] = o[i];
}
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
}  // namespace

struct svs_frontend {
  svs_ctx *ctx = nullptr;
  svs_frontend_params prm{};
  svs_cam cams[3]{};
  int w[3]{}, h[3]{}, stride[3]{};
  int max_points = 0, max_keyframes = 0, n_points = 0, n_new_records = 0;
  bool have_prev = false;
  // device: two pyramids (current / previous, swapped every frame), right image, disparity, reference clouds (quarter grid)
  uint8_t *d_pyr[2][3] = {};
  int cur = 0;
  uint8_t *d_right = nullptr;
  float *d_disp = nullptr;
  float *d_cloud[3] = {};
  uint8_t *d_kf_pyr = nullptr;          // [max_keyframes] x 3 levels, packed
  size_t kf_level_off[3] = {}, kf_bytes = 0;
  svs_keyframe *d_kfs = nullptr;
  std::vector<svs_keyframe> h_kfs;
  svs_candidate_point *d_pts = nullptr;
  svs_match_result *d_res = nullptr;
  svs_gated_point *d_gated = nullptr;
  // small device block: T (12) | T_actkey_from_w (12) | T_cur_from_w (12) | T_w_from_actkey (12) | pose stats | point stats | passes
  double *d_small = nullptr;
  svs_pose_opt_stats *d_pstats = nullptr;
  svs_point_stats *d_ptstats = nullptr;
  int32_t *d_passes = nullptr;
  svs_fast *fast = nullptr;
  svs_stereo *stereo = nullptr;
  // pinned host staging: images in, results out
  uint8_t *h_in = nullptr; size_t h_in_bytes = 0;
  uint8_t *h_out = nullptr; size_t h_out_bytes = 0;
};

static void fastgrid_for_level(int w, int h, int level, svs_fastgrid *g) {      // stereo_frontend.cpp:73-88 + fast_grid.cpp:23-58
  const int dim = std::max(3 - (int)(level * 0.5), 1);
  const double inv_fac = 1.0 / (1 << level);
  const int total = (int)(2000 * inv_fac * inv_fac), per_cell = total / (dim * dim), bound = std::max(per_cell / 3, 10);
  g->gx = g->gy = dim;
  g->min_inner = (int)(per_cell - bound * 0.33); g->min_outer = per_cell - bound;
  g->max_inner = (int)(per_cell + bound * 0.33); g->max_outer = per_cell + bound;
  g->cell_w = w / dim; g->cell_h = h / dim;
  g->fast_min = 10; g->fast_max = 40;
  for (int i = 0; i < SVS_MAX_CELLS; ++i) g->thr[i] = 25;
}

extern "C" int svs_frontend_destroy(svs_frontend *fe) {
  if (!fe) return SVS_OK;
  (void)hipStreamSynchronize(fe->ctx->stream);
  if (fe->fast) svs_fast_destroy(fe->fast);
  if (fe->stereo) svs_stereo_destroy(fe->stereo);
  for (int k = 0; k < 2; ++k) for (int l = 0; l < 3; ++l) if (fe->d_pyr[k][l]) (void)hipFree(fe->d_pyr[k][l]);
  for (int l = 0; l < 3; ++l) if (fe->d_cloud[l]) (void)hipFree(fe->d_cloud[l]);
  void *ptrs[] = {fe->d_right, fe->d_disp, fe->d_kf_pyr, fe->d_kfs, fe->d_pts, fe->d_res, fe->d_gated, fe->d_small};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  if (fe->h_in) (void)hipHostFree(fe->h_in);
  if (fe->h_out) (void)hipHostFree(fe->h_out);
  delete fe;
  return SVS_OK;
}

extern "C" int svs_frontend_create(svs_ctx *ctx, const svs_cam *cam, const svs_frontend_params *prm, int max_points, int max_keyframes,
                                   svs_frontend **out) {
  SVS_REQUIRE(ctx, ctx && cam && prm && out && max_points >= 1 && max_keyframes >= 1);
  SVS_REQUIRE(ctx, cam->w % 16 == 0 && cam->h % 16 == 0);              // quarter grid on three levels (dense_tracking.cpp:45-46)
  SVS_DEVICE(ctx);
  svs_frontend *fe = new svs_frontend();
  fe->ctx = ctx; fe->prm = *prm; fe->max_points = max_points; fe->max_keyframes = max_keyframes;
  int rc = SVS_OK;
  auto fail = [&](int code) { svs_frontend_destroy(fe); return code; };
  size_t off = 0;
  for (int l = 0; l < 3; ++l) {
    const double s = (double)(1 << l);
    fe->cams[l] = svs_cam{cam->f / s, cam->cx / s, cam->cy / s, cam->b * (1 << l), (int)(cam->w / s), (int)(cam->h / s)};      // frame_grabber-impl.cpp:48-60
    fe->w[l] = fe->cams[l].w; fe->h[l] = fe->cams[l].h; fe->stride[l] = round_up(fe->w[l], 64);
    fe->kf_level_off[l] = off;
    off += (size_t)fe->h[l] * fe->stride[l];
  }
  fe->kf_bytes = off;
  for (int k = 0; k < 2; ++k)
    for (int l = 0; l < 3; ++l) if (hipMalloc(&fe->d_pyr[k][l], (size_t)fe->h[l] * fe->stride[l]) != hipSuccess) return fail(SVS_ERR_HIP);
  const size_t px0 = (size_t)fe->h[0] * fe->stride[0];
  if (hipMalloc(&fe->d_right, px0) != hipSuccess || hipMalloc(&fe->d_disp, px0 * sizeof(float)) != hipSuccess) return fail(SVS_ERR_HIP);
  for (int l = 0; l < 3; ++l) if (hipMalloc(&fe->d_cloud[l], sizeof(float) * 4 * (size_t)(fe->w[l] / 4) * (fe->h[l] / 4)) != hipSuccess) return fail(SVS_ERR_HIP);
  if (hipMalloc(&fe->d_kf_pyr, fe->kf_bytes * max_keyframes) != hipSuccess || hipMalloc(&fe->d_kfs, sizeof(svs_keyframe) * max_keyframes) != hipSuccess ||
      hipMalloc(&fe->d_pts, sizeof(svs_candidate_point) * max_points) != hipSuccess || hipMalloc(&fe->d_res, sizeof(svs_match_result) * max_points) != hipSuccess ||
      hipMalloc(&fe->d_gated, sizeof(svs_gated_point) * max_points) != hipSuccess)
    return fail(SVS_ERR_HIP);
  const size_t small_bytes = sizeof(double) * 48 + sizeof(svs_pose_opt_stats) + sizeof(svs_point_stats) + 64;
  if (hipMalloc(&fe->d_small, small_bytes) != hipSuccess) return fail(SVS_ERR_HIP);
  fe->d_pstats = reinterpret_cast<svs_pose_opt_stats *>(fe->d_small + 48);
  fe->d_ptstats = reinterpret_cast<svs_point_stats *>(reinterpret_cast<char *>(fe->d_pstats) + sizeof(svs_pose_opt_stats));
  fe->d_passes = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(fe->d_ptstats) + sizeof(svs_point_stats));
  fe->h_kfs.assign(max_keyframes, svs_keyframe{});
  svs_fastgrid grids[3];
  for (int l = 0; l < 3; ++l) fastgrid_for_level(fe->w[l], fe->h[l], l, &grids[l]);
  rc = svs_fast_create(ctx, 3, fe->w, fe->h, grids, 1, 8192, &fe->fast);
  if (rc) return fail(rc);
  if (prm->use_block_matching) { rc = svs_stereo_create(ctx, fe->w[0], fe->h[0], 1, &prm->stereo, &fe->stereo); if (rc) return fail(rc); }
  fe->h_in_bytes = 2 * (size_t)fe->w[0] * fe->h[0] + sizeof(float) * (size_t)fe->w[0] * fe->h[0] + sizeof(double) * 24;
  fe->h_out_bytes = small_bytes + (sizeof(svs_match_result) + sizeof(svs_gated_point)) * (size_t)max_points;
  if (hipHostMalloc((void **)&fe->h_in, fe->h_in_bytes, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void **)&fe->h_out, fe->h_out_bytes, hipHostMallocDefault) != hipSuccess)
    return fail(SVS_ERR_HIP);
  *out = fe;
  return SVS_OK;
}

extern "C" int svs_frontend_set_candidates(svs_frontend *fe, const svs_candidate_point *h_pts, int n, int n_new_records) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && n >= 0 && n <= fe->max_points && (n == 0 || h_pts) && n_new_records >= 0 && n_new_records <= n);
  SVS_DEVICE(ctx);
  for (int i = 0; i < n; ++i) SVS_REQUIRE(ctx, h_pts[i].kf_index >= 0 && h_pts[i].kf_index < fe->max_keyframes);
  if (n) SVS_HIP(ctx, hipMemcpyAsync(fe->d_pts, h_pts, sizeof(svs_candidate_point) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));      // h_pts may be pageable and reused by the caller
  fe->n_points = n; fe->n_new_records = n_new_records;
  return SVS_OK;
}

extern "C" int svs_frontend_keep_keyframe(svs_frontend *fe, int slot, const double *T_kf_from_w) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && T_kf_from_w && slot >= 0 && slot < fe->max_keyframes && fe->have_prev);
  SVS_DEVICE(ctx);
  // Frame::clone of the frame processed last (it sits in the "previous" slot after the swap at the end of process_frame)
  const int src = 1 - fe->cur;
  uint8_t *base = fe->d_kf_pyr + fe->kf_bytes * (size_t)slot;
  svs_keyframe &k = fe->h_kfs[slot];
  for (int l = 0; l < 3; ++l) {
    SVS_HIP(ctx, hipMemcpyAsync(base + fe->kf_level_off[l], fe->d_pyr[src][l], (size_t)fe->h[l] * fe->stride[l], hipMemcpyDeviceToDevice, ctx->stream));
    k.pyr[l] = base + fe->kf_level_off[l]; k.stride[l] = fe->stride[l];
  }
  for (int i = 0; i < 12; ++i) k.T_anchor_from_w[i] = T_kf_from_w[i];
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_kfs + slot, &k, sizeof k, hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVS_OK;
}

// upload + pyramid (+ block matching) of a new frame into the "current" slot
static int frontend_ingest(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride, const float *h_disp, int dstride) {
  svs_ctx *ctx = fe->ctx;
  const int w = fe->w[0], h = fe->h[0];
  SVS_REQUIRE(ctx, h_left && lstride >= w);
  SVS_REQUIRE(ctx, fe->prm.use_block_matching ? (h_right && rstride >= w) : (h_disp && dstride >= w));
  uint8_t *in_left = fe->h_in, *in_right = fe->h_in + (size_t)w * h;
  float *in_disp = reinterpret_cast<float *>(fe->h_in + 2 * (size_t)w * h);
  for (int y = 0; y < h; ++y) __builtin_memcpy(in_left + (size_t)y * w, h_left + (size_t)y * lstride, w);
  SVS_HIP(ctx, hipMemcpy2DAsync(fe->d_pyr[fe->cur][0], fe->stride[0], in_left, w, w, h, hipMemcpyHostToDevice, ctx->stream));
  if (fe->prm.use_block_matching) {
    for (int y = 0; y < h; ++y) __builtin_memcpy(in_right + (size_t)y * w, h_right + (size_t)y * rstride, w);
    SVS_HIP(ctx, hipMemcpy2DAsync(fe->d_right, fe->stride[0], in_right, w, w, h, hipMemcpyHostToDevice, ctx->stream));
  } else {
    for (int y = 0; y < h; ++y) __builtin_memcpy(in_disp + (size_t)y * w, h_disp + (size_t)y * dstride, sizeof(float) * w);
    SVS_HIP(ctx, hipMemcpy2DAsync(fe->d_disp, sizeof(float) * fe->stride[0], in_disp, sizeof(float) * w, sizeof(float) * w, h, hipMemcpyHostToDevice, ctx->stream));
  }
  for (int l = 1; l < 3; ++l) {
    const int rc = svs_pyr_down_u8(ctx, fe->d_pyr[fe->cur][l - 1], fe->w[l - 1], fe->h[l - 1], fe->stride[l - 1], 0, fe->d_pyr[fe->cur][l], fe->stride[l], 0, 1);
    if (rc) return rc;
  }
  return SVS_OK;
}
static int frontend_disparity(svs_frontend *fe) {
  if (!fe->prm.use_block_matching) return SVS_OK;
  return svs_stereo_compute(fe->stereo, fe->d_pyr[fe->cur][0], fe->stride[0], 0, fe->d_right, fe->stride[0], 0, fe->d_disp, fe->stride[0], 0, 1);
}
static int frontend_cloud(svs_frontend *fe, const double *d_T) {
  for (int l = 0; l < 3; ++l) {
    const int rc = svs_pointcloud_cpu_sem(fe->ctx, fe->d_disp, fe->stride[0], 0, &fe->cams[l], l, d_T, fe->d_cloud[l], 0, 1);
    if (rc) return rc;
  }
  return SVS_OK;
}

// StereoFrontend::processFirstFrame (stereo_frontend.cpp:110-131): disparity, FAST with 5 trials, reference cloud at the identity
extern "C" int svs_frontend_first_frame(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride,
                                        const float *h_disp, int dstride) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe);
  SVS_DEVICE(ctx);
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));      // the staging area is free again
  int rc = frontend_ingest(fe, h_left, lstride, h_right, rstride, h_disp, dstride);
  if (rc) return rc;
  if ((rc = frontend_disparity(fe))) return rc;
  const uint8_t *imgs[3] = {fe->d_pyr[fe->cur][0], fe->d_pyr[fe->cur][1], fe->d_pyr[fe->cur][2]};
  const size_t bs[3] = {0, 0, 0};
  if ((rc = svs_fast_detect(fe->fast, imgs, fe->stride, bs, 1, fe->prm.fast_trials > 1 ? fe->prm.fast_trials - 1 : 5))) return rc;
  const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  double *in_T = reinterpret_cast<double *>(fe->h_in + 2 * (size_t)fe->w[0] * fe->h[0] + sizeof(float) * (size_t)fe->w[0] * fe->h[0]);
  for (int i = 0; i < 12; ++i) in_T[i] = I[i];
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_small, in_T, sizeof(double) * 12, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = frontend_cloud(fe, fe->d_small))) return rc;
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  fe->cur = 1 - fe->cur;
  fe->have_prev = true;
  return SVS_OK;
}

extern "C" int svs_frontend_process_frame(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride,
                                          const float *h_disp, int dstride, const double *T_cur_from_actkey, const double *T_actkey_from_w,
                                          svs_frame_result *out, svs_match_result *h_matches, svs_gated_point *h_gated) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && T_cur_from_actkey && T_actkey_from_w && out && fe->have_prev);
  SVS_DEVICE(ctx);
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));      // the staging areas are free again
  const int cur = fe->cur, prev = 1 - fe->cur, n = fe->n_points;
  int rc = frontend_ingest(fe, h_left, lstride, h_right, rstride, h_disp, dstride);        // "preprocess"
  if (rc) return rc;
  double *in_T = reinterpret_cast<double *>(fe->h_in + 2 * (size_t)fe->w[0] * fe->h[0] + sizeof(float) * (size_t)fe->w[0] * fe->h[0]);
  for (int i = 0; i < 12; ++i) { in_T[i] = T_cur_from_actkey[i]; in_T[12 + i] = T_actkey_from_w[i]; }
  double *d_T = fe->d_small, *d_Ta = fe->d_small + 12, *d_Tcw = fe->d_small + 24, *d_Twa = fe->d_small + 36;
  SVS_HIP(ctx, hipMemcpyAsync(d_T, in_T, sizeof(double) * 24, hipMemcpyHostToDevice, ctx->stream));
  // "dense tracking": previous frame's cloud + pyramid against the current u8 pyramid
  svs_dense_track_args ta{};
  for (int l = 0; l < 3; ++l) {
    ta.d_cloud[l] = fe->d_cloud[l]; ta.d_prev_u8[l] = fe->d_pyr[prev][l]; ta.pstride[l] = fe->stride[l];
    ta.d_cur_u8[l] = fe->d_pyr[cur][l]; ta.c8stride[l] = fe->stride[l]; ta.cam_vec[l] = fe->cams[l];
  }
  if ((rc = svs_dense_track_cpu_sem(ctx, &ta, d_T, fe->d_passes, 1))) return rc;
  if ((rc = frontend_disparity(fe))) return rc;                                             // "stereo"
  const uint8_t *imgs[3] = {fe->d_pyr[cur][0], fe->d_pyr[cur][1], fe->d_pyr[cur][2]};
  const size_t bs[3] = {0, 0, 0};
  if ((rc = svs_fast_detect(fe->fast, imgs, fe->stride, bs, 1, fe->prm.fast_trials))) return rc;      // "fast"
  if (n > 0) {                                                                              // "match" + calcFastMotionOnly + "process points"
    hipLaunchKernelGGL(frontend_pose_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double *)d_T, (const double *)d_Ta, d_Tcw, d_Twa);
    SVS_LAUNCH_CHECK(ctx);
    svs_match_args ma{};
    ma.d_kfs = fe->d_kfs; ma.n_kf = fe->max_keyframes; ma.d_pts = fe->d_pts; ma.n_pts = n;
    ma.d_T_cur_from_w = d_Tcw; ma.d_T_w_from_actkey = d_Twa;
    for (int l = 0; l < 3; ++l) { ma.d_cur_pyr[l] = fe->d_pyr[cur][l]; ma.cur_stride[l] = fe->stride[l]; ma.cam_vec[l] = fe->cams[l]; }
    ma.d_disp = fe->d_disp; ma.disp_stride = fe->stride[0];
    ma.search_radius = fe->prm.search_radius; ma.thr_mean = fe->prm.thr_mean; ma.thr_std = fe->prm.thr_std; ma.n_batch = 1;
    if ((rc = svs_match(ctx, &ma, fe->fast, fe->d_res))) return rc;
    if ((rc = svs_motion_only(ctx, fe->d_res, n, 0, &fe->cams[0], &fe->prm.pose_opt, d_T, fe->d_pstats, 1))) return rc;
    if ((rc = svs_process_matched_points(ctx, fe->d_res, fe->d_pts, n, 0, 0, fe->n_new_records, &fe->cams[0], d_T, fe->prm.max_reproj_error,
                                         fe->d_gated, 0, fe->d_ptstats, 1)))
      return rc;
  } else {
    SVS_HIP(ctx, hipMemsetAsync(fe->d_pstats, 0, sizeof(svs_pose_opt_stats) + sizeof(svs_point_stats), ctx->stream));
  }
  if ((rc = frontend_cloud(fe, d_T))) return rc;                                            // "dense point cloud" (reference for the next frame)
  // one download: small block, then the records
  const size_t small_bytes = sizeof(double) * 48 + sizeof(svs_pose_opt_stats) + sizeof(svs_point_stats) + 64;
  SVS_HIP(ctx, hipMemcpyAsync(fe->h_out, fe->d_small, small_bytes, hipMemcpyDeviceToHost, ctx->stream));
  svs_match_result *o_res = reinterpret_cast<svs_match_result *>(fe->h_out + small_bytes);
  svs_gated_point *o_gated = reinterpret_cast<svs_gated_point *>(fe->h_out + small_bytes + sizeof(svs_match_result) * (size_t)fe->max_points);
  if (n > 0 && h_matches) SVS_HIP(ctx, hipMemcpyAsync(o_res, fe->d_res, sizeof(svs_match_result) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  if (n > 0 && h_gated) SVS_HIP(ctx, hipMemcpyAsync(o_gated, fe->d_gated, sizeof(svs_gated_point) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const double *o_small = reinterpret_cast<const double *>(fe->h_out);
  for (int i = 0; i < 12; ++i) out->T_cur_from_actkey[i] = o_small[i];
  __builtin_memcpy(&out->pose_stats, fe->h_out + sizeof(double) * 48, sizeof(svs_pose_opt_stats));
  __builtin_memcpy(&out->point_stats, fe->h_out + sizeof(double) * 48 + sizeof(svs_pose_opt_stats), sizeof(svs_point_stats));
  __builtin_memcpy(&out->dense_passes, fe->h_out + sizeof(double) * 48 + sizeof(svs_pose_opt_stats) + sizeof(svs_point_stats), sizeof(int32_t));
  out->n_points = n;
  out->n_matched = n > 0 ? out->pose_stats.num_obs : 0;
  out->tracking_ok = out->n_matched >= 20 ? 1 : 0;                                           // matchAndTrack's minimum (stereo_frontend.cpp:1053-1056)
  if (n > 0 && h_matches) __builtin_memcpy(h_matches, o_res, sizeof(svs_match_result) * (size_t)n);
  if (n > 0 && h_gated) __builtin_memcpy(h_gated, o_gated, sizeof(svs_gated_point) * (size_t)n);
  fe->cur = 1 - fe->cur;                                                                      // this frame is the previous one from now on
  return SVS_OK;
}

/* computeDensePointCloudCpu again, at a pose the caller decided on after the frame (keyframe switch, stereo_frontend.cpp:277-281) */
extern "C" int svs_frontend_recompute_cloud(svs_frontend *fe, const double *T_cur_from_actkey) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && T_cur_from_actkey && fe->have_prev);
  SVS_DEVICE(ctx);
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double *in_T = reinterpret_cast<double *>(fe->h_in + 2 * (size_t)fe->w[0] * fe->h[0] + sizeof(float) * (size_t)fe->w[0] * fe->h[0]);
  for (int i = 0; i < 12; ++i) in_T[i] = T_cur_from_actkey[i];
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_small, in_T, sizeof(double) * 12, hipMemcpyHostToDevice, ctx->stream));
  const int rc = frontend_cloud(fe, fe->d_small);
  if (rc) return rc;
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVS_OK;
}

/* device views for tests / chaining: level images of the frame processed last, its disparity, the reference clouds */
extern "C" int svs_frontend_device_view(svs_frontend *fe, const uint8_t **d_pyr_last, int32_t *stride, const float **d_disp, const float **d_cloud,
                                        svs_fast **fast) {
  if (!fe) return SVS_ERR_INVALID;
  for (int l = 0; l < 3; ++l) {
    if (d_pyr_last) d_pyr_last[l] = fe->d_pyr[1 - fe->cur][l];
    if (stride) stride[l] = fe->stride[l];
    if (d_cloud) d_cloud[l] = fe->d_cloud[l];
  }
  if (d_disp) *d_disp = fe->d_disp;
  if (fast) *fast = fe->fast;
  return SVS_OK;
}
