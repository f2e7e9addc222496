// frontend.hip -- one call per camera frame: the data-parallel part of StereoFrontend::processFrame
// (stereo_frontend.cpp:183-306), for ONE stream with host buffers in and out (the way stereo_slam's main loop calls it,
// stereo_slam.cpp:705) or for a BATCH of independent camera streams whose frames are already in device memory.  Chains the
// kernels of image.hip / dense.hip / dense_full.hip / stereo.hip / fast.hip / match.hip on the context's stream without a host
// round trip in between:
//   upload (pinned staging, optionally prefetched on a copy stream while the frame before is processed)
//   -> FrameGrabber::preprocessing (u8 pyramid; CUDA build: + f32 pyramid and derivatives)       frame_grabber.cpp:285-336
//   -> DenseTracker::denseTrackingCpu / denseTrackingGpu (device-resident LM)                    stereo_frontend.cpp:191-197
//   -> calcDisparityCpu (cv::StereoBM) unless a disparity image is given                         :199-225
//   -> computeFastCorners (FastGrid::detectAdaptively, 6 trials, use_n_levels_in_frontent levels) :228-233
//   -> matchAndTrack: GuidedMatcher::match on the active keyframe's new points, the neighbours' new points (while
//      2 * observations < ui.num_max_points), the neighbourhood's points; calcFastMotionOnly      :235-241, :976-1069
//   -> processMatchedPoints (reprojection gate + PointStatistics)                                :245-262, :834-974
//   -> computeDensePointCloudCpu / Gpu at the refined pose                                       :298-302
//   -> one download of {pose, pass count, match records, gate records, statistics}.
// What stays with the caller is the reference's bookkeeping: keyframe switching / dropping (:265-296), list building, new
// point seeding.  The object owns the device copies of the previous frame (pyramid + reference cloud), the keyframes it was
// told to keep (Frame::clone, keyframes.h:72-83), the candidate points (ap_map) and the FAST threshold state -- per stream.
#include "common.h"
#include <string.h>
#include <algorithm>
#include <vector>

namespace {
constexpr int MAX_GROUPS = 64;

// T_cur_from_w = T_cur_from_actkey * T_actkey_from_w and T_w_from_actkey = T_actkey_from_w^-1 (matcher.cpp:326-330), formed on the
// device from the tracked pose so that the matcher needs no host round trip.  Operation order of the oracle / the host mirror:
// (a0 b0 + a1 b1) + a2 b2, then + t; this file is built without contraction.  One workgroup per stream.
__global__ void frontend_pose_kernel(const double *__restrict__ T_cur_from_actkey, const double *__restrict__ T_actkey_from_w,
                                     double *__restrict__ T_cur_from_w, double *__restrict__ T_w_from_actkey) {
  if (threadIdx.x != 0) return;
  const size_t s = (size_t)blockIdx.x * 12;
  const double *A = T_cur_from_actkey + s, *Bm = T_actkey_from_w + s;
  double *Tcw = T_cur_from_w + s, *Twa = T_w_from_actkey + s;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) Tcw[4 * i + j] = A[4 * i] * Bm[j] + A[4 * i + 1] * Bm[4 + j] + A[4 * i + 2] * Bm[8 + j];
    Tcw[4 * i + 3] += A[4 * i + 3];
  }
  double o[12];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[4 * i + j] = Bm[4 * j + i];
  for (int i = 0; i < 3; ++i) o[4 * i + 3] = -(o[4 * i] * Bm[3] + o[4 * i + 1] * Bm[7] + o[4 * i + 2] * Bm[11]);
  for (int i = 0; i < 12; ++i) Twa[i] = o[i];
}

// matchAndTrack's neighbour cut (stereo_frontend.cpp:1000-1024): the new-point list of neighbour j is matched only while
// 2 * obs_list.size() < ui.num_max_points.  All groups were matched by one launch; this kernel counts the observations group by
// group (list order = record order) and marks the records of the neighbour lists the reference would not have visited.
// group_end[g] = one past the last record of group g; group 0 = the active keyframe's new points, groups 1 .. G-2 = the
// neighbours' new points in strength order, group G-1 = the neighbourhood's points.  One workgroup per stream.
__global__ __launch_bounds__(256) void frontend_group_cut_kernel(svs_match_result *__restrict__ res, size_t res_b, const int32_t *__restrict__ group_end,
                                                                 const int32_t *__restrict__ n_groups, int num_max_points) {
  __shared__ int s_end[MAX_GROUPS], s_cnt[MAX_GROUPS], s_cut[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int G = n_groups[b];
  if (G <= 2) return;                          // no neighbour lists
  res += (size_t)b * res_b;
  if (tid < MAX_GROUPS) { s_end[tid] = tid < G ? group_end[(size_t)b * MAX_GROUPS + tid] : 0; s_cnt[tid] = 0; }
  __syncthreads();
  const int n_new = s_end[G - 2];
  for (int i = tid; i < n_new; i += 256) {
    if (res[i].status != SVS_MATCH_OK) continue;
    int g = 0;
    while (i >= s_end[g]) ++g;
    atomicAdd(&s_cnt[g], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int obs = s_cnt[0], cut = G - 1;
    for (int j = 1; j <= G - 2; ++j) {
      if (!(2 * obs < num_max_points)) { cut = j; break; }
      obs += s_cnt[j];
    }
    s_cut[0] = cut == G - 1 ? n_new : s_end[cut - 1];
    s_cut[1] = n_new;
  }
  __syncthreads();
  for (int i = s_cut[0] + tid; i < s_cut[1]; i += 256) {
    svs_match_result r{};
    r.status = SVS_MATCH_SKIPPED;
    res[i] = r;
  }
}

// rows of `wbytes` bytes (a multiple of 4), dword copies; grid (x blocks, rows, batch)
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint8_t *__restrict__ src, size_t sstride, size_t s_b, uint8_t *__restrict__ dst, size_t dstride,
                                                        size_t d_b, int wdwords) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= wdwords) return;
  const uint32_t *s = reinterpret_cast<const uint32_t *>(src + blockIdx.z * s_b + blockIdx.y * sstride);
  uint32_t *d = reinterpret_cast<uint32_t *>(dst + blockIdx.z * d_b + blockIdx.y * dstride);
  d[x] = s[x];
}
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
}  // namespace

struct svs_frontend {
  svs_ctx *ctx = nullptr;
  svs_frontend_params prm{};
  svs_cam cams[3]{};
  int w[3]{}, h[3]{}, stride[3]{};
  size_t lvl_elems[3]{};                // h * stride: elements between the level images of consecutive streams
  int B = 1, max_points = 0, max_keyframes = 0, n_launch = 0, max_groups_used = 2;
  std::vector<int> n_points, n_new_records;
  std::vector<uint8_t> kept;            // [B][max_keyframes]: slot filled by svs_frontend_keep_keyframe
  bool have_prev = false;
  // device: three pyramid slots per stream (previous / current / next: the next frame may be uploaded while the current one is
  // processed) with their right images / disparities, reference clouds
  uint8_t *d_pyr[3][3] = {};
  int i_prev = 2, i_cur = 0, i_next = 1;
  uint8_t *d_right[3] = {};            // right image / disparity of the frame in pyramid slot k
  float *d_disp[3] = {};
  const float *last_disp = nullptr; int last_dstride = 0; size_t last_dbstride = 0;      // disparity the frame processed last was given / produced
  const uint8_t *ext_left = nullptr; int ext_lstride = 0; size_t ext_lbstride = 0;       // a caller's device frame waiting to be taken in by the first pyramid step
  float *d_cloud[3] = {};               // quarter grid (CPU build) or full resolution (CUDA build), float4 per sample
  size_t cloud_elems[3]{};              // floats per stream and level
  // CUDA build (prm.cuda_build): f32 pyramids of the current / previous frame, derivative images of the current one
  float *d_f32[2][3] = {}, *d_dx[3] = {}, *d_dy[3] = {};
  int i_f32 = 0;
  uint8_t *d_kf_pyr = nullptr;          // [B][max_keyframes] x 3 levels, packed
  size_t kf_level_off[3] = {}, kf_bytes = 0;
  svs_keyframe *d_kfs = nullptr;        // [B][max_keyframes]
  std::vector<svs_keyframe> h_kfs;
  svs_candidate_point *d_pts = nullptr; // [B][max_points]
  svs_candidate_point *h_cand_stage = nullptr;      // pinned, [B][max_points]: svs_frontend_set_candidates_all's staging block (allocated on first use)
  svs_match_result *d_res = nullptr;
  svs_gated_point *d_gated = nullptr;
  int32_t *d_group_end = nullptr, *d_n_groups = nullptr, *d_n_new = nullptr;      // [B][MAX_GROUPS], [B], [B]: records of the new-feature lists
  // small device block: T [B][12] | T_actkey_from_w [B][12] | T_cur_from_w [B][12] | T_w_from_actkey [B][12] | pose stats [B] | point stats [B] | passes [B]
  double *d_small = nullptr;
  size_t small_bytes = 0;
  svs_pose_opt_stats *d_pstats = nullptr;
  svs_point_stats *d_ptstats = nullptr;
  int32_t *d_passes = nullptr;
  svs_fast *fast = nullptr;
  // cross-frame pipeline (ctx option "fe_pipeline"): the pyramid of frame N+1 is built on the side stream while frame N's pose refinement / gate / cloud run
  hipEvent_t ev_early[2] = {}, ev_late[2] = {};      // by frame parity: pyramid done (side stream) / the whole frame done (context's stream)
  hipEvent_t ev_trk[2] = {};                         // ... / the point of the context's stream right in front of the pose refinement's launch
  unsigned pipe_run = 0;                             // frames issued through the pipelined path since the last frame that was not
  svs_stereo *stereo = nullptr;
  // pinned host staging: two input sets (images of stream 0, poses of all streams), one output set
  uint8_t *h_in[2] = {}; size_t h_in_bytes = 0; int i_stage = 0;
  uint8_t *h_out = nullptr; size_t h_out_bytes = 0;
  uint8_t *d_out_block = nullptr;      // one stream: d_small | d_res | d_gated carved from ONE allocation in h_out's layout -- the frame's results go home in one copy
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_upload[2] = {}, ev_done[2] = {};
  // FAST (and block matching) need the new pyramid only, the dense tracker runs ~18 dependent sweeps per stream with a long tail (streams finish at
  // different times): the detector stages are enqueued on a second stream and meet the chain again in front of the matcher
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool prefetched = false, submitted = false, want_matches = false, want_gated = false;
  int n_submitted = 0;
  // accept / reject record of the dense tracker's LM loop, per stream (svs_frontend_dense_records)
  svs_dense_lm_record *d_rec = nullptr; int32_t *d_nrec = nullptr;
  void *d_trk_work = nullptr;           // state of the tracker's balanced launch (dense.hip: LM work of every stream's last frame -> workgroups per stream); big batches only
  // optional stage timing (svs_frontend_set_timing): events between the stages of the last call
  bool timing = false;
  hipEvent_t ev_stage[SVS_FRONTEND_STAGES + 1] = {};
};
constexpr int REC_CAP = 64;

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
  if (fe->copy_stream) (void)hipStreamSynchronize(fe->copy_stream);
  if (fe->side_stream) (void)hipStreamSynchronize(fe->side_stream);
  if (fe->fast) svs_fast_destroy(fe->fast);
  for (hipEvent_t e : fe->ev_early) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : fe->ev_late) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : fe->ev_trk) if (e) (void)hipEventDestroy(e);
  if (fe->stereo) svs_stereo_destroy(fe->stereo);
  for (int k = 0; k < 3; ++k) for (int l = 0; l < 3; ++l) if (fe->d_pyr[k][l]) (void)hipFree(fe->d_pyr[k][l]);
  for (int k = 0; k < 2; ++k) for (int l = 0; l < 3; ++l) if (fe->d_f32[k][l]) (void)hipFree(fe->d_f32[k][l]);
  for (int l = 0; l < 3; ++l) { if (fe->d_cloud[l]) (void)hipFree(fe->d_cloud[l]); if (fe->d_dx[l]) (void)hipFree(fe->d_dx[l]); if (fe->d_dy[l]) (void)hipFree(fe->d_dy[l]); }
  if (fe->d_out_block) { (void)hipFree(fe->d_out_block); fe->d_res = nullptr; fe->d_gated = nullptr; fe->d_small = nullptr; }
  void *ptrs[] = {fe->d_right[0], fe->d_right[1], fe->d_right[2], fe->d_disp[0], fe->d_disp[1], fe->d_disp[2], fe->d_n_new, fe->d_kf_pyr, fe->d_kfs, fe->d_pts, fe->d_res, fe->d_gated, fe->d_small,
                  fe->d_group_end, fe->d_n_groups};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  for (int k = 0; k < 2; ++k) {
    if (fe->h_in[k]) (void)hipHostFree(fe->h_in[k]);
    if (fe->ev_upload[k]) (void)hipEventDestroy(fe->ev_upload[k]);
    if (fe->ev_done[k]) (void)hipEventDestroy(fe->ev_done[k]);
  }
  if (fe->h_out) (void)hipHostFree(fe->h_out);
  if (fe->h_cand_stage) (void)hipHostFree(fe->h_cand_stage);
  if (fe->d_rec) (void)hipFree(fe->d_rec);
  if (fe->d_nrec) (void)hipFree(fe->d_nrec);
  if (fe->d_trk_work) (void)hipFree(fe->d_trk_work);
  for (hipEvent_t e : fe->ev_stage) if (e) (void)hipEventDestroy(e);
  if (fe->copy_stream) (void)hipStreamDestroy(fe->copy_stream);
  if (fe->side_stream) (void)hipStreamDestroy(fe->side_stream);
  if (fe->ev_fork) (void)hipEventDestroy(fe->ev_fork);
  if (fe->ev_join) (void)hipEventDestroy(fe->ev_join);
  delete fe;
  return SVS_OK;
}

extern "C" int svs_frontend_create_batch(svs_ctx *ctx, const svs_cam *cam, const svs_frontend_params *prm, int max_points, int max_keyframes,
                                         int n_streams, svs_frontend **out) {
  SVS_REQUIRE(ctx, ctx && cam && prm && out && max_points >= 1 && max_keyframes >= 1 && n_streams >= 1);
  SVS_REQUIRE(ctx, cam->w % 16 == 0 && cam->h % 16 == 0);              // quarter grid on three levels (dense_tracking.cpp:45-46)
  SVS_REQUIRE(ctx, prm->n_levels >= 0 && prm->n_levels <= 3 && prm->num_max_points >= 0 && prm->min_matches >= 0);
  SVS_DEVICE(ctx);
  svs_frontend *fe = new svs_frontend();
  fe->ctx = ctx; fe->prm = *prm; fe->max_points = max_points; fe->max_keyframes = max_keyframes; fe->B = n_streams;
  if (fe->prm.n_levels == 0) fe->prm.n_levels = 3;
  if (fe->prm.num_max_points == 0) fe->prm.num_max_points = 300;      // ui.num_max_points (stereo_frontend.cpp:1000)
  if (fe->prm.min_matches == 0) fe->prm.min_matches = 20;             // :1053
  const size_t B = (size_t)n_streams;
  fe->n_points.assign(B, 0); fe->n_new_records.assign(B, 0);
  fe->kept.assign(B * max_keyframes, 0);
  int rc = SVS_OK;
  auto fail = [&](int code) { svs_frontend_destroy(fe); return code; };
  size_t off = 0;
  for (int l = 0; l < 3; ++l) {
    const double s = (double)(1 << l);
    fe->cams[l] = svs_cam{cam->f / s, cam->cx / s, cam->cy / s, cam->b * (1 << l), (int)(cam->w / s), (int)(cam->h / s)};      // frame_grabber-impl.cpp:48-60
    fe->w[l] = fe->cams[l].w; fe->h[l] = fe->cams[l].h; fe->stride[l] = round_up(fe->w[l], 64);
    fe->lvl_elems[l] = (size_t)fe->h[l] * fe->stride[l];
    fe->kf_level_off[l] = off;
    off += fe->lvl_elems[l];
  }
  fe->kf_bytes = off;
  for (int k = 0; k < 3; ++k)
    for (int l = 0; l < 3; ++l) if (hipMalloc(&fe->d_pyr[k][l], fe->lvl_elems[l] * B) != hipSuccess) return fail(SVS_ERR_HIP);
  const size_t px0 = fe->lvl_elems[0];
  for (int k = 0; k < 3; ++k) {
    if (prm->use_block_matching && hipMalloc(&fe->d_right[k], px0 * B) != hipSuccess) return fail(SVS_ERR_HIP);
    if (hipMalloc(&fe->d_disp[k], px0 * B * sizeof(float)) != hipSuccess) return fail(SVS_ERR_HIP);
  }
  for (int l = 0; l < 3; ++l) {
    fe->cloud_elems[l] = prm->cuda_build ? 4 * (size_t)fe->w[l] * fe->h[l] : 4 * (size_t)(fe->w[l] / 4) * (fe->h[l] / 4);
    if (hipMalloc(&fe->d_cloud[l], sizeof(float) * fe->cloud_elems[l] * B) != hipSuccess) return fail(SVS_ERR_HIP);
    if (prm->cuda_build) {
      const size_t bytes = sizeof(float) * fe->lvl_elems[l] * B;
      if (hipMalloc(&fe->d_f32[0][l], bytes) != hipSuccess || hipMalloc(&fe->d_f32[1][l], bytes) != hipSuccess || hipMalloc(&fe->d_dx[l], bytes) != hipSuccess ||
          hipMalloc(&fe->d_dy[l], bytes) != hipSuccess)
        return fail(SVS_ERR_HIP);
    }
  }
  // per-stream scalars (poses in and out, statistics): a multiple of 256 bytes, so that the record arrays may follow it in one allocation
  fe->small_bytes = ((sizeof(double) * 48 + sizeof(svs_pose_opt_stats) + sizeof(svs_point_stats) + 8) * B + 255) & ~(size_t)255;
  if (B == 1) {      // latency mode: d_small | d_res | d_gated in the layout of the pinned result buffer -- ONE device-to-host copy per frame instead of three (12 us apart each)
    const size_t bytes = fe->small_bytes + (sizeof(svs_match_result) + sizeof(svs_gated_point)) * (size_t)max_points;
    if (hipMalloc(&fe->d_out_block, bytes) != hipSuccess) return fail(SVS_ERR_HIP);
    fe->d_small = reinterpret_cast<double *>(fe->d_out_block);
    fe->d_res = reinterpret_cast<svs_match_result *>(fe->d_out_block + fe->small_bytes);
    fe->d_gated = reinterpret_cast<svs_gated_point *>(fe->d_out_block + fe->small_bytes + sizeof(svs_match_result) * (size_t)max_points);
  } else if (hipMalloc(&fe->d_res, sizeof(svs_match_result) * max_points * B) != hipSuccess || hipMalloc(&fe->d_gated, sizeof(svs_gated_point) * max_points * B) != hipSuccess ||
             hipMalloc(&fe->d_small, fe->small_bytes) != hipSuccess)
    return fail(SVS_ERR_HIP);
  if (hipMalloc(&fe->d_kf_pyr, fe->kf_bytes * max_keyframes * B) != hipSuccess || hipMalloc(&fe->d_kfs, sizeof(svs_keyframe) * max_keyframes * B) != hipSuccess ||
      hipMalloc(&fe->d_pts, sizeof(svs_candidate_point) * max_points * B) != hipSuccess || hipMalloc(&fe->d_group_end, sizeof(int32_t) * MAX_GROUPS * B) != hipSuccess ||
      hipMalloc(&fe->d_n_groups, sizeof(int32_t) * B) != hipSuccess || hipMalloc(&fe->d_n_new, sizeof(int32_t) * B) != hipSuccess)
    return fail(SVS_ERR_HIP);
  // unused keyframe slots hold null pyramids, unused candidate records kf_index = -1 (-> SVS_MATCH_NO_ANCHOR): nothing a kernel could follow
  if (hipMemsetAsync(fe->d_kfs, 0, sizeof(svs_keyframe) * max_keyframes * B, ctx->stream) != hipSuccess ||
      hipMemsetAsync(fe->d_pts, 0xff, sizeof(svs_candidate_point) * max_points * B, ctx->stream) != hipSuccess ||
      hipMemsetAsync(fe->d_n_groups, 0, sizeof(int32_t) * B, ctx->stream) != hipSuccess || hipMemsetAsync(fe->d_n_new, 0, sizeof(int32_t) * B, ctx->stream) != hipSuccess)
    return fail(SVS_ERR_HIP);
  if (hipMemsetAsync(fe->d_small, 0, fe->small_bytes, ctx->stream) != hipSuccess) return fail(SVS_ERR_HIP);
  fe->d_pstats = reinterpret_cast<svs_pose_opt_stats *>(fe->d_small + 48 * B);
  fe->d_ptstats = reinterpret_cast<svs_point_stats *>(fe->d_pstats + B);
  fe->d_passes = reinterpret_cast<int32_t *>(fe->d_ptstats + B);
  fe->h_kfs.assign(max_keyframes * B, svs_keyframe{});
  svs_fastgrid grids[3];
  for (int l = 0; l < 3; ++l) fastgrid_for_level(fe->w[l], fe->h[l], l, &grids[l]);
  rc = svs_fast_create(ctx, fe->prm.n_levels, fe->w, fe->h, grids, n_streams, 8192, &fe->fast);
  if (rc) return fail(rc);
  if (ctx->fe_pipeline && n_streams > 1 && n_streams <= 2 * ctx->n_cu && !prm->use_block_matching && !prm->cuda_build) {
    for (int k = 0; k < 2; ++k)
      if (hipEventCreateWithFlags(&fe->ev_early[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&fe->ev_late[k], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&fe->ev_trk[k], hipEventDisableTiming) != hipSuccess)
        return fail(SVS_ERR_HIP);
  }
  if (prm->use_block_matching) { rc = svs_stereo_create(ctx, fe->w[0], fe->h[0], n_streams, &prm->stereo, &fe->stereo); if (rc) return fail(rc); }
  fe->h_in_bytes = 2 * (size_t)fe->w[0] * fe->h[0] + sizeof(float) * (size_t)fe->w[0] * fe->h[0] + sizeof(double) * 24 * B;
  fe->h_out_bytes = fe->small_bytes + (sizeof(svs_match_result) + sizeof(svs_gated_point)) * (size_t)max_points;
  for (int k = 0; k < 2; ++k) {
    if (hipHostMalloc((void **)&fe->h_in[k], fe->h_in_bytes, hipHostMallocDefault) != hipSuccess) return fail(SVS_ERR_HIP);
    if (hipEventCreateWithFlags(&fe->ev_upload[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&fe->ev_done[k], hipEventDisableTiming) != hipSuccess)
      return fail(SVS_ERR_HIP);
  }
  if (hipHostMalloc((void **)&fe->h_out, fe->h_out_bytes, hipHostMallocDefault) != hipSuccess) return fail(SVS_ERR_HIP);
  if (hipStreamCreateWithFlags(&fe->copy_stream, hipStreamNonBlocking) != hipSuccess) return fail(SVS_ERR_HIP);
  {
    int least = 0, greatest = 0;      // the chain's own stream keeps the first claim on the CUs
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    if (hipStreamCreateWithPriority(&fe->side_stream, hipStreamNonBlocking, least) != hipSuccess) return fail(SVS_ERR_HIP);
    if (hipEventCreateWithFlags(&fe->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&fe->ev_join, hipEventDisableTiming) != hipSuccess)
      return fail(SVS_ERR_HIP);
  }
  if (hipMalloc(&fe->d_rec, sizeof(svs_dense_lm_record) * REC_CAP * B) != hipSuccess || hipMalloc(&fe->d_nrec, sizeof(int32_t) * B) != hipSuccess ||
      hipMemsetAsync(fe->d_nrec, 0, sizeof(int32_t) * B, ctx->stream) != hipSuccess)
    return fail(SVS_ERR_HIP);
  if (B >= 2 * ctx->n_cu && B <= 4096) {
    if (hipMalloc(&fe->d_trk_work, svs_dense_track_balance_bytes(B)) != hipSuccess) return fail(SVS_ERR_HIP);
    if ((rc = svs_dense_track_balance_init(ctx, fe->d_trk_work, B))) return fail(rc);
  }
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(SVS_ERR_HIP);
  *out = fe;
  return SVS_OK;
}

extern "C" int svs_frontend_create(svs_ctx *ctx, const svs_cam *cam, const svs_frontend_params *prm, int max_points, int max_keyframes,
                                   svs_frontend **out) {
  return svs_frontend_create_batch(ctx, cam, prm, max_points, max_keyframes, 1, out);
}

extern "C" int svs_frontend_set_candidates_grouped(svs_frontend *fe, int stream, const svs_candidate_point *h_pts, int n, const int32_t *h_group_end, int n_groups) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && stream >= 0 && stream < fe->B && n >= 0 && n <= fe->max_points && (n == 0 || h_pts));
  SVS_REQUIRE(ctx, !fe->submitted);      // between submit_frame and wait_frame the list of the frame in flight is latched (wait_frame sizes its copies by it)
  SVS_REQUIRE(ctx, h_group_end && n_groups >= 2 && n_groups <= MAX_GROUPS && h_group_end[n_groups - 1] == n);
  for (int g = 0; g < n_groups; ++g) SVS_REQUIRE(ctx, h_group_end[g] >= (g ? h_group_end[g - 1] : 0));
  SVS_DEVICE(ctx);
  // a candidate must name a keyframe slot that svs_frontend_keep_keyframe has filled (kf_index < 0 = "anchor not in the map", matcher.cpp:336-339)
  for (int i = 0; i < n; ++i)
    SVS_REQUIRE(ctx, h_pts[i].kf_index < 0 || (h_pts[i].kf_index < fe->max_keyframes && fe->kept[(size_t)stream * fe->max_keyframes + h_pts[i].kf_index]));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  svs_candidate_point *dst = fe->d_pts + (size_t)stream * fe->max_points;
  if (n) SVS_HIP(ctx, hipMemcpyAsync(dst, h_pts, sizeof(svs_candidate_point) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  if (fe->n_points[stream] > n) SVS_HIP(ctx, hipMemsetAsync(dst + n, 0xff, sizeof(svs_candidate_point) * (size_t)(fe->n_points[stream] - n), ctx->stream));
  int32_t ge[MAX_GROUPS] = {};
  for (int g = 0; g < n_groups; ++g) ge[g] = h_group_end[g];
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_group_end + (size_t)stream * MAX_GROUPS, ge, sizeof ge, hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_n_groups + stream, &n_groups, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_n_new + stream, &h_group_end[n_groups - 2], sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));      // h_pts may be pageable and reused by the caller
  fe->n_points[stream] = n; fe->n_new_records[stream] = h_group_end[n_groups - 2];
  fe->n_launch = *std::max_element(fe->n_points.begin(), fe->n_points.end());
  fe->max_groups_used = std::max(fe->max_groups_used, n_groups);
  return SVS_OK;
}

extern "C" int svs_frontend_set_candidates(svs_frontend *fe, const svs_candidate_point *h_pts, int n, int n_new_records) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && n >= 0 && n_new_records >= 0 && n_new_records <= n);
  const int32_t ge[2] = {n_new_records, n};
  return svs_frontend_set_candidates_grouped(fe, 0, h_pts, n, ge, 2);
}

extern "C" int svs_frontend_keep_keyframe_of(svs_frontend *fe, int stream, int slot, const double *T_kf_from_w) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && T_kf_from_w && stream >= 0 && stream < fe->B && slot >= 0 && slot < fe->max_keyframes && fe->have_prev && !fe->submitted);
  SVS_DEVICE(ctx);
  // Frame::clone of the frame processed last (it sits in the "previous" slot after the rotation at the end of process_frame)
  const size_t idx = (size_t)stream * fe->max_keyframes + slot;
  uint8_t *base = fe->d_kf_pyr + fe->kf_bytes * idx;
  svs_keyframe &k = fe->h_kfs[idx];
  for (int l = 0; l < 3; ++l) {
    SVS_HIP(ctx, hipMemcpyAsync(base + fe->kf_level_off[l], fe->d_pyr[fe->i_prev][l] + fe->lvl_elems[l] * stream, fe->lvl_elems[l], hipMemcpyDeviceToDevice, ctx->stream));
    k.pyr[l] = base + fe->kf_level_off[l]; k.stride[l] = fe->stride[l];
  }
  for (int i = 0; i < 12; ++i) k.T_anchor_from_w[i] = T_kf_from_w[i];
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_kfs + idx, &k, sizeof k, hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  fe->kept[idx] = 1;
  return SVS_OK;
}
extern "C" int svs_frontend_keep_keyframe(svs_frontend *fe, int slot, const double *T_kf_from_w) { return svs_frontend_keep_keyframe_of(fe, 0, slot, T_kf_from_w); }

/* the same for ALL streams in one go: the frame each stream processed last becomes its keyframe `slot`; h_T_kf_from_w [n_streams][12].  One strided copy per pyramid
   level and one table upload instead of 4 small copies and a synchronisation per stream. */
extern "C" int svs_frontend_keep_keyframes(svs_frontend *fe, int slot, const double *h_T_kf_from_w) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && h_T_kf_from_w && slot >= 0 && slot < fe->max_keyframes && fe->have_prev && !fe->submitted);
  SVS_DEVICE(ctx);
  const size_t B = (size_t)fe->B, slot_pitch = fe->kf_bytes * (size_t)fe->max_keyframes;      // distance between the same slot of two consecutive streams
  for (int l = 0; l < 3; ++l)
    SVS_HIP(ctx, hipMemcpy2DAsync(fe->d_kf_pyr + fe->kf_bytes * (size_t)slot + fe->kf_level_off[l], slot_pitch, fe->d_pyr[fe->i_prev][l], fe->lvl_elems[l], fe->lvl_elems[l], B,
                                  hipMemcpyDeviceToDevice, ctx->stream));
  for (size_t b = 0; b < B; ++b) {
    const size_t idx = b * fe->max_keyframes + slot;
    svs_keyframe &k = fe->h_kfs[idx];
    uint8_t *base = fe->d_kf_pyr + fe->kf_bytes * idx;
    for (int l = 0; l < 3; ++l) { k.pyr[l] = base + fe->kf_level_off[l]; k.stride[l] = fe->stride[l]; }
    for (int i = 0; i < 12; ++i) k.T_anchor_from_w[i] = h_T_kf_from_w[12 * b + i];
  }
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_kfs, fe->h_kfs.data(), sizeof(svs_keyframe) * fe->h_kfs.size(), hipMemcpyHostToDevice, ctx->stream));      // (the whole table: h_kfs mirrors it)
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t b = 0; b < B; ++b) fe->kept[b * fe->max_keyframes + slot] = 1;
  return SVS_OK;
}

/* svs_frontend_set_candidates_grouped for ALL streams in one go: h_pts = the streams' records back to back (stream b: h_n[b] records), h_group_end
   [n_streams][n_groups] (every stream with the same number of groups; an absent neighbour list is an empty group).  Staged through one host buffer: three
   uploads per call instead of four per stream. */
extern "C" int svs_frontend_set_candidates_all(svs_frontend *fe, const svs_candidate_point *h_pts, const int32_t *h_n, const int32_t *h_group_end, int n_groups) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && h_n && h_group_end && n_groups >= 2 && n_groups <= MAX_GROUPS && !fe->submitted);
  SVS_DEVICE(ctx);
  const int B = fe->B;
  size_t total = 0;
  for (int b = 0; b < B; ++b) {
    SVS_REQUIRE(ctx, h_n[b] >= 0 && h_n[b] <= fe->max_points && h_group_end[(size_t)b * n_groups + n_groups - 1] == h_n[b]);
    for (int g = 0; g < n_groups; ++g) SVS_REQUIRE(ctx, h_group_end[(size_t)b * n_groups + g] >= (g ? h_group_end[(size_t)b * n_groups + g - 1] : 0));
    total += (size_t)h_n[b];
  }
  SVS_REQUIRE(ctx, total == 0 || h_pts);
  {
    size_t off = 0;
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < h_n[b]; ++i, ++off)
        SVS_REQUIRE(ctx, h_pts[off].kf_index < 0 || (h_pts[off].kf_index < fe->max_keyframes && fe->kept[(size_t)b * fe->max_keyframes + h_pts[off].kf_index]));
  }
  // the device list is [n_streams][max_points]: lay the records out like that in a pinned block the front end keeps (unused tails = 0xff, "no candidate") and send
  // the first m records of every stream as one strided copy, m = the longest list now or before (what lies behind a stream's list on the device is 0xff: svs_frontend_create, the per-stream setters)
  if (!fe->h_cand_stage) SVS_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&fe->h_cand_stage), sizeof(svs_candidate_point) * (size_t)fe->B * fe->max_points, hipHostMallocDefault));
  int m = 0;
  for (int b = 0; b < B; ++b) m = std::max(m, std::max(h_n[b], fe->n_points[b]));
  std::vector<int32_t> ge((size_t)B * MAX_GROUPS, 0), ng((size_t)B, n_groups), nn((size_t)B);
  size_t off = 0;
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));      // (the staging block of the previous call has been read)
  for (int b = 0; b < B; ++b) {
    svs_candidate_point *row = fe->h_cand_stage + (size_t)b * fe->max_points;
    if (h_n[b]) __builtin_memcpy(row, h_pts + off, sizeof(svs_candidate_point) * (size_t)h_n[b]);
    if (m > h_n[b]) __builtin_memset(row + h_n[b], 0xff, sizeof(svs_candidate_point) * (size_t)(m - h_n[b]));
    off += (size_t)h_n[b];
    for (int g = 0; g < n_groups; ++g) ge[(size_t)b * MAX_GROUPS + g] = h_group_end[(size_t)b * n_groups + g];
    nn[b] = h_group_end[(size_t)b * n_groups + n_groups - 2];
  }
  if (m > 0)
    SVS_HIP(ctx, hipMemcpy2DAsync(fe->d_pts, sizeof(svs_candidate_point) * (size_t)fe->max_points, fe->h_cand_stage, sizeof(svs_candidate_point) * (size_t)fe->max_points,
                                  sizeof(svs_candidate_point) * (size_t)m, (size_t)B, hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_group_end, ge.data(), sizeof(int32_t) * ge.size(), hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_n_groups, ng.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_n_new, nn.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int b = 0; b < B; ++b) { fe->n_points[b] = h_n[b]; fe->n_new_records[b] = nn[b]; }
  fe->n_launch = *std::max_element(fe->n_points.begin(), fe->n_points.end());
  fe->max_groups_used = std::max(fe->max_groups_used, n_groups);
  return SVS_OK;
}

extern "C" int svs_frontend_input_view(svs_frontend *fe, uint8_t **d_left, int32_t *lstride, size_t *l_bstride, uint8_t **d_right, int32_t *rstride,
                                       size_t *r_bstride, float **d_disp, int32_t *dstride, size_t *d_bstride) {
  if (!fe) return SVS_ERR_INVALID;
  if (d_left) *d_left = fe->d_pyr[fe->i_cur][0];
  if (lstride) *lstride = fe->stride[0];
  if (l_bstride) *l_bstride = fe->lvl_elems[0];
  if (d_right) *d_right = fe->d_right[fe->i_cur];      // NULL without block matching
  if (rstride) *rstride = fe->stride[0];
  if (r_bstride) *r_bstride = fe->lvl_elems[0];
  if (d_disp) *d_disp = fe->d_disp[fe->i_cur];        // with block matching: where the disparity will be computed
  if (dstride) *dstride = fe->stride[0];
  if (d_bstride) *d_bstride = fe->lvl_elems[0];
  return SVS_OK;
}

/* the pinned host buffers the NEXT frame of stream 0 is staged in (w x h, contiguous): a frame grabber that writes its images straight into them
   (and then passes these pointers with stride = w) saves the host-side copy of 1.5 MB per 640 x 480 frame.  Valid until the next first_frame /
   submit_frame / process_frame / prefetch_frame call, which moves on to the other staging set. */
extern "C" int svs_frontend_staging_view(svs_frontend *fe, uint8_t **h_left, uint8_t **h_right, float **h_disp) {
  if (!fe) return SVS_ERR_INVALID;
  const size_t px = (size_t)fe->w[0] * fe->h[0];
  uint8_t *base = fe->h_in[fe->i_stage];
  if (h_left) *h_left = base;
  if (h_right) *h_right = base + px;
  if (h_disp) *h_disp = reinterpret_cast<float *>(base + 2 * px);
  return SVS_OK;
}

namespace {
// stage the caller's images (stream 0) in pinned memory and enqueue their upload into the "current" slots on `s`
int frontend_upload(svs_frontend *fe, int stage, hipStream_t s, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride, const float *h_disp,
                    int dstride) {
  svs_ctx *ctx = fe->ctx;
  const int w = fe->w[0], h = fe->h[0];
  SVS_REQUIRE(ctx, h_left && lstride >= w);
  SVS_REQUIRE(ctx, fe->prm.use_block_matching ? (h_right && rstride >= w) : (h_disp && dstride >= w));
  uint8_t *in_left = fe->h_in[stage], *in_right = fe->h_in[stage] + (size_t)w * h;
  float *in_disp = reinterpret_cast<float *>(fe->h_in[stage] + 2 * (size_t)w * h);
  // a frame the caller produced IN the staging set (svs_frontend_staging_view) needs no copy; a contiguous image is one memcpy
  auto stage_rows = [](void *dst, const void *src, size_t row_bytes, size_t sstride_bytes, int rows) {
    if (dst == src) return;
    if (sstride_bytes == row_bytes) { __builtin_memcpy(dst, src, row_bytes * (size_t)rows); return; }
    for (int y = 0; y < rows; ++y) __builtin_memcpy(static_cast<char *>(dst) + (size_t)y * row_bytes, static_cast<const char *>(src) + (size_t)y * sstride_bytes, row_bytes);
  };
  stage_rows(in_left, h_left, (size_t)w, (size_t)lstride, h);
  SVS_HIP(ctx, hipMemcpy2DAsync(fe->d_pyr[fe->i_cur][0], fe->stride[0], in_left, w, w, h, hipMemcpyHostToDevice, s));
  if (fe->prm.use_block_matching) {
    stage_rows(in_right, h_right, (size_t)w, (size_t)rstride, h);
    SVS_HIP(ctx, hipMemcpy2DAsync(fe->d_right[fe->i_cur], fe->stride[0], in_right, w, w, h, hipMemcpyHostToDevice, s));
  } else {
    stage_rows(in_disp, h_disp, sizeof(float) * (size_t)w, sizeof(float) * (size_t)dstride, h);
    SVS_HIP(ctx, hipMemcpy2DAsync(fe->d_disp[fe->i_cur], sizeof(float) * fe->stride[0], in_disp, sizeof(float) * w, sizeof(float) * w, h, hipMemcpyHostToDevice, s));
  }
  return SVS_OK;
}
// device-resident frames of all streams -> the "current" slots (skipped for pointers handed out by svs_frontend_input_view)
int frontend_take_device_frames(svs_frontend *fe, const svs_frames_dev *in) {
  svs_ctx *ctx = fe->ctx;
  const int w = fe->w[0], h = fe->h[0];
  SVS_REQUIRE(ctx, in->d_left && in->lstride >= w && (fe->B == 1 || in->l_bstride >= (size_t)h * in->lstride));
  SVS_REQUIRE(ctx, fe->prm.use_block_matching ? (in->d_right && in->rstride >= w) : (in->d_disp && in->dstride >= w));
  SVS_REQUIRE(ctx, in->lstride % 4 == 0 && (!in->d_right || in->rstride % 4 == 0));
  // "frames are complete when this event has fired, recorded on whichever stream": every read of the caller's buffers on the context's stream comes behind it -- the
  // right-image copy below, the first pyramid step of the unpipelined chain and of svs_frontend_first_frames (the pipelined chain waits on its side stream as well)
  if (in->ready_event) SVS_HIP(ctx, hipStreamWaitEvent(ctx->stream, static_cast<hipEvent_t>(in->ready_event), 0));
  const dim3 grid(div_up(w / 4, 256), h, fe->B);
  // the left image is taken into the level-0 buffer by the first pyramid step (frontend_chain): one read of the frame instead of two
  fe->ext_left = in->d_left != fe->d_pyr[fe->i_cur][0] ? in->d_left : nullptr;
  fe->ext_lstride = in->lstride; fe->ext_lbstride = in->l_bstride;
  if (fe->prm.use_block_matching && in->d_right != fe->d_right[fe->i_cur]) {
    hipLaunchKernelGGL(copy_rows_kernel, grid, dim3(256), 0, ctx->stream, in->d_right, (size_t)in->rstride, in->r_bstride, fe->d_right[fe->i_cur], (size_t)fe->stride[0],
                       fe->lvl_elems[0], w / 4);
    SVS_LAUNCH_CHECK(ctx);
  }
  return SVS_OK;
}

struct DispView { const float *p; int stride; size_t bstride; };
// everything behind the arrival of the images: pyramid, (tracking), stereo, FAST, (match, motion-only, gate), cloud.  first: processFirstFrame
#define STAGE_MARK(k) do { if (fe->timing) SVS_HIP(ctx, hipEventRecord(fe->ev_stage[k], ctx->stream)); } while (0)
int frontend_chain(svs_frontend *fe, bool first, DispView dv, bool ext_frames = false, hipEvent_t frames_ready = nullptr) {
  svs_ctx *ctx = fe->ctx;
  const int B = fe->B, cur = fe->i_cur, prev = fe->i_prev, n = fe->n_launch;
  int rc;
  // Cross-frame pipeline.  The pyramid of a frame needs nothing of the frames before but a free slot (three slots: last read by the frame before the previous
  // one).  With the frames in caller-owned device buffers whose completion the caller has tied to an EVENT (svs_frames_dev::ready_event -- without one the frames
  // may still be in the making on the context's stream, and only that stream's order protects the read) it goes to the
  // low-priority side stream and is released when the PREVIOUS frame reaches its pose refinement: a caller that enqueues frame N+1 while frame N is still running
  // gets the 0.2 ms pyramid of N+1 into the tail of N's refinement (one workgroup per stream, <= 15 dependent LM iterations, most streams done early), its gate
  // and its cloud.  Results are those of the one-stream order (tests/test_gpu_frontend_batch.py).
  const bool pipe = !first && ext_frames && frames_ready && fe->ev_early[0] && fe->side_stream && ctx->fe_pipeline && ctx->fe_overlap && !fe->timing && B <= 2 * ctx->n_cu;
  svs_fast *const F = fe->fast;
  const int par = (int)(fe->pipe_run & 1u);
  hipStream_t const chain_stream0 = ctx->stream;
  if (pipe) {
    if (fe->pipe_run == 0) {                                                                  // first pipelined frame: behind everything enqueued so far
      SVS_HIP(ctx, hipEventRecord(fe->ev_fork, chain_stream0));
      SVS_HIP(ctx, hipStreamWaitEvent(fe->side_stream, fe->ev_fork, 0));
    } else {
      if (fe->pipe_run >= 2) SVS_HIP(ctx, hipStreamWaitEvent(fe->side_stream, fe->ev_late[par], 0));      // frame N-2: the last reader of this pyramid slot
      SVS_HIP(ctx, hipStreamWaitEvent(fe->side_stream, fe->ev_trk[1 - par], 0));                          // frame N-1 has reached its pose refinement
    }
    SVS_HIP(ctx, hipStreamWaitEvent(fe->side_stream, frames_ready, 0));                          // the caller's word that the frames are complete
    ctx->stream = fe->side_stream;                                                            // (a context is used by one thread at a time)
  } else fe->pipe_run = 0;
  auto back_to_chain = [&](int rc_) { ctx->stream = chain_stream0; return rc_; };
  for (int l = 1; l < 3; ++l) {                                                               // "preprocess"
    if (l == 1 && fe->ext_left)
      rc = svs_pyr_down_u8_copy(ctx, fe->ext_left, fe->w[0], fe->h[0], fe->ext_lstride, fe->ext_lbstride, fe->d_pyr[cur][1], fe->stride[1], fe->lvl_elems[1],
                                fe->d_pyr[cur][0], fe->stride[0], fe->lvl_elems[0], B);
    else
      rc = svs_pyr_down_u8(ctx, fe->d_pyr[cur][l - 1], fe->w[l - 1], fe->h[l - 1], fe->stride[l - 1], fe->lvl_elems[l - 1], fe->d_pyr[cur][l], fe->stride[l],
                           fe->lvl_elems[l], B);
    if (rc) return back_to_chain(rc);
  }
  fe->ext_left = nullptr;
  if (pipe) {
    const hipError_t e = hipEventRecord(fe->ev_early[par], fe->side_stream);
    ctx->stream = chain_stream0;
    SVS_HIP(ctx, e);
    SVS_HIP(ctx, hipStreamWaitEvent(ctx->stream, fe->ev_early[par], 0));
  }
  const int f32c = fe->i_f32, f32p = 1 - fe->i_f32;
  if (fe->prm.cuda_build) {
    if ((rc = svs_preprocess_gpu_sem(ctx, fe->d_pyr[cur][0], fe->w[0], fe->h[0], fe->stride[0], fe->lvl_elems[0], fe->d_f32[f32c], fe->d_dx, fe->d_dy, fe->stride,
                                     fe->lvl_elems, 3, B)))
      return rc;
  }
  double *d_T = fe->d_small, *d_Ta = fe->d_small + 12 * (size_t)B, *d_Tcw = fe->d_small + 24 * (size_t)B, *d_Twa = fe->d_small + 36 * (size_t)B;
  STAGE_MARK(1);
  // "stereo" + "fast": they read the new pyramid (and the right image) only
  auto detect = [&]() -> int {
    int rc2;
    if (fe->prm.use_block_matching) {                                                         // "stereo"
      if ((rc2 = svs_stereo_compute(fe->stereo, fe->d_pyr[cur][0], fe->stride[0], fe->lvl_elems[0], fe->d_right[fe->i_cur], fe->stride[0], fe->lvl_elems[0], fe->d_disp[cur],
                                    fe->stride[0], fe->lvl_elems[0], B)))
        return rc2;
      dv = DispView{fe->d_disp[cur], fe->stride[0], fe->lvl_elems[0]};
    }
    STAGE_MARK(3);
    const uint8_t *imgs[3] = {fe->d_pyr[cur][0], fe->d_pyr[cur][1], fe->d_pyr[cur][2]};
    const int trials = first ? (fe->prm.fast_trials > 1 ? fe->prm.fast_trials - 1 : 5) : fe->prm.fast_trials;      // stereo_frontend.cpp:118 / :232
    return svs_fast_detect(F, imgs, fe->stride, fe->lvl_elems, B, trials);            // "fast"
  };
  // With the stage clocks off, the detector stages of a tracked frame go to the side stream.  The fork event is recorded in FRONT of the tracker's launch, so
  // nothing but the streams' priorities (side stream: lowest) orders the two: the detector's workgroups fill what the tracker leaves idle -- above all its
  // tail, when most streams have converged.  (With the clocks on, every stage runs
  // alone on the chain's stream so that the stage times add up to the step.)
  // (Only while all of the tracker's workgroups -- one per stream, two per CU -- are resident at once: beyond that the tracker has its own queue of
  // workgroups to fill the tail with, and detector workgroups in between only delay it: 7.01 vs 6.78 ms per step at 1024 streams.)
  const bool side = !first && fe->side_stream && ctx->fe_overlap && !fe->timing && B <= 2 * ctx->n_cu;
  if (side) SVS_HIP(ctx, hipEventRecord(fe->ev_fork, ctx->stream));
  if (!first) {                                                                               // "dense tracking"
    if (fe->prm.cuda_build) {
      svs_dense_track_full_args ta{};
      for (int l = 0; l < 3; ++l) {
        ta.d_cloud4[l] = fe->d_cloud[l]; ta.stride_f4[l] = fe->w[l]; ta.cloud_bstride[l] = fe->cloud_elems[l] / 4;
        ta.d_prev[l] = fe->d_f32[f32p][l]; ta.d_cur[l] = fe->d_f32[f32c][l]; ta.d_dx[l] = fe->d_dx[l]; ta.d_dy[l] = fe->d_dy[l];
        ta.stride_f[l] = fe->stride[l]; ta.f_bstride[l] = fe->lvl_elems[l]; ta.w[l] = fe->w[l]; ta.h[l] = fe->h[l];
        ta.f[l] = fe->cams[l].f; ta.cx[l] = fe->cams[l].cx; ta.cy[l] = fe->cams[l].cy;
      }
      ta.d_record_out = fe->d_rec; ta.record_cap = REC_CAP; ta.d_n_record_out = fe->d_nrec;
      if ((rc = svs_dense_track_full(ctx, &ta, d_T, fe->d_passes, B))) return rc;
    } else {
      svs_dense_track_args ta{};
      for (int l = 0; l < 3; ++l) {
        ta.d_cloud[l] = fe->d_cloud[l]; ta.cloud_bstride[l] = fe->cloud_elems[l];
        ta.d_prev_u8[l] = fe->d_pyr[prev][l]; ta.pstride[l] = fe->stride[l]; ta.p_bstride[l] = fe->lvl_elems[l];
        ta.d_cur_u8[l] = fe->d_pyr[cur][l]; ta.c8stride[l] = fe->stride[l]; ta.c8_bstride[l] = fe->lvl_elems[l]; ta.cam_vec[l] = fe->cams[l];
      }
      ta.d_record_out = fe->d_rec; ta.record_cap = REC_CAP; ta.d_n_record_out = fe->d_nrec;
      if ((rc = svs_dense_track_cpu_sem_work(ctx, &ta, d_T, fe->d_passes, B, fe->d_trk_work))) return rc;
    }
  }
  STAGE_MARK(2);
  if (side) {
    hipStream_t chain_stream = ctx->stream;
    SVS_HIP(ctx, hipStreamWaitEvent(fe->side_stream, fe->ev_fork, 0));
    ctx->stream = fe->side_stream;                                                            // (a context is used by one thread at a time)
    rc = detect();
    const hipError_t e = rc ? hipSuccess : hipEventRecord(fe->ev_join, fe->side_stream);
    ctx->stream = chain_stream;
    if (rc) return rc;
    SVS_HIP(ctx, e);
    SVS_HIP(ctx, hipStreamWaitEvent(ctx->stream, fe->ev_join, 0));
  } else if ((rc = detect())) return rc;
  STAGE_MARK(4);
  bool fused_tail = false;
  if (!first) {
    if (n > 0) {                                                                              // "match" + calcFastMotionOnly + "process points"
      // T_cur_from_w / T_w_from_actkey (matcher.cpp:326-330): with a handful of keyframes per stream the matcher's prediction kernel forms them itself (match.hip:
      // match_predict_kernel<true>, same expressions, same bits) -- between the tracker and the matcher every launch is on the step's critical path
      const bool pose_in_matcher = fe->max_keyframes <= 8;
      if (!pose_in_matcher) {
        hipLaunchKernelGGL(frontend_pose_kernel, dim3(B), dim3(64), 0, ctx->stream, (const double *)d_T, (const double *)d_Ta, d_Tcw, d_Twa);
        SVS_LAUNCH_CHECK(ctx);
      }
      svs_match_args ma{};
      ma.d_kfs = fe->d_kfs; ma.n_kf = fe->max_keyframes; ma.kf_bstride = (size_t)fe->max_keyframes; ma.d_pts = fe->d_pts; ma.n_pts = n;
      ma.pts_bstride = (size_t)fe->max_points; ma.out_bstride = (size_t)fe->max_points;
      ma.d_T_cur_from_w = d_Tcw; ma.d_T_w_from_actkey = d_Twa;
      for (int l = 0; l < 3; ++l) { ma.d_cur_pyr[l] = fe->d_pyr[cur][l]; ma.cur_stride[l] = fe->stride[l]; ma.cur_bstride[l] = fe->lvl_elems[l]; ma.cam_vec[l] = fe->cams[l]; }
      ma.d_disp = dv.p; ma.disp_stride = dv.stride; ma.disp_bstride = dv.bstride;
      ma.search_radius = fe->prm.search_radius; ma.thr_mean = fe->prm.thr_mean; ma.thr_std = fe->prm.thr_std; ma.n_batch = B;
      ctx->match_src_T = pose_in_matcher ? d_T : nullptr; ctx->match_src_Ta = pose_in_matcher ? d_Ta : nullptr;
      rc = svs_match(ctx, &ma, F, fe->d_res);
      ctx->match_src_T = ctx->match_src_Ta = nullptr;
      if (rc) return rc;
      if (fe->max_groups_used > 2) {
        hipLaunchKernelGGL(frontend_group_cut_kernel, dim3(B), dim3(256), 0, ctx->stream, fe->d_res, (size_t)fe->max_points, (const int32_t *)fe->d_group_end,
                           (const int32_t *)fe->d_n_groups, fe->prm.num_max_points);
        SVS_LAUNCH_CHECK(ctx);
      }
      STAGE_MARK(5);
      svs_pose_opt_params po = fe->prm.pose_opt;
      po.min_obs = fe->prm.min_matches;
      if (pipe) SVS_HIP(ctx, hipEventRecord(fe->ev_trk[par], ctx->stream));                     // releases the next frame's pyramid (side stream)
      // refinement, gate and clouds in one launch (not with the stage clocks on: the stages are then launched, and timed, one by one)
      // (and not for a stream or two: their gate and clouds are faster spread over a hundred workgroups than behind one another in the stream's own)
      if (!fe->timing && !fe->prm.cuda_build && ctx->fe_fuse_tail && 2 * B >= ctx->n_cu) {
        svs_mo_tail tl{};
        tl.pts = fe->d_pts; tl.pts_b = (size_t)fe->max_points; tl.n_new = fe->d_n_new; tl.mre = fe->prm.max_reproj_error; tl.gated = fe->d_gated;
        tl.gated_b = (size_t)fe->max_points; tl.ptstats = fe->d_ptstats; tl.disp = dv.p; tl.ds = dv.stride; tl.disp_b = dv.bstride;
        for (int l = 0; l < 3; ++l) { tl.cams[l] = fe->cams[l]; tl.cloud[l] = fe->d_cloud[l]; tl.cloud_b[l] = fe->cloud_elems[l]; }
        tl.order = fe->d_trk_work ? svs_dense_track_balance_order(ctx, fe->d_trk_work, B) : nullptr;
        rc = svs_motion_only_gate_cloud(ctx, fe->d_res, n, (size_t)fe->max_points, &fe->cams[0], &po, d_T, fe->d_pstats, &tl, B);
        if (rc == SVS_OK) fused_tail = true;
        else if (rc != SVS_ERR_UNSUPPORTED) return rc;
      }
      if (!fused_tail) {
        if ((rc = svs_motion_only(ctx, fe->d_res, n, (size_t)fe->max_points, &fe->cams[0], &po, d_T, fe->d_pstats, B))) return rc;
        STAGE_MARK(6);
        if ((rc = svs_process_matched_points_dev(ctx, fe->d_res, fe->d_pts, n, (size_t)fe->max_points, (size_t)fe->max_points, fe->d_n_new, &fe->cams[0], d_T,
                                                 fe->prm.max_reproj_error, fe->d_gated, (size_t)fe->max_points, fe->d_ptstats, B)))
          return rc;
      }
    } else {
      if (pipe) SVS_HIP(ctx, hipEventRecord(fe->ev_trk[par], ctx->stream));
      SVS_HIP(ctx, hipMemsetAsync(fe->d_pstats, 0, (sizeof(svs_pose_opt_stats) + sizeof(svs_point_stats)) * (size_t)B, ctx->stream));
      STAGE_MARK(5); STAGE_MARK(6);
    }
  } else { STAGE_MARK(5); STAGE_MARK(6); }
  STAGE_MARK(7);
  if (!fused_tail && !fe->prm.cuda_build) {                                                   // "dense point cloud" (reference for the next frame): three levels, one launch
    size_t cb[3] = {fe->cloud_elems[0], fe->cloud_elems[1], fe->cloud_elems[2]};
    if ((rc = svs_pointcloud_cpu_sem_levels(ctx, dv.p, dv.stride, dv.bstride, fe->cams, d_T, fe->d_cloud, cb, B))) return rc;
  }
  for (int l = 0; l < 3 && !fused_tail && fe->prm.cuda_build; ++l) {
    if (fe->prm.cuda_build)
      rc = svs_pointcloud_full_pose(ctx, d_T, &fe->cams[l], dv.p, dv.stride, dv.bstride, fe->w[l], fe->h[l], fe->w[l], fe->cloud_elems[l] / 4, 1 << l, fe->d_cloud[l], B);
    else
      rc = svs_pointcloud_cpu_sem(ctx, dv.p, dv.stride, dv.bstride, &fe->cams[l], l, d_T, fe->d_cloud[l], fe->cloud_elems[l], B);
    if (rc) return rc;
  }
  STAGE_MARK(8);
  if (pipe) { SVS_HIP(ctx, hipEventRecord(fe->ev_late[par], ctx->stream)); ++fe->pipe_run; }
  fe->last_disp = dv.p; fe->last_dstride = dv.stride; fe->last_dbstride = dv.bstride;
  return SVS_OK;
}
void frontend_rotate(svs_frontend *fe) {
  const int p = fe->i_prev;
  fe->i_prev = fe->i_cur; fe->i_cur = fe->i_next; fe->i_next = p;       // this frame is the previous one from now on
  fe->i_f32 = 1 - fe->i_f32;
  fe->prefetched = false;
}
double *stage_poses(svs_frontend *fe, int stage) { return reinterpret_cast<double *>(fe->h_in[stage] + 2 * (size_t)fe->w[0] * fe->h[0] + sizeof(float) * (size_t)fe->w[0] * fe->h[0]); }
// make staging set `stage` writable again: the copies that read it last have completed
int stage_acquire(svs_frontend *fe, int stage) {
  SVS_HIP(fe->ctx, hipEventSynchronize(fe->ev_upload[stage]));
  return SVS_OK;
}
}  // namespace

/* upload the NEXT frame of stream 0 on a copy stream while the frame submitted last is still being processed (the reference's FrameData double
   buffer, frame_grabber.hpp:93-155).  The next svs_frontend_submit_frame / process_frame / first_frame must then pass NULL images. */
extern "C" int svs_frontend_prefetch_frame(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride, const float *h_disp, int dstride) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && fe->B == 1 && !fe->prefetched);
  SVS_DEVICE(ctx);
  const int stage = fe->i_stage;
  int rc = stage_acquire(fe, stage);
  if (rc) return rc;
  // the slots written here (pyramid / disparity / right-image slot i_cur) were last READ by the frame before the one in flight
  SVS_HIP(ctx, hipStreamWaitEvent(fe->copy_stream, fe->ev_done[(fe->n_submitted + 1) & 1], 0));
  if ((rc = frontend_upload(fe, stage, fe->copy_stream, h_left, lstride, h_right, rstride, h_disp, dstride))) return rc;
  SVS_HIP(ctx, hipEventRecord(fe->ev_upload[stage], fe->copy_stream));
  fe->prefetched = true;
  return SVS_OK;
}

static int frontend_begin(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride, const float *h_disp, int dstride, int *stage_out) {
  svs_ctx *ctx = fe->ctx;
  const int stage = fe->i_stage;
  int rc;
  if (fe->prefetched) {
    SVS_REQUIRE(ctx, !h_left);                                        // the frame is on its way already
    SVS_HIP(ctx, hipStreamWaitEvent(ctx->stream, fe->ev_upload[stage], 0));
  } else {
    if ((rc = stage_acquire(fe, stage))) return rc;
    if ((rc = frontend_upload(fe, stage, ctx->stream, h_left, lstride, h_right, rstride, h_disp, dstride))) return rc;
  }
  *stage_out = stage;
  return SVS_OK;
}
static int frontend_end(svs_frontend *fe, int stage, bool host_images) {
  svs_ctx *ctx = fe->ctx;
  if (host_images) {
    SVS_HIP(ctx, hipEventRecord(fe->ev_upload[stage], ctx->stream));      // images (this stream waited for a prefetch) and poses have left the staging set
    fe->i_stage = 1 - fe->i_stage;
  }
  fe->n_submitted++;
  SVS_HIP(ctx, hipEventRecord(fe->ev_done[fe->n_submitted & 1], ctx->stream));
  frontend_rotate(fe);
  return SVS_OK;
}

// StereoFrontend::processFirstFrame (stereo_frontend.cpp:110-131): disparity, FAST with 5 trials, reference cloud at the identity
extern "C" int svs_frontend_first_frame(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride,
                                        const float *h_disp, int dstride) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && fe->B == 1 && !fe->submitted);
  SVS_DEVICE(ctx);
  int stage, rc;
  if ((rc = frontend_begin(fe, h_left, lstride, h_right, rstride, h_disp, dstride, &stage))) return rc;
  const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  double *in_T = stage_poses(fe, stage);
  for (int i = 0; i < 12; ++i) in_T[i] = I[i];
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_small, in_T, sizeof(double) * 12, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = frontend_chain(fe, true, DispView{fe->d_disp[fe->i_cur], fe->stride[0], fe->lvl_elems[0]}))) return rc;
  if ((rc = frontend_end(fe, stage, true))) return rc;
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  fe->have_prev = true;
  return SVS_OK;
}

/* processFirstFrame for all streams, frames in device memory (in == NULL: written in place through svs_frontend_input_view) */
extern "C" int svs_frontend_first_frames(svs_frontend *fe, const svs_frames_dev *in) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && !fe->submitted && !fe->prefetched);
  SVS_DEVICE(ctx);
  int rc;
  if (in && (rc = frontend_take_device_frames(fe, in))) return rc;
  std::vector<double> I((size_t)fe->B * 12, 0.0);
  for (int b = 0; b < fe->B; ++b) I[12 * b] = I[12 * b + 5] = I[12 * b + 10] = 1.0;
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_small, I.data(), sizeof(double) * 12 * fe->B, hipMemcpyHostToDevice, ctx->stream));
  DispView dv = in && in->d_disp ? DispView{in->d_disp, in->dstride, in->d_bstride} : DispView{fe->d_disp[fe->i_cur], fe->stride[0], fe->lvl_elems[0]};
  if ((rc = frontend_chain(fe, true, dv))) return rc;
  if ((rc = frontend_end(fe, 0, false))) return rc;
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  fe->have_prev = true;
  return SVS_OK;
}

/* processFrame for all streams, asynchronous on the context's stream: frames in device memory (in == NULL: written in place through
   svs_frontend_input_view), poses [n_streams][12] from the host.  Results stay on the device until svs_frontend_results / svs_frontend_poses. */
extern "C" int svs_frontend_process_frames(svs_frontend *fe, const svs_frames_dev *in, const double *h_T_cur_from_actkey, const double *h_T_actkey_from_w) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && h_T_cur_from_actkey && h_T_actkey_from_w && fe->have_prev && !fe->submitted && !fe->prefetched);
  SVS_DEVICE(ctx);
  int rc;
  if (fe->timing) SVS_HIP(ctx, hipEventRecord(fe->ev_stage[0], ctx->stream));
  if (in && (rc = frontend_take_device_frames(fe, in))) return rc;
  const int stage = fe->i_stage;
  if ((rc = stage_acquire(fe, stage))) return rc;
  double *in_T = stage_poses(fe, stage);
  const size_t nT = (size_t)12 * fe->B;
  __builtin_memcpy(in_T, h_T_cur_from_actkey, sizeof(double) * nT);
  __builtin_memcpy(in_T + nT, h_T_actkey_from_w, sizeof(double) * nT);
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_small, in_T, sizeof(double) * 2 * nT, hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipEventRecord(fe->ev_upload[stage], ctx->stream));
  fe->i_stage = 1 - fe->i_stage;
  DispView dv = in && in->d_disp ? DispView{in->d_disp, in->dstride, in->d_bstride} : DispView{fe->d_disp[fe->i_cur], fe->stride[0], fe->lvl_elems[0]};
  const bool ext_frames = in && fe->ext_left && in->d_disp && !fe->prm.use_block_matching;      // caller-owned, complete device frames: the pipelined schedule may run
  if ((rc = frontend_chain(fe, false, dv, ext_frames, in ? static_cast<hipEvent_t>(in->ready_event) : nullptr))) return rc;
  return frontend_end(fe, 0, false);
}

static void fill_result(const svs_frontend *fe, const uint8_t *small, int stream, svs_frame_result *out) {
  const size_t B = (size_t)fe->B;
  const double *T = reinterpret_cast<const double *>(small) + 12 * (size_t)stream;
  for (int i = 0; i < 12; ++i) out->T_cur_from_actkey[i] = T[i];
  const uint8_t *p = small + sizeof(double) * 48 * B;
  __builtin_memcpy(&out->pose_stats, p + sizeof(svs_pose_opt_stats) * stream, sizeof(svs_pose_opt_stats));
  p += sizeof(svs_pose_opt_stats) * B;
  __builtin_memcpy(&out->point_stats, p + sizeof(svs_point_stats) * stream, sizeof(svs_point_stats));
  p += sizeof(svs_point_stats) * B;
  __builtin_memcpy(&out->dense_passes, p + sizeof(int32_t) * stream, sizeof(int32_t));
  out->n_points = fe->n_points[stream];
  out->n_matched = out->n_points > 0 ? out->pose_stats.num_obs : 0;
  out->tracking_ok = out->n_matched >= fe->prm.min_matches ? 1 : 0;                            // matchAndTrack's minimum (stereo_frontend.cpp:1053-1056)
  // dense_passes < 0: the multi-workgroup tracker could not get its workgroups resident together (the device is shared with other work) and left the pose
  // untouched -- the frame was NOT tracked, whatever the matcher made of the motion-model pose (the callers turn this into SVS_ERR_BUSY)
  if (out->dense_passes < 0) out->tracking_ok = 0;
}

/* blocking: the results of one stream of the last svs_frontend_process_frames */
extern "C" int svs_frontend_results(svs_frontend *fe, int stream, svs_frame_result *out, svs_match_result *h_matches, svs_gated_point *h_gated) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && out && stream >= 0 && stream < fe->B && !fe->submitted);
  SVS_DEVICE(ctx);
  std::vector<uint8_t> small(fe->small_bytes);
  SVS_HIP(ctx, hipMemcpyAsync(small.data(), fe->d_small, fe->small_bytes, hipMemcpyDeviceToHost, ctx->stream));
  const int n = fe->n_points[stream];
  if (n > 0 && h_matches) SVS_HIP(ctx, hipMemcpyAsync(h_matches, fe->d_res + (size_t)stream * fe->max_points, sizeof(svs_match_result) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  if (n > 0 && h_gated) SVS_HIP(ctx, hipMemcpyAsync(h_gated, fe->d_gated + (size_t)stream * fe->max_points, sizeof(svs_gated_point) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  fill_result(fe, small.data(), stream, out);
  if (out->dense_passes < 0) { ctx->err = "dense tracker: workgroups of one stream not co-resident (device busy); frame not tracked, call may be repeated"; return SVS_ERR_BUSY; }
  return SVS_OK;
}

/* blocking: refined poses [n_streams][12], tracking flags [n_streams] (either may be NULL) of the last svs_frontend_process_frames */
extern "C" int svs_frontend_poses(svs_frontend *fe, double *h_T_cur_from_actkey, int32_t *h_tracking_ok) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && !fe->submitted);
  SVS_DEVICE(ctx);
  std::vector<uint8_t> small(fe->small_bytes);
  SVS_HIP(ctx, hipMemcpyAsync(small.data(), fe->d_small, fe->small_bytes, hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int b = 0; b < fe->B; ++b) {
    svs_frame_result r;
    fill_result(fe, small.data(), b, &r);
    if (h_T_cur_from_actkey) for (int i = 0; i < 12; ++i) h_T_cur_from_actkey[12 * b + i] = r.T_cur_from_actkey[i];
    if (h_tracking_ok) h_tracking_ok[b] = r.tracking_ok;
  }
  return SVS_OK;
}

/* processFrame of stream 0 in two halves: submit enqueues upload (unless prefetched: pass NULL images) + all stages + the download and returns;
   wait blocks and hands the results out.  Between the two the caller may prefetch the next frame. */
extern "C" int svs_frontend_submit_frame(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride, const float *h_disp, int dstride,
                                         const double *T_cur_from_actkey, const double *T_actkey_from_w, int want_matches, int want_gated) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && fe->B == 1 && T_cur_from_actkey && T_actkey_from_w && fe->have_prev && !fe->submitted);
  SVS_DEVICE(ctx);
  int stage, rc;
  if (fe->timing) SVS_HIP(ctx, hipEventRecord(fe->ev_stage[0], ctx->stream));
  if ((rc = frontend_begin(fe, h_left, lstride, h_right, rstride, h_disp, dstride, &stage))) return rc;
  double *in_T = stage_poses(fe, stage);
  for (int i = 0; i < 12; ++i) { in_T[i] = T_cur_from_actkey[i]; in_T[12 + i] = T_actkey_from_w[i]; }
  SVS_HIP(ctx, hipMemcpyAsync(fe->d_small, in_T, sizeof(double) * 24, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = frontend_chain(fe, false, DispView{fe->d_disp[fe->i_cur], fe->stride[0], fe->lvl_elems[0]}))) return rc;
  // one download: small block, then the records
  const int n = fe->n_points[0];
  if (fe->d_out_block && n > 0 && want_matches && want_gated) {      // the device block has the pinned buffer's layout: everything up to the last gated record in ONE copy
    SVS_HIP(ctx, hipMemcpyAsync(fe->h_out, fe->d_out_block, fe->small_bytes + sizeof(svs_match_result) * (size_t)fe->max_points + sizeof(svs_gated_point) * (size_t)n,
                                hipMemcpyDeviceToHost, ctx->stream));
  } else {
    SVS_HIP(ctx, hipMemcpyAsync(fe->h_out, fe->d_small, fe->small_bytes, hipMemcpyDeviceToHost, ctx->stream));
    svs_match_result *o_res = reinterpret_cast<svs_match_result *>(fe->h_out + fe->small_bytes);
    svs_gated_point *o_gated = reinterpret_cast<svs_gated_point *>(fe->h_out + fe->small_bytes + sizeof(svs_match_result) * (size_t)fe->max_points);
    if (n > 0 && want_matches) SVS_HIP(ctx, hipMemcpyAsync(o_res, fe->d_res, sizeof(svs_match_result) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (n > 0 && want_gated) SVS_HIP(ctx, hipMemcpyAsync(o_gated, fe->d_gated, sizeof(svs_gated_point) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  }
  if ((rc = frontend_end(fe, stage, true))) return rc;
  fe->submitted = true; fe->want_matches = want_matches != 0; fe->want_gated = want_gated != 0;
  return SVS_OK;
}
extern "C" int svs_frontend_wait_frame(svs_frontend *fe, svs_frame_result *out, svs_match_result *h_matches, svs_gated_point *h_gated) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && out && fe->submitted);
  SVS_REQUIRE(ctx, (!h_matches || fe->want_matches) && (!h_gated || fe->want_gated));
  SVS_DEVICE(ctx);
  fe->submitted = false;
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  fill_result(fe, fe->h_out, 0, out);
  const int n = fe->n_points[0];
  const svs_match_result *o_res = reinterpret_cast<const svs_match_result *>(fe->h_out + fe->small_bytes);
  const svs_gated_point *o_gated = reinterpret_cast<const svs_gated_point *>(fe->h_out + fe->small_bytes + sizeof(svs_match_result) * (size_t)fe->max_points);
  if (n > 0 && h_matches) __builtin_memcpy(h_matches, o_res, sizeof(svs_match_result) * (size_t)n);
  if (n > 0 && h_gated) __builtin_memcpy(h_gated, o_gated, sizeof(svs_gated_point) * (size_t)n);
  return SVS_OK;
}

extern "C" int svs_frontend_process_frame(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride,
                                          const float *h_disp, int dstride, const double *T_cur_from_actkey, const double *T_actkey_from_w,
                                          svs_frame_result *out, svs_match_result *h_matches, svs_gated_point *h_gated) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && out);
  const int rc = svs_frontend_submit_frame(fe, h_left, lstride, h_right, rstride, h_disp, dstride, T_cur_from_actkey, T_actkey_from_w, h_matches != nullptr,
                                           h_gated != nullptr);
  if (rc) return rc;
  return svs_frontend_wait_frame(fe, out, h_matches, h_gated);
}

/* computeDensePointCloudCpu / Gpu again, at a pose the caller decided on after the frame (keyframe switch, stereo_frontend.cpp:277-281) */
extern "C" int svs_frontend_recompute_cloud(svs_frontend *fe, const double *T_cur_from_actkey) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && T_cur_from_actkey && fe->have_prev && !fe->submitted);      // T_cur_from_actkey [n_streams][12]
  SVS_DEVICE(ctx);
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  SVS_HIP(ctx, hipMemcpy(fe->d_small, T_cur_from_actkey, sizeof(double) * 12 * fe->B, hipMemcpyHostToDevice));
  const float *disp = fe->last_disp;      // the disparity the frame processed last was given (the caller's buffer, if it passed one) or produced
  SVS_REQUIRE(ctx, disp);
  for (int l = 0; l < 3; ++l) {
    int rc;
    if (fe->prm.cuda_build)
      rc = svs_pointcloud_full_pose(ctx, fe->d_small, &fe->cams[l], disp, fe->last_dstride, fe->last_dbstride, fe->w[l], fe->h[l], fe->w[l], fe->cloud_elems[l] / 4, 1 << l, fe->d_cloud[l], fe->B);
    else
      rc = svs_pointcloud_cpu_sem(ctx, disp, fe->last_dstride, fe->last_dbstride, &fe->cams[l], l, fe->d_small, fe->d_cloud[l], fe->cloud_elems[l], fe->B);
    if (rc) return rc;
  }
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVS_OK;
}

/* device views for tests / chaining: level images of the frame processed last, its disparity, the reference clouds -- of one stream */
extern "C" int svs_frontend_device_view(svs_frontend *fe, int stream, const uint8_t **d_pyr_last, int32_t *stride, const float **d_disp, const float **d_cloud,
                                        svs_fast **fast) {
  if (!fe || stream < 0 || stream >= fe->B) return SVS_ERR_INVALID;
  for (int l = 0; l < 3; ++l) {
    if (d_pyr_last) d_pyr_last[l] = fe->d_pyr[fe->i_prev][l] + fe->lvl_elems[l] * stream;
    if (stride) stride[l] = fe->stride[l];
    if (d_cloud) d_cloud[l] = fe->d_cloud[l] + fe->cloud_elems[l] * stream;
  }
  if (d_disp) *d_disp = fe->last_disp ? fe->last_disp + fe->last_dbstride * stream : nullptr;
  if (fast) *fast = fe->fast;
  return SVS_OK;
}

/* profiling: hipEvents between the stages of the next svs_frontend_process_frame(s) / submit_frame call (off by default: ~4 us of stream time each) */
extern "C" int svs_frontend_set_timing(svs_frontend *fe, int on) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe);
  SVS_DEVICE(ctx);
  if (on && !fe->ev_stage[0])
    for (hipEvent_t &e : fe->ev_stage) SVS_HIP(ctx, hipEventCreate(&e));
  fe->timing = on != 0;
  return SVS_OK;
}
/* blocking: ms[SVS_FRONTEND_STAGES] of the last timed call, in the order of the reference's per_mon_ stages: preprocess (upload / copy + pyramid),
   dense tracking, stereo, fast, match, pose refinement (calcFastMotionOnly), process points, dense point cloud */
extern "C" int svs_frontend_stage_times(svs_frontend *fe, float *ms) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && ms && fe->timing && fe->ev_stage[0]);
  SVS_DEVICE(ctx);
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < SVS_FRONTEND_STAGES; ++k) SVS_HIP(ctx, hipEventElapsedTime(&ms[k], fe->ev_stage[k], fe->ev_stage[k + 1]));
  return SVS_OK;
}
/* blocking: the accept / reject record of the dense tracker's LM loop of one stream in the last call (level, accepted, chi2 before / after per chi2
   evaluation); *n = records produced (only the first min(*n, cap, 64) are stored) */
extern "C" int svs_frontend_dense_records(svs_frontend *fe, int stream, svs_dense_lm_record *h_rec, int cap, int32_t *n) {
  svs_ctx *ctx = fe ? fe->ctx : nullptr;
  SVS_REQUIRE(ctx, fe && stream >= 0 && stream < fe->B && n && cap >= 0 && (cap == 0 || h_rec));
  SVS_DEVICE(ctx);
  SVS_HIP(ctx, hipMemcpyAsync(n, fe->d_nrec + stream, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int k = std::min(std::min((int)*n, cap), REC_CAP);
  if (k > 0) SVS_HIP(ctx, hipMemcpy(h_rec, fe->d_rec + (size_t)stream * REC_CAP, sizeof(svs_dense_lm_record) * (size_t)k, hipMemcpyDeviceToHost));
  return SVS_OK;
}
