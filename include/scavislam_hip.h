/*
 * scavislam_hip.h -- C ABI of the MI355X (gfx950) implementation of ScaViSLAM's two hot paths.
 *
 * The reference (strasdat/ScaViSLAM) has no FFI/plugin layer: its seams are C++ member functions
 * switched at compile time by SCAVISLAM_CUDA_SUPPORT.  Every entry point
 * below names the reference interface it replaces (file:line under /root/reference/scavislam).
 * INTEGRATION.md shows the adaptor a maintainer adds on the reference side; the C++ adaptor
 * classes with the reference's method names are in include/scavislam_hip.hpp.
 *
 * Conventions:
 *  - every function returns int status, 0 = SVS_OK; never throws, never aborts;
 *  - an svs_ctx owns one HIP stream + scratch; one ctx per calling thread (the reference calls
 *    FAST/matcher from both the front-end and the back-end thread); no process-global state;
 *  - pointers named d_* are DEVICE pointers, h_* are HOST pointers; strides are in ELEMENTS of
 *    the pointed-to type; *_bstride is the element distance between consecutive batch slots;
 *  - device entry points are asynchronous on the ctx stream; svs_ctx_sync() or any *_download /
 *    h_* output makes results visible (blocking, like every reference call);
 *  - poses are 3x4 row-major double[12] (R | t), T_a_from_b convention of the reference;
 *  - a "batch" is a set of independent camera streams / frames processed by one launch.
 */
#ifndef SCAVISLAM_HIP_H
#define SCAVISLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever a struct of this header grows or a signature changes (round 3 grew svs_pose_opt_params / svs_match_args and put a `stream`
   argument into svs_frontend_device_view without one -- INTEGRATION.md section 6).  A caller checks svs_api_version() == SVS_API_VERSION once, zero-initialises
   every parameter struct (or takes it from the *_default() initialisers) and sets only the fields it knows. */
#define SVS_API_VERSION 7
int svs_api_version(void);             /* the SVS_API_VERSION the loaded library was built with */

enum {
  SVS_OK = 0,
  SVS_ERR_INVALID = 1,      /* bad argument */
  SVS_ERR_HIP = 2,          /* HIP runtime error; see svs_last_error */
  SVS_ERR_NO_DEVICE = 3,    /* no gfx950 device / kernels not loadable */
  SVS_ERR_CAPACITY = 4,     /* caller-provided capacity too small */
  SVS_ERR_UNSUPPORTED = 5,
  SVS_ERR_BUSY = 6          /* the workgroups of a multi-workgroup kernel could not synchronise within its bounded wait (the device is shared with
                               other work): nothing was applied, the call may be retried */
};

#define SVS_NUM_PYR_LEVELS 3   /* global.h:107 */
#define SVS_MAX_CELLS 64

/* ---- POD types ---------------------------------------------------------------------------*/
typedef struct { double f, cx, cy, b; int32_t w, h; } svs_cam;   /* frame_grabber-impl.cpp:48-60 */

typedef struct {                       /* FastGrid, fast_grid.cpp:23-58 / keyframes.h:31-44 */
  int32_t gx, gy, cell_w, cell_h;
  int32_t min_inner, min_outer, max_inner, max_outer;
  int32_t fast_min, fast_max;
  int32_t thr[SVS_MAX_CELLS];
} svs_fastgrid;

typedef struct {                       /* CandidatePoint<3>, data_structures.h:37-69 */
  double xyz_anchor[3];
  double anchor_obs_pyr[3];
  int32_t anchor_level, kf_index, point_id, pad_;
} svs_candidate_point;

typedef struct {                       /* keyframe_map entry + vertex_map pose */
  double T_anchor_from_w[12];
  const uint8_t *pyr[3];               /* DEVICE pointers for svs_match */
  int32_t stride[3];
  int32_t pad_;
} svs_keyframe;

enum { SVS_MATCH_OK = 0, SVS_MATCH_NO_ANCHOR, SVS_MATCH_BORDER, SVS_MATCH_DEPTH,
       SVS_MATCH_TEXTURE, SVS_MATCH_NONE, SVS_MATCH_NO_DISP,
       SVS_MATCH_SKIPPED /* svs_frontend_*: a neighbour's new-point list behind matchAndTrack's cut (stereo_frontend.cpp:1000-1003) */ };

typedef struct {
  int32_t status, u, v, znssd;
  double obs[3];
  double xyz_actkey[3];
} svs_match_result;

typedef struct {                       /* GpuTrackingData, gpu/dense_tracking.cuh:28-277 */
  double H[21];                        /* packed upper-by-column */
  double b[6];
  double chi2;
  int64_t n_valid;
} svs_dense_sums;

typedef struct {                       /* addObsToG2o arguments, slam_graph-impl.cpp:44-97 */
  double obs[3];
  double info[3];
  int32_t point, pose, anchor, pad_;
} svs_ba_edge;

typedef struct {                       /* addConstraintToG2o, slam_graph-impl.cpp:99-126 */
  double T_21[12];
  double info[36];
  int32_t pose1, pose2;
} svs_ba_constraint;

typedef struct {                       /* OptParams slam_graph.hpp:36-50 + setupG2o/optimize */
  int32_t num_iters;
  int32_t use_robust;
  double huber_delta;
  double lambda_init;
  int32_t max_trials;
  int32_t self_edge_mode;              /* 0 = G2O_LITERAL, 1 = EXACT */
} svs_ba_params;

typedef struct {
  int32_t iterations, trials, accepted, terminated;
  double chi2_init, chi2_final, lambda_final;
} svs_ba_stats;

/* ---- context -----------------------------------------------------------------------------*/
typedef struct svs_ctx svs_ctx;
/* hip_stream: a hipStream_t to run on (e.g. torch's current stream), or NULL to create one */
int svs_ctx_create(int device, void *hip_stream, svs_ctx **out);
int svs_ctx_destroy(svs_ctx *ctx);
int svs_ctx_sync(svs_ctx *ctx);
/* experiment / test switches (0 = automatic choice): "trk_nwg" workgroups per stream of the latency-mode quarter-grid
   tracker, "trk_regs" its register budget (1: one workgroup per CU, 2: two), "full_nwg" workgroups per stream of the
   full-resolution tracker.  The environment (SVS_TRK_NWG, SVS_TRK_ONE_PER_CU / SVS_TRK_TWO_PER_CU, SVS_FULL_NWG) only
   supplies the initial values, read once by svs_ctx_create.  A/B switches between kernels that return the same results:
   "match_legacy" (0: four candidate points per wave where the search window allows it, 1: the round-1/2 kernel, 2: one wave per
   point with the lean scan), "mo_legacy" (1: the record-walking motion-only kernel), "fe_overlap" (default 1: the one-call
   front end enqueues FAST / block matching on a side stream beside the dense tracker; 0: everything on the context's stream),
   "xcd_swizzle" (default 1: tile kernels whose neighbours share image lines -- FAST score, pyramid, block matching, the four-points-per-wave matcher -- take their
   blocks in XCD-contiguous order; 0: the dispatcher's round robin),
   "trk_flat" (default 1: batches of more than one stream per CU run the flat tracker kernel -- the sweep inlined, the LM state in LDS; 0: the
   round-5 kernel, same bits), "trk_split" (default 10: in such batches a stream still iterating after that many trials on the finest level is
   finished by a second launch with up to eight workgroups per stream -- same accept decisions, poses equal to 1e-12; 0: one launch).
   A context and every handle made from it are used by ONE thread at a time. */
int svs_ctx_set_option(svs_ctx *ctx, const char *name, int value);
/* The accept test of the quarter-grid tracker: DenseTracker::denseTrackingCpu accepts an LM step iff `float chi2 - float new_chi2 > 0` on two sums accumulated
   sequentially in one float (dense_tracking.cpp:229-262, 341-383).  Default ("trk_lazy_chi2" = 1): the library takes exactly those decisions -- its f64 sums decide
   wherever their difference is outside the rigorous rounding-error bound of the float sums, and inside the bound the float sums themselves are formed, bit for bit.
   "trk_seq_chi2" = 1 forms every sum by the literal sequential chain (slow; cross-check), "trk_lazy_chi2" = 0 compares the f64 sums alone (rounds 1-4).
   Counters (blocking): "trk_exact_sums" float sums formed so far by this context's tracker launches, "trk_exact_fallbacks" how many of them needed the chain.
   Host-side counters of the spin gate (kernels whose workgroups wait for each other inside one launch -- latency-mode trackers, multi-workgroup solves -- are kept from
   starving each other across the contexts of a process; DenseTracker / SlamGraph::optimize on two threads, stereo_slam.cpp:196, backend.cpp:157-224):
   "spin_lane_launches" launches of this context that skipped the gate (<= 16 workgroups AND provably fitting beside everything of that kind in flight),
   "spin_gated_launches" launches that went through it. */
int svs_ctx_get_stat(svs_ctx *ctx, const char *name, long long *out);
void *svs_ctx_stream(svs_ctx *ctx);
const char *svs_last_error(svs_ctx *ctx);
int svs_malloc(svs_ctx *ctx, size_t bytes, void **d_ptr);
int svs_free(svs_ctx *ctx, void *d_ptr);
int svs_memcpy_h2d(svs_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int svs_memcpy_d2h(svs_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);  /* blocking */
/* hipEvent timing on the ctx stream (bench.py roofline leg) */
int svs_timer_start(svs_ctx *ctx);
int svs_timer_stop_ms(svs_ctx *ctx, float *ms);                                  /* blocking */

/* ---- preprocessing: replaces FrameGrabber::preprocessing, frame_grabber.cpp:285-336 --------*/
/* one cv::pyrDown step on u8 (cv::buildPyramid at :290), bit-exact to OpenCV semantics */
int svs_pyr_down_u8(svs_ctx *ctx, const uint8_t *d_src, int w, int h, int sstride, size_t s_bstride,
                    uint8_t *d_dst, int dstride, size_t d_bstride, int batch);
/* convertTo(CV_32F,1/255.) + Sobel(ksize=1) dx,dy (:315-333) */
int svs_convert_sobel_f32(svs_ctx *ctx, const uint8_t *d_src, int w, int h, int sstride,
                          size_t s_bstride, float *d_img, float *d_dx, float *d_dy, int fstride,
                          size_t f_bstride, int batch);

/* ---- stereo block matching: replaces StereoFrontend::calcDisparityCpu (stereo_frontend.cpp:620-653),
   i.e. cv::StereoBM(left pyr level 0, right, disp, CV_32F) with the state set there.  [cv::StereoBM is
   OpenCV 2.4.2, external; semantics as restated in oracle/stereo.c incl. its definitions D1/D2.] ---------*/
typedef struct {
  int32_t prefilter_cap;      /* stereo_frontend.cpp:626  (31)  */
  int32_t sad_window;         /* :627  (7; only 7 supported)    */
  int32_t min_disparity;      /* :628  (0; only 0 supported)    */
  int32_t num_disparities;    /* :636  (num_disp16*16 = 32; only 32 supported) */
  int32_t texture_threshold;  /* :630  (10) */
  int32_t uniqueness_ratio;   /* :631  (15) */
  int32_t speckle_window;     /* :632  (100; 0 = no speckle filter) */
  int32_t speckle_range;      /* :633  (32)  */
  int32_t disp12_max_diff;    /* :634  (1; < 0 = no left-right check) */
} svs_stereo_params;
typedef struct svs_stereo svs_stereo;
/* scratch for `max_batch` independent w x h frames (prefiltered images, 16-bit disparity, cost, labels) */
int svs_stereo_create(svs_ctx *ctx, int w, int h, int max_batch, const svs_stereo_params *prm, svs_stereo **out);
int svs_stereo_destroy(svs_stereo *s);
/* d_disp[b][y*dstride + x] = disparity in pixels, (min_disparity - 1) where filtered (all device pointers;
   strides in elements, batch strides in elements of the respective type) */
int svs_stereo_compute(svs_stereo *s, const uint8_t *d_left, int lstride, size_t l_bstride, const uint8_t *d_right,
                       int rstride, size_t r_bstride, float *d_disp, int dstride, size_t d_bstride, int n_batch);

/* ---- grid FAST: replaces FastGrid (fast_grid.h:27-63) ----------------------------------------*/
typedef struct svs_fast svs_fast;
/* one FastGrid per level (stereo_frontend.cpp:73-88); `batch` independent threshold states */
int svs_fast_create(svs_ctx *ctx, int n_levels, const int32_t *w, const int32_t *h,
                    const svs_fastgrid *grids, int batch, int corner_cap_per_level,
                    svs_fast **out);
int svs_fast_destroy(svs_fast *f);
/* FastGrid::detectAdaptively(img, trials, qt) for all levels and `n_batch` slots
   (fast_grid.cpp:86-152); trials == 0 => FastGrid::detect at the stored thresholds (:60-83).
   d_img[l] = level-l image of slot 0. */
int svs_fast_detect(svs_fast *f, const uint8_t *const *d_img, const int32_t *stride,
                    const size_t *bstride, int n_batch, int trials);
/* blocking download of one slot/level: corners in the reference's quadtree insertion order
   (cells row-major, row-major inside the cell), per-cell counts, threshold used by the last
   detection of each cell, and the persistent thresholds (cell_grid2d()). Any pointer may be NULL */
int svs_fast_download(svs_fast *f, int slot, int level, int16_t *h_xy, int cap, int32_t *h_n,
                      int32_t *h_cell_count, int32_t *h_emit_thr, int32_t *h_thr_state);
int svs_fast_set_thresholds(svs_fast *f, int slot, int level, const int32_t *h_thr);
/* device view of a level's result for callers that chain their own kernels: the corner bitmap the matcher reads (1 bit per pixel, set = a corner of the last
   detection; pixel (x, y) of cell column ci = x / cell_w is bit x + (cell_column_bits - cell_w) * ci of the row at byte y * row_stride_bytes -- cell columns start
   on dword boundaries, rows end in >= 8 zero bytes) and the per-cell thresholds of that detection.  (API 7: replaces the dense score map of API <= 6.) */
int svs_fast_device_view(svs_fast *f, int level, const uint32_t **d_corner_bits, int32_t *row_stride_bytes, size_t *batch_stride_bytes,
                         int32_t *cell_column_bits, const int32_t **d_emit_thr, size_t *emit_bstride);

/* ---- guided matcher: replaces GuidedMatcher<StereoCamera>::match (matcher.hpp:67-83) --------*/
typedef struct {
  const svs_keyframe *d_kfs; int32_t n_kf;        /* keyframe table (device) */
  const svs_candidate_point *d_pts; int32_t n_pts; /* per slot: [n_batch][n_pts] */
  const double *d_T_cur_from_w;                    /* [n_batch][12] */
  const double *d_T_w_from_actkey;                 /* [n_batch][12] */
  const uint8_t *d_cur_pyr[3]; int32_t cur_stride[3]; size_t cur_bstride[3];
  const float *d_disp; int32_t disp_stride; size_t disp_bstride;
  svs_cam cam_vec[3];
  int32_t search_radius, thr_mean, thr_std;        /* 8, 22, 10 at stereo_frontend.cpp:989-1004 */
  int32_t n_batch;
  /* element distances between the tables of consecutive slots; 0 = the default: one keyframe table shared by all slots,
     point and result arrays packed [n_batch][n_pts] */
  size_t kf_bstride, pts_bstride, out_bstride;
} svs_match_args;
/* corners come from `f` (the feature_tree argument of the reference): candidate set and
   tie-break order are those of QuadTree::query. d_out: [n_batch][n_pts] */
int svs_match(svs_ctx *ctx, const svs_match_args *a, svs_fast *f, svs_match_result *d_out);

/* ---- motion-only refinement: replaces BA_SE3_XYZ_STEREO::calcFastMotionOnly (pose_optimizer.h:134-298) as
   called behind the matcher at stereo_frontend.cpp:1058-1063 ------------------------------------------------*/
typedef struct {
  int32_t robust_kernel;   /* PoseOptimizerParams(true, 2, 15): pose_optimizer.h:36-58 */
  int32_t num_iter;
  double kernel_param;
  double initial_mu;       /* -1 => tau * max diag(J^T J) (pose_optimizer.h:187-190) */
  double tau;              /* 1e-5 */
  int32_t min_obs;         /* matchAndTrack returns before calcFastMotionOnly with fewer than 20 observations (stereo_frontend.cpp:1053-1056):
                              with fewer than min_obs the pose is left untouched and status = 3; 0 = no minimum */
  int32_t pad_;
} svs_pose_opt_params;
/* PoseOptimizerParams(robust_kernel = true, kernel_param = 2, num_iter = 15) as the front end passes it (stereo_frontend.cpp:1061); initial_mu = -1, tau = 1e-5
   (pose_optimizer.h:36-58), min_obs = 0 */
void svs_pose_opt_params_default(svs_pose_opt_params *p);
typedef struct {           /* OptimizerStatistics, pose_optimizer.h:59-98 */
  double initial_chi2, chi2, max_err;
  int32_t num_obs;
  int32_t status;          /* 0 ok; 1 empty observation list (the reference asserts); 2 residual became NaN (the reference throws);
                              3 fewer than min_obs observations: nothing done */
} svs_pose_opt_stats;
/* obs_list / point_list = the SVS_MATCH_OK entries of d_results[b][0..n) in order (several svs_match outputs may be
   concatenated, as matchAndTrack appends into one TrackData); d_T_io[b][12] = T_cur_from_actkey in/out */
int svs_motion_only(svs_ctx *ctx, const svs_match_result *d_results, int n, size_t res_bstride, const svs_cam *cam,
                    const svs_pose_opt_params *prm, double *d_T_io, svs_pose_opt_stats *d_stats, int batch);

/* ---- gating of the matched points: the data-parallel part of StereoFrontend::processMatchedPoints
   (stereo_frontend.cpp:834-974), which runs right behind calcFastMotionOnly on the same TrackData ----------*/
typedef struct {           /* per matcher record; only meaningful where status == SVS_MATCH_OK */
  int32_t accepted;        /* |uvu - map_uvu(T xyz)| < max_reproj_error * 2^level (u, v), < 3 max_reproj_error (u_right): :869-871 */
  int32_t is_new;          /* id_obs.point_id < num_new_feat_matched (:920): goes to new_point_list, else track_point_list */
  double uv_pyr[2];        /* pyrFromZero_2d(uvu.head(2), anchor_level): point_tree insert position (:918), draw line start */
  double curkey_uv_pyr[2]; /* pyrFromZero_2d(se3xyz.map(SE3(), point), anchor_level) (:896-898) */
} svs_gated_point;
typedef struct {           /* PointStatistics (stereo_frontend.h) + the track-length accumulators (:857-858,925-926,966) */
  int32_t num_points_grid2x2[4];   /* [i * 2 + j], i from u, j from v (:875-880) */
  int32_t num_points_grid3x3[9];   /* [i * 3 + j] (:882-892) */
  int32_t num_matched_points[3];   /* per anchor level (:895) */
  int32_t num_track_points;
  int32_t num_obs;                 /* obs_list.size() */
  int32_t pad_[2];
  double sum_track_length;         /* av_track_length_ = sum_track_length / num_track_points (:966) */
} svs_point_stats;
/* d_results [batch][n] (res_bstride records apart) and d_pts [batch][n] (pts_bstride apart) are the arrays svs_match was
   called with / produced (concatenated calls allowed, as in matchAndTrack); records with index < n_new_records stem from
   the "new feature" match calls (stereo_frontend.cpp:989-1030), so an OK record is "new" iff its index is below that --
   the same split as point_id < num_new_feat_matched.  d_T [batch][12] = T_cur_from_actkey after calcFastMotionOnly.
   cam = level-0 stereo camera.  max_reproj_error: ui.max_reproj_error, default 2 (:845-846). */
int svs_process_matched_points(svs_ctx *ctx, const svs_match_result *d_results, const svs_candidate_point *d_pts, int n,
                               size_t res_bstride, size_t pts_bstride, int n_new_records, const svs_cam *cam,
                               const double *d_T, float max_reproj_error, svs_gated_point *d_gated, size_t gated_bstride,
                               svs_point_stats *d_stats, int batch);

/* ---- dense tracker: replaces DenseTracker / GpuTracker ---------------------------------------*/
/* computeDensePointCloudCpu (dense_tracking.cpp:393-423): quarter-grid cloud of one level */
int svs_pointcloud_cpu_sem(svs_ctx *ctx, const float *d_disp, int disp_stride, size_t disp_bstride,
                           const svs_cam *cam, int level, const double *d_T_cur_from_actkey,
                           float *d_cloud, size_t cloud_bstride, int batch);
/* one pass of denseTrackingCpu's loop body (dense_tracking.cpp:229-261 / :278-331), CPU-path
   semantics (quarter grid, clamp +-0.1, f64 geometry).  d_out[batch] */
int svs_dense_pass_cpu_sem(svs_ctx *ctx, const float *d_cloud, size_t cloud_bstride,
                           const uint8_t *d_prev_u8, int pstride, size_t p_bstride,
                           const float *d_cur, const float *d_dx, const float *d_dy, int fstride,
                           size_t f_bstride, const svs_cam *cam, const double *d_T, int do_jac,
                           svs_dense_sums *d_out, int batch);
typedef struct {                       /* one entry per chi2 evaluation of the dense trackers' LM loops */
  int32_t level;
  int32_t accepted;                    /* 1 / 0 = trial accepted / rejected (rho > 0: dense_tracking.cpp:142-144 / :367-369); 2 = the level's initial chi2 */
  float chi2, new_chi2;                /* chi2 before the trial, chi2 at the trial pose (both `float` in the reference) */
} svs_dense_lm_record;
/* whole DenseTracker::denseTrackingCpu(SE3*) (dense_tracking.cpp:222-391), device resident:
   3 levels x <=15 LM iterations with no host round trip. d_T_io [batch][12] in/out */
typedef struct {
  const float *d_cloud[3]; size_t cloud_bstride[3];
  const uint8_t *d_prev_u8[3]; int32_t pstride[3]; size_t p_bstride[3];
  const float *d_cur[3]; const float *d_dx[3]; const float *d_dy[3];
  int32_t fstride[3]; size_t f_bstride[3];
  svs_cam cam_vec[3];
  /* optional fused source: current u8 pyramid.  If d_cur_u8[0] != NULL the f32 image and its Sobel
     taps are formed on the fly from it (bit-identical values, ~1/8 of the HBM traffic) and
     d_cur/d_dx/d_dy are ignored, i.e. the convertTo/Sobel part of preprocessing can be skipped. */
  const uint8_t *d_cur_u8[3]; int32_t c8stride[3]; size_t c8_bstride[3];
  /* optional output [batch][3][12]: the pose the LAST H,b pass of each level ran at, i.e. the pose
     DenseTracker::residual_img[level] shows (dense_tracking.cpp:279-329); feed it to
     svs_dense_residual_image_cpu_sem.  NULL = not wanted. */
  double *d_T_jac_out;
  /* optional accept / reject record of the loop, [batch][record_cap] (+ the number produced per stream).  The reference repeats a
     rejected trial once (its undamped solve gives the identical step, so the identical rejection, then stops at trial == 2,
     dense_tracking.cpp:379-384); the device loop stops at the first rejection and records it once. */
  svs_dense_lm_record *d_record_out;
  int32_t record_cap;
  int32_t *d_n_record_out;
} svs_dense_track_args;
int svs_dense_track_cpu_sem(svs_ctx *ctx, const svs_dense_track_args *a, double *d_T_io,
                            int32_t *d_passes_out, int batch);
/* Diagnostic of the accept test above (tests): out[b] = the value of `float chi2 = 0; for (i < n) chi2 += t[b][i];` (dense_tracking.cpp:229-262) for batch rows of
   non-negative terms.  how = 0: formed as the tracker forms it (parallel, bit-identical by construction: csrc/seqsum.h), 1: the same with the terms read past the caches
   (latency mode), 2: by the literal sequential chain.  Rows 16-byte aligned and followed by >= 64 readable floats (bstride >= n + 64, a multiple of 4).
   d_fell_back (optional, [batch]): 1 where the chain was used after all (n > 20 480, or a failed self-check of how = 0 / 1). */
int svs_dense_seq_sum_f32(svs_ctx *ctx, const float *d_terms, int n, size_t bstride, int batch, int how, float *d_out, int32_t *d_fell_back);
/* DenseTracker::residual_img[level] (dense_tracking.cpp:52-54,279-329): float4 per quarter-grid sample as left by an
   H,b pass at pose d_T[b * T_bstride .. +12): (0,1,0,1) no depth, (1,0,0,1) out of frame, else grey 1 - 50 res^2.
   Either d_cur (f32 level image) or d_cur_u8 (fused source, as in svs_dense_track_args) must be given. */
int svs_dense_residual_image_cpu_sem(svs_ctx *ctx, const float *d_cloud, size_t cloud_bstride,
                                     const uint8_t *d_prev_u8, int pstride, size_t p_bstride,
                                     const float *d_cur, int fstride, size_t f_bstride,
                                     const uint8_t *d_cur_u8, int c8stride, size_t c8_bstride,
                                     const svs_cam *cam, const double *d_T, size_t T_bstride,
                                     float *d_res_img4, size_t res_bstride, int batch);
/* ---- full-resolution dense tracker: the CUDA build's DenseTracker / GpuTracker (dense_tracking.cpp:60-215,
   gpu/dense_tracking.cuh:281-342).  Per-pixel arithmetic = the reference's f32 code (pinned against the reference-compiled
   kernels, oracle/_ref); sums in f64.  The texture fetch tex2D(uv + 0.5f) is the exact-weight bilinear at
   RN(uv + 0.5f) - 0.5f with clamp addressing. ------------------------------------------------------------------------*/
/* FrameGrabber::preprocessing, CUDA branch (frame_grabber.cpp:291-313; filters :102-115): level 0 = convertTo(CV_32F, 1/255.),
   level l = gpu::pyrDown(level l-1) on f32, dx / dy = ksize-1 derivative with BORDER_REPLICATE on every level.
   d_img / d_dx / d_dy: `levels` device pointers each; level l has size ((w+1)/2.., (h+1)/2..). [cv::gpu is external: semantics
   as restated in oracle/vision.c] */
int svs_preprocess_gpu_sem(svs_ctx *ctx, const uint8_t *d_src, int w, int h, int sstride, size_t s_bstride,
                           float *const *d_img, float *const *d_dx, float *const *d_dy, const int32_t *fstride,
                           const size_t *f_bstride, int levels, int batch);
typedef struct {
  const float *d_cloud4[3]; int32_t stride_f4[3]; size_t cloud_bstride[3];   /* dev_ref_dense_points_[l]: float4 per pixel, strides in float4 */
  const float *d_prev[3];              /* prev_left().gpu_pyr_float32[l] */
  const float *d_cur[3];               /* cur_left().gpu_pyr_float32[l] */
  const float *d_dx[3], *d_dy[3];      /* gpu_pyr_float32_dx / _dy[l]; all NULL => the derivative taps are formed on the fly from d_cur
                                          (I(x+1) - I(x-1), REPLICATE: bit-identical to the filters of frame_grabber.cpp:102-115, 8 B/px less) */
  int32_t stride_f[3]; size_t f_bstride[3];
  int32_t w[3], h[3];
  double f[3], cx[3], cy[3];           /* cam_vec[l] intrinsics (narrowed to float as GpuIntrinsics::set does) */
  double *d_T_jac_out;                 /* optional [batch][3][12]: pose of the last jacobianReduction of each level = the pose the reference
                                          renders residualImage with (dense_tracking.cpp:177-186) */
  svs_dense_lm_record *d_record_out;   /* optional [batch][record_cap] */
  int32_t record_cap;
  int32_t *d_n_record_out;             /* optional [batch]: records the loop produced (may exceed record_cap; only the first cap are stored) */
} svs_dense_track_full_args;
/* whole DenseTracker::denseTrackingGpu(SE3*) (dense_tracking.cpp:60-193) for `batch` independent streams in ONE launch: levels 2..0,
   damped LM (H += mu diag(H), mu0 = 0.01f), <= 15 accepted steps per level, two rejections in a row stop.  d_T_io [batch][12] in/out.
   d_passes_out [batch] (optional): fused sweeps executed (one per chi2 evaluation), or -1 if the stream's workgroups could not
   synchronise (device oversubscribed; T is then unchanged garbage-free but NOT converged). */
int svs_dense_track_full(svs_ctx *ctx, const svs_dense_track_full_args *a, double *d_T_io, int32_t *d_passes_out, int batch);
/* parity probe: per-pixel terms of jacobianReduction_kernel before its reduction, d_terms8[v * w + u] = {J0..J5, res, valid};
   d_dx == d_dy == NULL selects the on-the-fly derivative taps */
int svs_dense_pixel_terms_full(svs_ctx *ctx, const float *d_cloud4, int w, int h, int stride_f4, const float *d_prev,
                               const float *d_cur, const float *d_dx, const float *d_dy, int stride_f, float f, float cx,
                               float cy, const float *h_T34_colmajor, float *d_terms8);
/* GpuTracker::jacobianReduction / chi2 (gpu/dense_tracking.cuh:291-342, .cu:172-263,376-453):
   full resolution, f32, no clamp; T is GpuMatrix34 (12 floats column-major) by value */
int svs_dense_pass_full(svs_ctx *ctx, const float *d_cloud4, int w, int h, int stride_f4,
                        const float *d_prev, const float *d_cur, const float *d_dx,
                        const float *d_dy, int stride_f, float f, float cx, float cy,
                        const float *h_T34_colmajor, int do_jac, svs_dense_sums *d_out);
/* GpuTracker::residualImage (gpu/dense_tracking.cuh:329-337, .cu:495-569); d_res_img4 has the cloud's stride */
int svs_dense_residual_image_full(svs_ctx *ctx, const float *d_cloud4, int w, int h, int stride_f4,
                                  const float *d_prev, const float *d_cur, int stride_f, float f,
                                  float cx, float cy, const float *h_T34_colmajor, float *d_res_img4);
/* computePointCloud (gpu/dense_tracking.cu:82-148) */
int svs_pointcloud_full(svs_ctx *ctx, const float *h_TQ_colmajor, const float *d_disp, int w, int h,
                        int stride_in, int stride_out, int factor, float *d_cloud4);
/* one level of DenseTracker::computeDensePointCloudGpu (dense_tracking.cpp:195-215) for `batch` streams with the poses on the DEVICE:
   TQ = T_cur_from_actkey^-1 * cam.Q() in double (products summed in ascending k), narrowed to float, then computePointCloud.
   d_T [batch][12]; cam = cam_vec[level]; factor = 2^level; strides in elements (float / float4), batch strides likewise */
int svs_pointcloud_full_pose(svs_ctx *ctx, const double *d_T, const svs_cam *cam, const float *d_disp, int disp_stride, size_t disp_bstride,
                             int w, int h, int stride_out, size_t cloud_bstride, int factor, float *d_cloud4, int batch);

/* ---- one call per frame: replaces the data-parallel part of StereoFrontend::processFrame(bool*) / processFirstFrame()
   (stereo_frontend.h:88-95, stereo_frontend.cpp:110-131,183-306) for one camera stream, HOST buffers in and out.  The stages are
   the entry points above, chained on the context's stream with no host round trip in between; keyframe switching / dropping and
   list building stay with the caller (stereo_frontend.cpp:265-296). -----------------------------------------------------------*/
typedef struct {
  int32_t fast_trials;                 /* 6 (stereo_frontend.cpp:232); the first frame uses one less (:118) */
  int32_t search_radius, thr_mean, thr_std;      /* 8, 22, 10 (:989-1004, CPU build) */
  float max_reproj_error;              /* ui.max_reproj_error = 2 (:845-846) */
  int32_t use_block_matching;          /* 0: a disparity image comes with every frame (have_disp_img); 1: calcDisparityCpu from the right image */
  svs_pose_opt_params pose_opt;        /* PoseOptimizerParams(true, 2, 15) (:1061) */
  svs_stereo_params stereo;            /* cv::StereoBM state (:620-653); used if use_block_matching */
  int32_t n_levels;                    /* use_n_levels_in_frontent (:68): pyramid levels FAST and the matcher run on; the code default is 2, the shipped
                                          configurations set 3.  0 = 3 */
  int32_t num_max_points;              /* ui.num_max_points (:1000): neighbours' new-point lists are matched while 2 * observations < this.  0 = 300 */
  int32_t min_matches;                 /* matchAndTrack fails below this many observations (:1053).  0 = 20 */
  int32_t cuda_build;                  /* 0: the reference's CPU build (quarter-grid denseTrackingCpu, computeDensePointCloudCpu; search_radius 8);
                                          1: its SCAVISLAM_CUDA_SUPPORT build (full-resolution denseTrackingGpu on f32 pyramids, computeDensePointCloudGpu;
                                          the caller passes search_radius 4, :1043-1047) */
} svs_frontend_params;
typedef struct {
  double T_cur_from_actkey[12];        /* after dense tracking and calcFastMotionOnly */
  int32_t dense_passes;                /* fused H,b / chi2 sweeps of the dense tracker (-1: its workgroups could not synchronise) */
  int32_t n_points;                    /* candidate points matched against (= records in h_matches / h_gated) */
  int32_t n_matched;                   /* SVS_MATCH_OK records = obs_list.size() */
  int32_t tracking_ok;                 /* n_matched >= 20: what matchAndTrack returns (stereo_frontend.cpp:1053-1056) */
  svs_pose_opt_stats pose_stats;
  svs_point_stats point_stats;
} svs_frame_result;
typedef struct svs_frontend svs_frontend;
/* cam = level-0 stereo camera (w, h multiples of 16).  max_points: capacity of the candidate list (ap_map); max_keyframes: keyframe slots */
int svs_frontend_create(svs_ctx *ctx, const svs_cam *cam, const svs_frontend_params *prm, int max_points, int max_keyframes, svs_frontend **out);
/* the same for n_streams independent camera streams (one front end each: own previous frame, keyframes, candidate points, FAST thresholds) whose
   stages run as ONE launch per stage.  Stream 0 is the one the host-buffer entry points (first_frame / process_frame / ...) talk to; they need n_streams == 1 */
int svs_frontend_create_batch(svs_ctx *ctx, const svs_cam *cam, const svs_frontend_params *prm, int max_points, int max_keyframes, int n_streams,
                              svs_frontend **out);
int svs_frontend_destroy(svs_frontend *fe);
/* processFirstFrame: pyramid, disparity, FAST (fast_trials - 1), reference cloud at the identity.  h_right or h_disp per use_block_matching */
int svs_frontend_first_frame(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride, const float *h_disp, int dstride);
/* Frame::clone of the frame processed last into keyframe slot `slot` with its pose (keyframe_map entry + vertex_map pose) */
int svs_frontend_keep_keyframe(svs_frontend *fe, int slot, const double *T_kf_from_w);
int svs_frontend_keep_keyframe_of(svs_frontend *fe, int stream, int slot, const double *T_kf_from_w);
/* ... of ALL streams at once into slot `slot` of each; h_T_kf_from_w [n_streams][12] (one strided copy per level + one table upload) */
int svs_frontend_keep_keyframes(svs_frontend *fe, int slot, const double *h_T_kf_from_w);
/* ap_map: h_pts[i].kf_index = keyframe slot of the anchor (a slot svs_frontend_keep_keyframe has filled; < 0 = anchor frame not in keyframe_map,
   matcher.cpp:336-339); records [0, n_new_records) are the "new feature" candidates (:989-1030) */
int svs_frontend_set_candidates(svs_frontend *fe, const svs_candidate_point *h_pts, int n, int n_new_records);
/* the candidate lists of matchAndTrack (stereo_frontend.cpp:976-1050) in the order it walks them, for one stream: group 0 = newpoint_map[actkey_id],
   groups 1 .. n_groups-2 = newpoint_map[neighbour] in the order of active_vertex.strength_to_neighbors, group n_groups-1 = neighborhood_->point_list.
   h_group_end[g] = one past the last record of group g (n_groups >= 2, <= 64; h_group_end[n_groups-1] == n).  The neighbour lists behind the cut
   "2 * obs_list.size() < ui.num_max_points" come back with status SVS_MATCH_SKIPPED and take no part in the refinement or the gate. */
int svs_frontend_set_candidates_grouped(svs_frontend *fe, int stream, const svs_candidate_point *h_pts, int n, const int32_t *h_group_end, int n_groups);
/* the lists of ALL streams in one staged upload: h_pts = the streams' records back to back (h_n[b] of stream b), h_group_end [n_streams][n_groups] (the same
   number of groups for every stream; an absent list is an empty group) */
int svs_frontend_set_candidates_all(svs_frontend *fe, const svs_candidate_point *h_pts, const int32_t *h_n, const int32_t *h_group_end, int n_groups);
/* processFrame.  T_cur_from_actkey: the motion-model guess (in); T_actkey_from_w: pose of the active keyframe.  h_matches / h_gated:
   n_points records each (may be NULL).  Blocking, like the reference call. */
int svs_frontend_process_frame(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride, const float *h_disp,
                               int dstride, const double *T_cur_from_actkey, const double *T_actkey_from_w, svs_frame_result *out,
                               svs_match_result *h_matches, svs_gated_point *h_gated);
/* processFrame in two halves (stream 0, n_streams == 1): submit stages the images, enqueues upload + all stages + the download and returns; wait blocks
   and hands the results out (h_matches / h_gated only if asked for at submit).  In between the caller is free -- e.g. to prefetch the next frame. */
int svs_frontend_submit_frame(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride, const float *h_disp, int dstride,
                              const double *T_cur_from_actkey, const double *T_actkey_from_w, int want_matches, int want_gated);
int svs_frontend_wait_frame(svs_frontend *fe, svs_frame_result *out, svs_match_result *h_matches, svs_gated_point *h_gated);
/* upload the NEXT frame on a copy stream while the frame submitted last is being processed -- the reference's FrameData double buffer
   (frame_grabber.hpp:93-155: the grabber thread fills the next frame while the front end works on the current one).  The following
   first_frame / submit_frame / process_frame must pass NULL images. */
int svs_frontend_prefetch_frame(svs_frontend *fe, const uint8_t *h_left, int lstride, const uint8_t *h_right, int rstride, const float *h_disp, int dstride);
/* the pinned host buffers the NEXT frame is staged in (w x h each, contiguous; right image or disparity per use_block_matching): a frame grabber that writes
   its images straight into them and passes these pointers (stride = w) to the next first_frame / submit_frame / process_frame / prefetch_frame call saves the
   host-side copy (1.5 MB per 640 x 480 frame).  The pointers change with every such call (two staging sets) */
int svs_frontend_staging_view(svs_frontend *fe, uint8_t **h_left, uint8_t **h_right, float **h_disp);
/* ---- all streams at once, frames in DEVICE memory (camera DMA target, decoder output, another kernel's result) ---- */
typedef struct {                       /* images of all streams: stream b at + b * bstride (elements); unused members NULL */
  const uint8_t *d_left; int32_t lstride; size_t l_bstride;
  const uint8_t *d_right; int32_t rstride; size_t r_bstride;   /* use_block_matching */
  const float *d_disp; int32_t dstride; size_t d_bstride;      /* otherwise; read in place during the call, not copied */
  /* WHEN are these frames complete?  NULL: when the work enqueued on the context's stream before this call has run (a kernel or copy of the caller's on that
     stream may still be writing them at call time: the library reads them behind it, in stream order).  A hipEvent_t: when that event has fired -- the caller
     recorded it behind whatever produces the frames, on whichever stream.  Only then may the library read the frames on ANOTHER stream than the context's,
     which is what lets svs_frontend_process_frames build the pyramid of frame N + 1 beside the pose refinement of frame N ("fe_pipeline"). */
  void *ready_event;
} svs_frames_dev;
/* where the NEXT frames may be written in place (then pass in == NULL below): level-0 image, right image (NULL without block matching), disparity.
   Valid until the next first_frames / process_frames call, which moves on to other buffers */
int svs_frontend_input_view(svs_frontend *fe, uint8_t **d_left, int32_t *lstride, size_t *l_bstride, uint8_t **d_right, int32_t *rstride, size_t *r_bstride,
                            float **d_disp, int32_t *dstride, size_t *d_bstride);
/* processFirstFrame for every stream (blocking) */
int svs_frontend_first_frames(svs_frontend *fe, const svs_frames_dev *in);
/* processFrame for every stream: poses [n_streams][12] from the host; ASYNCHRONOUS on the context's stream, results stay on the device.
   The frames are read during the call chain, in place: they must stay untouched until the chain has run (svs_ctx_sync, or any blocking call of this front end),
   and in->ready_event says from when on they are valid (see svs_frames_dev). */
int svs_frontend_process_frames(svs_frontend *fe, const svs_frames_dev *in, const double *h_T_cur_from_actkey, const double *h_T_actkey_from_w);
/* blocking downloads after svs_frontend_process_frames: everything of one stream; refined poses [n_streams][12] + tracking flags of all (NULL = not wanted) */
int svs_frontend_results(svs_frontend *fe, int stream, svs_frame_result *out, svs_match_result *h_matches, svs_gated_point *h_gated);
int svs_frontend_poses(svs_frontend *fe, double *h_T_cur_from_actkey, int32_t *h_tracking_ok);
/* profiling: hipEvents between the stages of the following process_frame(s) / submit_frame calls (off by default: an event costs ~4 us of stream time).
   svs_frontend_stage_times (blocking): ms[SVS_FRONTEND_STAGES] of the last such call, in the order of the reference's per_mon_ stages
   (stereo_frontend.cpp:190-302): preprocess (upload or copy + pyramid), dense tracking, stereo, fast, match, pose refinement (calcFastMotionOnly),
   process points, dense point cloud */
#define SVS_FRONTEND_STAGES 8
int svs_frontend_set_timing(svs_frontend *fe, int on);
int svs_frontend_stage_times(svs_frontend *fe, float *ms);
/* blocking: the accept / reject record of the dense tracker's LM loop of one stream in the last call (one entry per chi2 evaluation: level, accepted,
   chi2 before / after); *n = records produced, of which the first min(*n, cap, 64) are stored */
int svs_frontend_dense_records(svs_frontend *fe, int stream, svs_dense_lm_record *h_rec, int cap, int32_t *n);
/* computeDensePointCloudCpu / Gpu again at a pose decided after the frame (keyframe switch, :277-281, :298-302); T_cur_from_actkey [n_streams][12] */
int svs_frontend_recompute_cloud(svs_frontend *fe, const double *T_cur_from_actkey);
/* device views of one stream (tests, chaining): level images of the frame processed last, strides, its disparity (the caller's buffer if it passed
   one), reference clouds (quarter grid, or full resolution in the CUDA build), the FastGrid object (shared by all streams: slot = stream) */
int svs_frontend_device_view(svs_frontend *fe, int stream, const uint8_t **d_pyr_last, int32_t *stride, const float **d_disp, const float **d_cloud,
                             svs_fast **fast);

/* ---- multi-GPU: the library-owned collective of the landmark-sharded back-end.  The reference has no
   distributed code; one process per GPU each creates a context and a communicator (RCCL, bound at run time) --------------*/
typedef struct { char bytes[128]; } svs_unique_id;      /* = ncclUniqueId */
typedef struct svs_comm svs_comm;
/* rank 0 obtains the id and hands it to the other ranks out of band (MPI, a TCP store, a file) */
int svs_comm_get_unique_id(svs_ctx *ctx, svs_unique_id *out);
/* collective over all ranks: ncclCommInitRank on the context's device */
int svs_comm_create(svs_ctx *ctx, const svs_unique_id *id, int rank, int world, svs_comm **out);
int svs_comm_destroy(svs_comm *c);
/* sum `count` doubles at d_buf in place across ranks, on the context's stream (asynchronous like every device entry point) */
int svs_comm_allreduce_f64(svs_comm *c, void *d_buf, size_t count);
int svs_comm_stats(svs_comm *c, int32_t *rank, int32_t *world, uint64_t *n_calls, uint64_t *n_doubles);
/* Second transport behind the same svs_comm handle (svs_comm_allreduce_f64 / svs_ba_set_comm work unchanged): a ONE-SHOT exchange over peer-mapped
   mailboxes (hipIpc + xGMI P2P writes) instead of RCCL's ring -- every rank pushes its vector into every peer's mailbox, waits for the N-1 arrivals and
   sums in rank order (bit-identical on all ranks).  Meant for the small, latency-bound messages of the sharded back end; world <= 16.
     1. every rank: svs_comm_create_p2p (mailbox of 2 x world x capacity_doubles; longer messages travel in pieces) -> its svs_ipc_handle
     2. all-gather the handles out of band (MPI, a TCP store, torch.distributed), then every rank: svs_comm_connect_p2p(c, handles[world])
   A peer that never arrives makes the result NaN and is counted (svs_comm_transport); destroy collectively (no rank may still be pushing).
   The mailbox is allocated fine-grained (hipExtMallocWithFlags; uncached, then plain memory as fall-backs -- svs_comm_transport says which): peers write it while
   the owner's kernel polls it.  hipIpcOpenMemHandle cannot open a handle inside the process that exported it: one process driving several GPUs cannot use this
   transport (one process per GPU, as everywhere in this library). */
typedef struct { char bytes[64]; } svs_ipc_handle;      /* = hipIpcMemHandle_t */
int svs_comm_create_p2p(svs_ctx *ctx, int rank, int world, size_t capacity_doubles, svs_comm **out, svs_ipc_handle *h_mine);
int svs_comm_connect_p2p(svs_comm *c, const svs_ipc_handle *h_all);
/* *kind = 0: RCCL; one-shot P2P with its mailbox in 1: fine-grained, 2: uncached, 3: plain (coarse-grained) device memory; *timeouts = reduce launches that gave
   up waiting for a peer (blocking; either may be NULL) */
int svs_comm_transport(svs_comm *c, int32_t *kind, uint32_t *timeouts);

/* ---- BA: replaces SlamGraph::optimize (slam_graph.hpp:457-462, slam_graph.cpp:312-355) -------*/
typedef struct svs_ba svs_ba;
/* all-reduce hook for landmark-sharded operation: sum `count` doubles at d_buf in place across
   ranks, on the ctx stream (RCCL via torch.distributed in scavislam_amd/backend.py) */
typedef int (*svs_allreduce_fn)(void *d_buf, size_t count, void *user);

int svs_ba_create(svs_ctx *ctx, svs_ba **out);
int svs_ba_destroy(svs_ba *ba);
/* copyDataToG2o (slam_graph.cpp:983-1032): poses of the double window, active points as psi
   (inverse depth), observation edges, pose-pose constraints.  Host arrays; edges may be in any
   order.  In sharded runs each rank passes only its landmarks' edges (point ids stay global
   0..L-1) and add_pose_terms = 1 on exactly one rank (constraints counted once). */
int svs_ba_set_problem(svs_ba *ba, int P, const double *h_poses, int L, const double *h_psi,
                       int E, const svs_ba_edge *h_edges, int C, const svs_ba_constraint *h_cons,
                       const svs_cam *cam, const svs_ba_params *prm, int add_pose_terms);
/* Persistent window -- the alternative to svs_ba_set_problem for a caller whose window evolves by about one
   keyframe per optimize(): the library keeps every observation it has been given on the device; a call brings the window's definition
   by the graph's own ids (frame ids and point ids come from one counter in the reference, stereo_frontend.cpp:826-831), the current
   values, the observations made SINCE THE LAST CALL and the window's pose-pose constraints.
     h_pose_ids [P] / h_poses [P][12]      the double window in block-row order (copyPosesToG2o, slam_graph.cpp:924-935)
     h_point_ids [L] / h_psi [L][3] / h_anchor_pose_ids [L]   the active points (addPointToG2o :907-922), anchor by frame id
     h_new_obs [n_new]                     svs_ba_edge records whose .point / .pose hold IDS (.anchor ignored): addObsToG2o's arguments
     h_cons [C]                            constraints whose .pose1 / .pose2 hold frame IDS (copyContraintsToG2o :937-981)
   The window's edges = all stored observations whose point is active, whose keyframe is in the window and whose point's anchor is in
   the window (exactly the filter of slam_graph.cpp:1001-1006).  State comes back through svs_ba_get_state in the order given here.
   svs_ba_window_reset forgets the stored observations. */
int svs_ba_window_update(svs_ba *ba, int P, const int32_t *h_pose_ids, const double *h_poses, int L, const int32_t *h_point_ids,
                         const double *h_psi, const int32_t *h_anchor_pose_ids, int n_new, const svs_ba_edge *h_new_obs, int C,
                         const svs_ba_constraint *h_cons, const svs_cam *cam, const svs_ba_params *prm);
int svs_ba_window_reset(svs_ba *ba);
/* drop the stored observations of keyframes that will never be part of a window again (marginalised / removed from the map): the store is otherwise
   grow-only, and every svs_ba_window_update filters and sorts all of it.  A failed svs_ba_window_update leaves the store as it found it. */
int svs_ba_window_forget_keyframes(svs_ba *ba, const int32_t *h_pose_ids, int n);
/* optimizer.optimize(num_iters) (slam_graph.cpp:346) incl. LM control flow; allreduce may be
   NULL (single GPU) */
int svs_ba_optimize(svs_ba *ba, svs_allreduce_fn allreduce, void *user, svs_ba_stats *stats);
/* throughput mode: n independent windows (each svs_ba created on its OWN context, i.e. its own stream), all enqueued before the
   first wait so that their kernels overlap on the device; stats[n] optional.  Single-GPU (no collectives). */
int svs_ba_optimize_batch(svs_ba *const *bas, int n, svs_ba_stats *stats);
/* restoreDataFromG2o (slam_graph.cpp:1035-1058): poses [P][12], psi [L][3] */
int svs_ba_get_state(svs_ba *ba, double *h_poses, double *h_psi);
/* landmark-sharded operation over a library-owned communicator: svs_ba_optimize(ba, NULL, NULL, stats) then all-reduces the
   packed reduced system and the two trial sums of every LM trial (32 doubles: each sum is kept in 16 partial slots) (and, once per problem, the co-visibility pattern) with
   ncclAllReduce on the ctx stream.  comm == NULL detaches.  A non-NULL `allreduce` argument of svs_ba_optimize takes precedence
   (test hook). */
int svs_ba_set_comm(svs_ba *ba, svs_comm *comm);
/* experiment / test switches of one optimizer (0 = default behaviour): "no_speculation", "one_front", "no_fused_solve",
   "no_lds_solve", "no_grid_solve", "no_tile_solve" (wide envelopes: the multi-workgroup Cholesky by block rows instead of the tile-resident one), "no_fused_cons", "debug" (1: phase timers, 2: Schur kernel timeline), "nw" (waves per Schur workgroup, 4..8),
   "p1" (rows of the reversed front), "group" (anchors dealt round-robin), "host_threads", "host_marshal" (svs_ba_set_problem: 0 = device marshalling from
   8k edges, 1 = always host, 2 = always device), "no_graph" (1: svs_ba_optimize enqueues kernel by kernel instead of replaying its recorded graph,
   see svs_ba_graph_stats), "no_lds_panel" (1: the fused solve's back-substitution panel through global memory, rounds 3-5).  The environment (SVS_BA_*,
   SVS_HOST_THREADS) only supplies the initial values, read once by svs_ba_create; values are clamped to their valid ranges. */
int svs_ba_set_option(svs_ba *ba, const char *name, int value);
/* building blocks exposed for parity tests and profiling */
int svs_ba_reset_state(svs_ba *ba, const double *h_poses, const double *h_psi);
int svs_ba_reduced_system(svs_ba *ba, double lambda, double *h_Hred /* (6P)^2 full sym */,
                          double *h_bred /* 6P */, double *h_chi2);
/* what the last svs_ba_set_problem led to: solve_kind 0 = global-memory blocked Cholesky, 1 = LDS-window pipeline, 2 = fused
   register-resident elimination (one front), 3 = the same with two fronts, 4 = multi-workgroup blocked Cholesky (one block row per step, trailing matrix in global memory; option "no_tile_solve"), 5 = multi-workgroup
   tile-resident blocked Cholesky (24 x 24 tiles owned by workgroups in LDS: the default for wide envelopes); envelope_rows = widest filled block row of the reduced
   system (+1); wave chunks of the Schur kernel; landmarks with more than 64 observations */
int svs_ba_info(svs_ba *ba, int32_t *solve_kind, int32_t *envelope_rows, int32_t *n_chunks, int32_t *n_wide);
/* the solve's pose order (what LinearSolverCSparse's block ordering is to the reference, slam_graph.cpp:1063-1074): envelope_rows above is that of the order the solve
   runs in; *envelope_rows_caller_order the one of the caller's pose order; *ordered = 1 when a fill-reducing order (reverse Cuthill-McKee on the pose block graph) is
   in use -- taken when it narrows the filled envelope by a quarter or more, e.g. a loop closure on a chain of keyframes; h_perm (optional, [P]): solver row k is
   the caller's pose h_perm[k] (the identity when not ordered).  Option "no_order" (svs_ba_set_option) keeps the caller's order. */
int svs_ba_order_info(svs_ba *ba, int32_t *ordered, int32_t *envelope_rows_caller_order, int32_t *h_perm);
/* profiling: bracket the Schur, solve and back-substitution kernels of every LM trial with hipEvents on the ctx stream.
   Off by default: each event record costs ~4 us of stream time (12 per optimize() = 15 % of it at 50 KF / 20k). */
int svs_ba_set_timing(svs_ba *ba, int on);
/* last-call timing of the dominant kernels, ms (zeros unless svs_ba_set_timing(ba, 1)) */
int svs_ba_kernel_times(svs_ba *ba, float *reduce_ms, float *solve_ms, float *backsub_ms,
                        int32_t *n_reduce_launches);
/* How svs_ba_optimize enqueues its work.  The all-accepted optimize of a resident window (SlamGraph::optimize, slam_graph.cpp:312-355: num_iters x { build, solve,
   update, compare }) has a fixed launch topology; the library records it once per problem layout (when a layout is seen for the second time) as a HIP graph and replays it with one launch per call (the first rejected
   trial falls back to the host-driven loop, as before).  *launches: optimizes replayed from a graph so far; *captures: recordings made (a new one whenever the layout,
   the buffers or the parameters change).  Option "no_graph" (svs_ba_set_option) keeps the kernel-by-kernel path.  Not used with an all-reduce callback / communicator,
   with svs_ba_set_timing, or with the multi-workgroup solves of wide envelopes. */
int svs_ba_graph_stats(svs_ba *ba, int64_t *launches, int64_t *captures);

#ifdef __cplusplus
}
#endif
#endif
