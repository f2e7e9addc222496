/*
 * svs_oracle.h -- CPU restatement ("oracle") of ScaViSLAM's two data-parallel hot paths.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under scavislam_amd/ may include, link, dlopen or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker / the timed CPU baseline, never as the product path.
 *
 * PARITY: PINNED against the reference's OWN code wherever that code can be compiled, UNPINNED for the
 * third-party arithmetic.  The reference (strasdat/ScaViSLAM) ships no tests, golden vectors or fixtures and
 * cannot be built as a whole (OpenCV 2.4.2, g2o, Sophus a621ff, VisionTools, Pangolin, Eigen, SuiteSparse,
 * Boost are not on disk).  oracle/Makefile therefore compiles the reference files / functions of the hot
 * path one by one, FROM WHERE THEY LIE under /root/reference, against stand-in headers for the type names of
 * the absent libraries (oracle/ref_shim/fake; their algebra is handed to this oracle's own helpers) and, for
 * the CUDA files, through a host emulation of the CUDA execution model.  The libraries under oracle/_ref:
 *   libsvs_ref_gpu.so       gpu/dense_tracking.{cuh,cu}: per-pixel helpers, the three kernels, GpuTracker
 *   libsvs_ref_densegpu.so  dense_tracking.cpp CUDA branch: denseTrackingGpu (LM loop), computeDensePointCloudGpu
 *   libsvs_ref_dense.so     dense_tracking.cpp CPU branch: denseTrackingCpu, computeDensePointCloudCpu, maths_utils.cpp
 *   libsvs_ref_qt.so        quadtree.h;   libsvs_ref_fastgrid.so  fast_grid.cpp (FAST-9/16 itself hooked to this oracle)
 *   libsvs_ref_matcher.so   matcher.cpp / matcher-impl.cpp: ZNSSD, warpAffinve, matchCandidates, match()
 *   libsvs_ref_pose.so      pose_optimizer.h: calcFastMotionOnly;   libsvs_ref_gate.so  processMatchedPoints, initialize, computeFastCorners
 *   libsvs_ref_track.so     stereo_frontend.cpp matchAndTrack = matcher + pose optimiser chained as the front end chains them
 *   libsvs_ref_frame.so     stereo_frontend.cpp processFrame with everything it calls on the hot path, one translation unit
 *   libsvs_ref_frame_cuda.so  the same with SCAVISLAM_CUDA_SUPPORT: the reference's CUDA build of the per-frame path
 *   libsvs_ref_edges.so     g2o_types/anchored_points.{h,cpp}: edge errors, Jacobians, oplus
 *   libsvs_ref_slamgraph.so slam_graph.cpp / -impl.cpp: optimize, copyDataToG2o and friends, into a recording g2o stand-in
 * tests/test_ref_pin_cpu.py holds this restatement BIT-EQUAL to every one of them.  What stays UNPINNED is
 * what is not under /root/reference: the arithmetic of FAST / pyrDown / Sobel / convertTo / StereoBM
 * (OpenCV 2.4.2), of the LM / Schur / Huber solve (g2o, unpinned fork), Sophus' SE3 exp, Eigen's ldlt and
 * VisionTools' pinhole maps.  This file restates
 *   - the reference's own code, citing file:line of /root/reference/scavislam/..., and
 *   - the published algorithm of the third-party calls at the reference's call sites
 *     (SURVEY.md Appendix A), marked "[3rd-party: ...]";
 * the latter is cross-checked by an independent NumPy/SciPy float64 model (tests/np_model.py) and by
 * analytic properties (numeric-vs-analytic Jacobians, FAST monotonicity, Schur solution
 * satisfies the full normal equations).
 *
 * All functions are plain C, single-threaded, no global mutable state.
 */
#ifndef SVS_ORACLE_H
#define SVS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- shared POD types (identical layout to include/scavislam_hip.h) ---------------------- */

/* per-level stereo camera: frame_grabber-impl.cpp:48-60 (f/2^l, c/2^l, size/2^l, b*2^l) */
typedef struct {
  double f, cx, cy, b;
  int32_t w, h;
} svs_cam;

/* keyframes.h:31-44 FastGridCell + fast_grid.cpp:23-58 FastGrid members */
#define SVS_MAX_CELLS 64
typedef struct {
  int32_t gx, gy;          /* grid_size.width/height */
  int32_t cell_w, cell_h;  /* img / grid, integer division */
  int32_t min_inner, min_outer, max_inner, max_outer;
  int32_t fast_min, fast_max;
  int32_t thr[SVS_MAX_CELLS]; /* per-cell persistent fast_thr, row-major [j*gx+i] */
} svs_fastgrid;

/* data_structures.h:37-69 CandidatePoint<3>; anchor_id resolved to an index by the caller */
typedef struct {
  double xyz_anchor[3];
  double anchor_obs_pyr[3];
  int32_t anchor_level;
  int32_t kf_index;   /* index into the keyframe table; <0 => anchor not in vertex_map */
  int32_t point_id;
  int32_t pad_;
} svs_candidate_point;

typedef struct {
  double T_anchor_from_w[12];   /* 3x4 row-major */
  const uint8_t *pyr[3];
  int32_t stride[3];
  int32_t pad_;
} svs_keyframe;

/* status codes of one candidate point through GuidedMatcher::match (matcher.cpp:312-398) */
enum {
  SVS_MATCH_OK = 0,            /* observation appended */
  SVS_MATCH_NO_ANCHOR = 1,     /* computePrediction: anchor not in vertex_map */
  SVS_MATCH_BORDER = 2,        /* anchor obs within HALFBOX of border */
  SVS_MATCH_DEPTH = 3,         /* inverse-depth ratio > 3 */
  SVS_MATCH_TEXTURE = 4,       /* sumA^2 - sumAA < thr_std^2*64 gate (sic) */
  SVS_MATCH_NONE = 5,          /* no candidate below thr_mean^2*64 */
  SVS_MATCH_NO_DISP = 6        /* best match found but disparity <= 0 */
};

typedef struct {
  int32_t status;
  int32_t u, v;       /* best uv_pyr (valid for status OK / NO_DISP) */
  int32_t znssd;      /* best score (min_dist) */
  double obs[3];      /* (u,v,u-d)*2^level, valid for OK */
  double xyz_actkey[3];
} svs_match_result;

/* 27 accumulators of one dense-tracking pass: GpuTrackingData (gpu/dense_tracking.cuh:28-277)
   hessian = 21 packed upper-by-column (0,0),(0,1),(1,1),(0,2)...; then J^T r (6). */
typedef struct {
  double H[21];
  double b[6];
  double chi2;
  int64_t n_valid;
} svs_dense_sums;

/* BA problem (SlamGraph::optimize, slam_graph.cpp:312-355 + copyDataToG2o :983-1032) */
typedef struct {
  double obs[3];        /* feat.center (u,v,u_r) level-0 px */
  double info[3];       /* diag(Lambda): 4^-lvl, 4^-lvl, 0.333^2  (slam_graph.cpp:1010-1015) */
  int32_t point;        /* landmark index 0..L-1 */
  int32_t pose;         /* observer pose index 0..P-1 */
  int32_t anchor;       /* anchor pose index 0..P-1 */
  int32_t pad_;
} svs_ba_edge;          /* 64 bytes */

typedef struct {
  double T_21[12];      /* measurement T_2_from_1, 3x4 row-major */
  double info[36];      /* 6x6 Lambda row-major */
  int32_t pose1, pose2; /* vertices[0], vertices[1] */
} svs_ba_constraint;

typedef struct {
  int32_t num_iters;        /* OptParams.num_iters (2) */
  int32_t use_robust;       /* OptParams.use_robust_kernel */
  double huber_delta;       /* g2o default 1.0 (OptParams.huber_kernel_width is dead code) */
  double lambda_init;       /* 50 (slam_graph.cpp:338) */
  int32_t max_trials;       /* 5 (slam_graph.cpp:1073) */
  int32_t self_edge_mode;   /* 0 = G2O_LITERAL (+M on anchor diag), 1 = EXACT (SURVEY B-7) */
} svs_ba_params;

typedef struct {
  int32_t iterations;       /* outer iterations executed */
  int32_t trials;           /* total LM trials */
  int32_t accepted;         /* accepted steps */
  int32_t terminated;       /* 1 if solver returned Terminate */
  double chi2_init, chi2_final, lambda_final;
} svs_ba_stats;

/* ---- image ops ----------------------------------------------------------------------------*/
/* [3rd-party: OpenCV 2.4.2 pyrDown u8, called at frame_grabber.cpp:290] */
void svs_ref_pyr_down_u8(const uint8_t *src, int w, int h, int sstride,
                         uint8_t *dst, int dstride);
/* [3rd-party: convertTo(CV_32F,1/255.) + Sobel(ksize=1), frame_grabber.cpp:315-333] */
void svs_ref_convert_sobel(const uint8_t *src, int w, int h, int sstride,
                           float *img, float *dx, float *dy, int fstride);

/* ---- FAST ---------------------------------------------------------------------------------*/
/* [3rd-party: cv::FastFeatureDetector(thr,false).detect, fast_grid.cpp:72-73,104-105] */
int svs_ref_fast9_16(const uint8_t *img, int w, int h, int stride, int thr,
                     int16_t *xy, int cap);
/* per-pixel score = max t s.t. corner at t (-1: never); SURVEY A.1 monotone form */
int svs_ref_fast_score(const uint8_t *img, int stride, int x, int y);
/* fast_grid.cpp:23-58 */
void svs_ref_fastgrid_init(svs_fastgrid *g, int img_w, int img_h, int n_per_cell,
                           int boundary, int fast_thr, int gx, int gy,
                           int fast_min, int fast_max);
/* stereo_frontend.cpp:73-88: per-level grid parameters */
void svs_ref_fastgrid_init_level(svs_fastgrid *g, int img_w, int img_h, int level);
/* fast_grid.cpp:86-152.  corners: (x,y) level coords in reference insertion order;
   cell_count[c] = corners of cell c; emit_thr[c] = threshold of the last executed detection */
int svs_ref_fastgrid_detect_adaptively(svs_fastgrid *g, const uint8_t *img, int stride,
                                       int trials, int16_t *xy, int cap,
                                       int32_t *cell_count, int32_t *emit_thr);
/* fast_grid.cpp:60-83 */
int svs_ref_fastgrid_detect(const svs_fastgrid *g, const uint8_t *img, int stride,
                            int16_t *xy, int cap, int32_t *cell_count);

/* ---- quadtree (quadtree.h:510-710) --------------------------------------------------------*/
typedef struct svs_ref_qt svs_ref_qt;
svs_ref_qt *svs_ref_qt_create(double x, double y, double w, double h, double delta);
void svs_ref_qt_destroy(svs_ref_qt *);
int svs_ref_qt_insert(svs_ref_qt *, double px, double py, int content);
/* the insertion loop of fast_grid.cpp:143-149: content = index within the cell */
void svs_ref_qt_insert_corners(svs_ref_qt *, const int16_t *xy, const int32_t *cell_count, int n_cells);
/* window query in the reference's DFS order; returns count, fills (x,y,content) triples */
int svs_ref_qt_query(const svs_ref_qt *, double wx, double wy, double ww, double wh,
                     int32_t *out_xyc, int cap);

/* ---- matcher (matcher.cpp) ----------------------------------------------------------------*/
/* matcher.cpp:403-458 warpAffinve -> 10x10 u8 (row=iy, col=ix) */
void svs_ref_warp_affine(const uint8_t *frame, int stride, const double T_c2_from_c1[12],
                         double depth, const double key_uv[2], const svs_cam *cam,
                         int halfpatch, uint8_t *patch /* (2*halfpatch)^2 */);
/* matcher.cpp:42-74 */
int svs_ref_znssd(const uint8_t key[64], const uint8_t cur[64], int sumA, int sumAA);
/* matcher.cpp:144-181 matchCandidates over (x, y, content) triples in query order; out = {min_dist, index, u, v} */
void svs_ref_match_candidates(const uint8_t *cur_img, int cur_stride, const svs_cam *cam, const int32_t *cand_xyc, int nc,
                              const uint8_t *key, int sumA, int sumAA, int init_dist, int *out);
/* GuidedMatcher<StereoCamera>::match for n points; corners per level given as quadtrees */
void svs_ref_match(const svs_keyframe *kfs, int n_kf,
                   const double T_cur_from_actkey[12], const double T_actkey_from_w[12],
                   const uint8_t *const cur_pyr[3], const int cur_stride[3],
                   const float *disp, int disp_stride,
                   svs_ref_qt *const feature_tree[3], const svs_cam cam_vec[3],
                   const svs_candidate_point *pts, int n, int search_radius,
                   int thr_mean, int thr_std, svs_match_result *out);

/* ---- dense tracking (dense_tracking.cpp) --------------------------------------------------*/
/* one pass of the CPU-path loop body (dense_tracking.cpp:278-331 / :229-261):
   cloud = ref_dense_points (w/4 x h/4 float4), prev_u8 = prev pyr level image,
   cur/dx/dy = f32 level images.  do_jac=0 => chi2 only. float chi2 accumulated serially. */
void svs_ref_dense_pass_cpu(const float *cloud, int cw, int ch,
                            const uint8_t *prev_u8, int pstride,
                            const float *cur, const float *dx, const float *dy, int fstride,
                            const svs_cam *cam, const double T[12], int do_jac,
                            svs_dense_sums *out, float *residual_img /* cw*ch*4 or NULL */);
/* whole denseTrackingCpu (dense_tracking.cpp:222-391); T in/out; returns #passes executed */
int svs_ref_dense_tracking_cpu(const float *const cloud[3],
                               const uint8_t *const prev_u8[3], const int pstride[3],
                               const float *const cur[3], const float *const dx[3],
                               const float *const dy[3], const int fstride[3],
                               const svs_cam cam_vec[3], double T[12]);
/* same + residual_img[level] (cw*ch*4 floats each) as left behind by the last H,b pass of every level */
int svs_ref_dense_tracking_cpu_rimg(const float *const cloud[3], const uint8_t *const prev_u8[3],
                                    const int pstride[3], const float *const cur[3],
                                    const float *const dx[3], const float *const dy[3],
                                    const int fstride[3], const svs_cam cam_vec[3], double *T,
                                    float *const rimg[3]);
int svs_ref_dense_tracking_cpu_rec(const float *const cloud[3], const uint8_t *const prev_u8[3],
                                   const int pstride[3], const float *const cur[3],
                                   const float *const dx[3], const float *const dy[3],
                                   const int fstride[3], const svs_cam cam_vec[3], double *T,
                                   float *const rimg[3], double *rec, int rec_cap, int *n_rec);
/* gpu/dense_tracking.cu:495-541 residualImage_kernel; rimg has the cloud's stride */
void svs_ref_residual_image_full(const float *cloud, int w, int h, int stride4, const float *prev,
                                 const float *cur, int stride_f, float f, float cx, float cy,
                                 const float T34_colmajor[12], float *rimg);
/* dense_tracking.cpp:393-423 computeDensePointCloudCpu for one level */
void svs_ref_pointcloud_cpu(const float *disp, int disp_stride, const svs_cam *cam, int level,
                            const double T_cur_from_actkey[12], float *cloud /* (w/4)*(h/4)*4 */);
/* full-resolution f32 variant = gpu/dense_tracking.cu:172-263,376-453 semantics with manual
   4-tap f32 bilinear (maths_utils.cpp:46-65) instead of the texture unit */
void svs_ref_dense_pass_full(const float *cloud, int w, int h, int stride4,
                             const float *prev, const float *cur, const float *dx,
                             const float *dy, int fstride, float f, float cx, float cy,
                             const float T34_colmajor[12], int do_jac, svs_dense_sums *out);
/* same with the summation mode explicit: 0 = f64 sums (SVS_SUM_F64), 1 = the reference's f32 block-tree + sequential host
   sum, bit-comparable with oracle/_ref (SVS_SUM_F32_TREE) */
void svs_ref_dense_pass_full_ex(const float *cloud, int w, int h, int stride4,
                                const float *prev, const float *cur, const float *dx,
                                const float *dy, int fstride, float f, float cx, float cy,
                                const float T34_colmajor[12], int do_jac, int sum_mode, svs_dense_sums *out);
void svs_ref_dense_pixel_terms_full(const float *cloud, int w, int h, int stride4, const float *prev, const float *cur,
                                    const float *dx, const float *dy, int fstride, float f, float cx, float cy,
                                    const float T34_colmajor[12], float *out8);
/* DenseTracker::denseTrackingGpu (dense_tracking.cpp:60-193); see vision.c */
int svs_ref_dense_tracking_gpu(const float *const cloud[3], const int stride4[3], const float *const prev[3],
                               const float *const cur[3], const float *const dx[3], const float *const dy[3],
                               const int fstride[3], const int w[3], const int h[3], const double f[3],
                               const double cx[3], const double cy[3], double T[12], int sum_mode, double *rec, int rec_cap,
                               int *n_rec, double *T_jac);
int svs_ref_dense_tracking_gpu_forced(const float *const cloud[3], const int stride4[3], const float *const prev[3],
                               const float *const cur[3], const float *const dx[3], const float *const dy[3],
                               const int fstride[3], const int w[3], const int h[3], const double f[3],
                               const double cx[3], const double cy[3], double T[12], int sum_mode, double *rec, int rec_cap,
                               int *n_rec, double *T_jac, const int *force, int n_force);
/* FrameGrabber::preprocessing, CUDA build (frame_grabber.cpp:291-313,102-115); OpenCV gpu semantics ASSUMED, see vision.c */
void svs_ref_pyr_down_f32(const float *src, int w, int h, int sstride, float *dst, int dstride);
void svs_ref_deriv_replicate(const float *img, int w, int h, int stride, float *dx, float *dy, int dstride);
void svs_ref_convert_f32(const uint8_t *src, int w, int h, int sstride, float *dst, int dstride);
/* gpu/dense_tracking.cu:82-122 pointcloud_kernel semantics (incl. the unscaled-row quirk) */
void svs_ref_pointcloud_full(const float TQ_colmajor[16], const float *disp, int w, int h,
                             int stride_in, int stride_out, int factor, float *cloud);

/* ---- SE3 (Sophus a621ff semantics) -------------------------------------------*/
void svs_ref_se3_exp(const double x[6], double T[12]);
void svs_ref_se3_log(const double T[12], double x[6]);
void svs_ref_se3_mul(const double A[12], const double B[12], double C[12]);
void svs_ref_se3_inv(const double A[12], double B[12]);

/* ---- BA (g2o semantics) ------------------------------------------------------*/
/* anchored_points.cpp:148-189: error + 3 Jacobians (row-major 3x3, 3x6, 3x6) of one edge */
void svs_ref_edge_psi2uvu(const double psi[3], const double T_obs[12], const double T_anc[12],
                          const double obs[3], const svs_cam *cam, double err[3],
                          double J_psi[9], double J_obs[18], double J_anc[18]);
/* anchored_points.cpp:207-235: error + 2 Jacobians (6x6 row-major) */
void svs_ref_edge_se3(const double T_21[12], const double T1[12], const double T2[12],
                      double err[6], double J1[36], double J2[36]);
/* robust chi2 (sum rho_0) of the whole problem at the given state */
double svs_ref_ba_chi2(int P, const double *poses, int L, const double *psi, int E,
                       const svs_ba_edge *edges, int C, const svs_ba_constraint *cons,
                       const svs_cam *cam, const svs_ba_params *prm);
/* build reduced system at (poses,psi) with damping lambda: Hred (6P x 6P row-major, full
   symmetric), bred (6P); returns 0.  Used by tests to check device partial sums. */
int svs_ref_ba_reduced_system(int P, const double *poses, int L, const double *psi, int E,
                              const svs_ba_edge *edges, int C, const svs_ba_constraint *cons,
                              const svs_cam *cam, const svs_ba_params *prm, double lambda,
                              double *Hred, double *bred);
/* full optimize: poses [P][12] and psi [L][3] in/out */
/* same system from `threads` host threads (context number for the benchmark only; landmark ranges per thread) */
int svs_ref_ba_reduced_system_mt(int threads, int P, const double *poses, int L, const double *psi, int E,
                                 const svs_ba_edge *edges, int C, const svs_ba_constraint *cons,
                                 const svs_cam *cam, const svs_ba_params *prm, double lambda,
                                 double *Hred, double *bred);
int svs_ref_ba_optimize(int P, double *poses, int L, double *psi, int E,
                        const svs_ba_edge *edges, int C, const svs_ba_constraint *cons,
                        const svs_cam *cam, const svs_ba_params *prm, svs_ba_stats *stats);

/* ---- motion-only pose refinement: PoseOptimizer::calcFastMotionOnly, pose_optimizer.h:134-298 -------------- */
typedef struct {
  int32_t robust_kernel;   /* PoseOptimizerParams(true, 2, 15) at stereo_frontend.cpp:1061 */
  int32_t num_iter;
  double kernel_param;
  double initial_mu;       /* -1 => tau * max diag(J^T J) */
  double tau;              /* 1e-5 (pose_optimizer.h:50) */
} svs_pose_opt_params;
typedef struct {           /* OptimizerStatistics, pose_optimizer.h:59-98 */
  double initial_chi2, chi2, max_err;
  int32_t num_obs;
  int32_t status;          /* 0 ok, 1 empty observation list (the reference asserts), 2 residual became NaN (the reference throws) */
} svs_pose_opt_stats;
int svs_ref_motion_only(const svs_match_result *res, int n, const svs_cam *cam, const svs_pose_opt_params *prm, double *T_io,
                        svs_pose_opt_stats *st);

/* ---- StereoFrontend::processMatchedPoints (stereo_frontend.cpp:834-974): reprojection gate at the refined pose,
   PointStatistics counters, pyramid-level positions, track-length sum; serial, in obs_list order ---------------- */
typedef struct {
  int32_t accepted, is_new;
  double uv_pyr[2];
  double curkey_uv_pyr[2];
} svs_gated_point;
typedef struct {
  int32_t num_points_grid2x2[4];
  int32_t num_points_grid3x3[9];
  int32_t num_matched_points[3];
  int32_t num_track_points;
  int32_t num_obs;
  int32_t pad_[2];
  double sum_track_length;
} svs_point_stats;
void svs_ref_process_matched_points(const svs_match_result *res, const svs_candidate_point *pts, int n, int n_new_records,
                                    const svs_cam *cam, const double *T, float max_reproj_error, svs_gated_point *gated,
                                    svs_point_stats *stats);

/* ---- stereo block matching: cv::StereoBM as configured at stereo_frontend.cpp:620-653 (oracle/stereo.c) ---- */
typedef struct {
  int32_t prefilter_cap;      /* 31 */
  int32_t sad_window;         /* 7 */
  int32_t min_disparity;      /* 0 */
  int32_t num_disparities;    /* 32 (num_disp16 * 16) */
  int32_t texture_threshold;  /* 10 */
  int32_t uniqueness_ratio;   /* 15 */
  int32_t speckle_window;     /* 100 */
  int32_t speckle_range;      /* 32 */
  int32_t disp12_max_diff;    /* 1 */
} svs_stereo_params;
void svs_ref_stereo_prefilter_xsobel(const uint8_t *src, int w, int h, int stride, int cap, uint8_t *dst);
void svs_ref_stereo_bm_core(const uint8_t *lp, const uint8_t *rp, int w, int h, const svs_stereo_params *p, int16_t *disp16, int32_t *cost);
void svs_ref_stereo_validate(int16_t *disp16, const int32_t *cost, int w, int h, const svs_stereo_params *p);
void svs_ref_stereo_filter_speckles(int16_t *disp16, int w, int h, int new_val, int max_size, int max_diff);
void svs_ref_stereo_bm(const uint8_t *left, const uint8_t *right, int w, int h, int stride, const svs_stereo_params *p, float *disp, int dstride);

#ifdef __cplusplus
}
#endif
#endif
