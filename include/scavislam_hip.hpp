// scavislam_hip.hpp -- C++ adaptor classes over the C ABI (scavislam_hip.h) that keep the
// reference's call surfaces (method names, argument meaning, blocking semantics, bool/void
// returns, no exceptions across the boundary -- README:295 house style), so that they can replace
// the reference classes inside stereo_slam where OpenCV/Eigen/Sophus exist.  This header itself
// depends on nothing but the C ABI and the STL: images are (pointer, w, h, stride) views, poses
// are 3x4 row-major double[12].  INTEGRATION.md shows the 10-line glue from cv::Mat / Sophus::SE3.
//
//   FastGrid        <- scavislam/fast_grid.h:27-63
//   GuidedMatcher   <- scavislam/matcher.hpp:62-186
//   DenseTracker    <- scavislam/dense_tracking.h:53-97  (CPU-path semantics, the parity target)
//   GpuTracker / DenseTrackerGpu <- gpu/dense_tracking.cuh:281-342, DenseTracker::denseTrackingGpu (dense_tracking.cpp:60-215)
//   StereoFrontend  <- StereoFrontend::processFrame / processFirstFrame, scavislam/stereo_frontend.h:88-95 (one call per frame)
//   SlamGraphBA     <- SlamGraph::optimize, scavislam/slam_graph.hpp:457-462
//   Communicator    <- (no reference counterpart) landmark-sharded optimize over RCCL, SURVEY.md 8e
//
// One svs_ctx per calling thread (front-end on main, re-registration matcher/FAST + optimize on
// the backend thread, backend.cpp:452-469,738,763): no static scratch like matcher.cpp:36-38.
#ifndef SCAVISLAM_HIP_HPP
#define SCAVISLAM_HIP_HPP

#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

#include "scavislam_hip.h"

namespace scavislam_hip {

struct Image8 { const uint8_t *data; int w, h, stride; };
struct ImageF { const float *data; int w, h, stride; };
struct Corner { int16_t x, y; };

class Context {
 public:
  explicit Context(int device = 0, void *hip_stream = nullptr) : ctx_(nullptr) { ok_ = svs_ctx_create(device, hip_stream, &ctx_) == SVS_OK; }
  ~Context() { if (ctx_) svs_ctx_destroy(ctx_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  bool ok() const { return ok_; }
  svs_ctx *get() const { return ctx_; }
  const char *error() const { return svs_last_error(ctx_); }
  bool check(int rc) const { if (rc != SVS_OK) std::fprintf(stderr, "scavislam_hip: status %d: %s\n", rc, error()); return rc == SVS_OK; }

 private:
  svs_ctx *ctx_;
  bool ok_;
};

// RAII device buffer
template <class T>
class DeviceBuffer {
 public:
  DeviceBuffer() : ctx_(nullptr), p_(nullptr), n_(0) {}
  DeviceBuffer(const Context &c, size_t n) : ctx_(c.get()), p_(nullptr), n_(n) { void *p = nullptr; if (svs_malloc(ctx_, n * sizeof(T), &p) == SVS_OK) p_ = (T *)p; }
  ~DeviceBuffer() { if (p_) svs_free(ctx_, p_); }
  DeviceBuffer(DeviceBuffer &&o) noexcept : ctx_(o.ctx_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
  DeviceBuffer &operator=(DeviceBuffer &&o) noexcept { if (this != &o) { if (p_) svs_free(ctx_, p_); ctx_ = o.ctx_; p_ = o.p_; n_ = o.n_; o.p_ = nullptr; } return *this; }
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  T *get() const { return p_; }
  size_t size() const { return n_; }
  bool upload(const T *h, size_t n) { return svs_memcpy_h2d(ctx_, p_, h, n * sizeof(T)) == SVS_OK; }
  bool download(T *h, size_t n) const { return svs_memcpy_d2h(ctx_, h, p_, n * sizeof(T)) == SVS_OK; }

 private:
  svs_ctx *ctx_;
  T *p_;
  size_t n_;
};

// Device-resident u8 pyramid of one frame (Frame::pyr, keyframes.h:85) + disparity (Frame::disp)
class FrameDev {
 public:
  FrameDev(const Context &c, int w, int h) : ctx_(c) {
    for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l) {
      w_[l] = w >> l; h_[l] = h >> l; stride_[l] = (w_[l] + 63) / 64 * 64;
      pyr_[l] = DeviceBuffer<uint8_t>(c, (size_t)stride_[l] * h_[l]);
    }
    disp_ = DeviceBuffer<float>(c, (size_t)stride_[0] * h_[0]);
    for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l) {     // pyr_float32 / _dx / _dy of the CPU path (frame_grabber.cpp:315-333)
      f32_[l] = DeviceBuffer<float>(c, (size_t)stride_[l] * h_[l]);
      dx_[l] = DeviceBuffer<float>(c, (size_t)stride_[l] * h_[l]);
      dy_[l] = DeviceBuffer<float>(c, (size_t)stride_[l] * h_[l]);
    }
  }
  // FrameGrabber::preprocessing (frame_grabber.cpp:285-336): upload level 0, build the pyramid
  bool preprocessing(const Image8 &left) {
    std::vector<uint8_t> tmp((size_t)stride_[0] * h_[0]);
    for (int y = 0; y < h_[0]; ++y) std::memcpy(&tmp[(size_t)y * stride_[0]], left.data + (size_t)y * left.stride, (size_t)w_[0]);
    if (!pyr_[0].upload(tmp.data(), tmp.size())) return false;
    for (int l = 1; l < SVS_NUM_PYR_LEVELS; ++l)
      if (!ctx_.check(svs_pyr_down_u8(ctx_.get(), pyr_[l - 1].get(), w_[l - 1], h_[l - 1], stride_[l - 1], 0, pyr_[l].get(), stride_[l], 0, 1))) return false;
    for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l)
      if (!ctx_.check(svs_convert_sobel_f32(ctx_.get(), pyr_[l].get(), w_[l], h_[l], stride_[l], 0, f32_[l].get(), dx_[l].get(), dy_[l].get(), stride_[l], 0, 1))) return false;
    return true;
  }
  bool setDisparity(const ImageF &d) {
    std::vector<float> tmp((size_t)stride_[0] * h_[0]);
    for (int y = 0; y < h_[0]; ++y) std::memcpy(&tmp[(size_t)y * stride_[0]], d.data + (size_t)y * d.stride, sizeof(float) * (size_t)w_[0]);
    return disp_.upload(tmp.data(), tmp.size());
  }
  bool getDisparity(std::vector<float> *d) const {      // row-major w x h, stride removed
    std::vector<float> tmp((size_t)stride_[0] * h_[0]);
    if (!disp_.download(tmp.data(), tmp.size())) return false;
    d->resize((size_t)w_[0] * h_[0]);
    for (int y = 0; y < h_[0]; ++y) std::memcpy(&(*d)[(size_t)y * w_[0]], &tmp[(size_t)y * stride_[0]], sizeof(float) * (size_t)w_[0]);
    return true;
  }
  const uint8_t *pyr(int l) const { return pyr_[l].get(); }
  const float *disp() const { return disp_.get(); }
  const float *f32(int l) const { return f32_[l].get(); }
  const float *dx(int l) const { return dx_[l].get(); }
  const float *dy(int l) const { return dy_[l].get(); }
  int w(int l) const { return w_[l]; }
  int h(int l) const { return h_[l]; }
  int stride(int l) const { return stride_[l]; }

 private:
  const Context &ctx_;
  DeviceBuffer<uint8_t> pyr_[SVS_NUM_PYR_LEVELS];
  DeviceBuffer<float> disp_;
  DeviceBuffer<float> f32_[SVS_NUM_PYR_LEVELS], dx_[SVS_NUM_PYR_LEVELS], dy_[SVS_NUM_PYR_LEVELS];
  int w_[SVS_NUM_PYR_LEVELS], h_[SVS_NUM_PYR_LEVELS], stride_[SVS_NUM_PYR_LEVELS];
};

// FastGrid constructor arithmetic (fast_grid.cpp:23-58) -- integer bookkeeping only
inline svs_fastgrid makeFastGrid(int img_w, int img_h, int num_features_per_cell, int boundary_per_cell, int fast_thr,
                                 int grid_w, int grid_h, int fast_min = 10, int fast_max = 40) {
  svs_fastgrid g;
  g.gx = grid_w; g.gy = grid_h;
  g.min_inner = (int)(num_features_per_cell - boundary_per_cell * 0.33);
  g.min_outer = num_features_per_cell - boundary_per_cell;
  g.max_inner = (int)(num_features_per_cell + boundary_per_cell * 0.33);
  g.max_outer = num_features_per_cell + boundary_per_cell;
  g.cell_w = img_w / grid_w; g.cell_h = img_h / grid_h;
  g.fast_min = fast_min; g.fast_max = fast_max;
  for (int i = 0; i < SVS_MAX_CELLS; ++i) g.thr[i] = fast_thr;
  return g;
}

// All per-level FastGrids of StereoFrontend::fast_grid_ (stereo_frontend.cpp:73-88) in one object.
class FastGrid {
 public:
  FastGrid(const Context &c, int n_levels, const int32_t *w, const int32_t *h, const svs_fastgrid *grids, int corner_cap = 8192)
      : ctx_(c), f_(nullptr), n_levels_(n_levels), cap_(corner_cap) {
    for (int l = 0; l < n_levels; ++l) grids_[l] = grids[l];
    c.check(svs_fast_create(c.get(), n_levels, w, h, grids, 1, corner_cap, &f_));
  }
  ~FastGrid() { if (f_) svs_fast_destroy(f_); }
  FastGrid(const FastGrid &) = delete;
  FastGrid &operator=(const FastGrid &) = delete;
  // void FastGrid::detectAdaptively(const cv::Mat& img, int trials, QuadTree<int>* qt) for every level
  // (stereo_frontend.cpp:657-679); corners[l] comes back in the quadtree insertion order
  bool detectAdaptively(const FrameDev &fr, int trials, std::vector<Corner> *corners /* [n_levels] */) { return run(fr, trials, corners); }
  // static void FastGrid::detect(const cv::Mat&, const CellGrid2d&, QuadTree<int>*): stored thresholds
  bool detect(const FrameDev &fr, std::vector<Corner> *corners) { return run(fr, 0, corners); }
  // const CellGrid2d& cell_grid2d() const : persistent per-cell thresholds of one level
  std::vector<int32_t> cell_grid2d(int level) {
    std::vector<int32_t> thr((size_t)grids_[level].gx * grids_[level].gy);
    ctx_.check(svs_fast_download(f_, 0, level, nullptr, 0, nullptr, nullptr, nullptr, thr.data()));
    return thr;
  }
  // corners kept per cell by the last detection, cells row-major: the reference's quadtree content is the corner's index
  // INSIDE its cell (fast_grid.cpp:143-149), so the host walks corners[l] with these counts
  std::vector<int32_t> cell_counts(int level) {
    std::vector<int32_t> cnt((size_t)grids_[level].gx * grids_[level].gy);
    ctx_.check(svs_fast_download(f_, 0, level, nullptr, 0, nullptr, cnt.data(), nullptr, nullptr));
    return cnt;
  }
  bool set_cell_grid2d(int level, const std::vector<int32_t> &thr) { return ctx_.check(svs_fast_set_thresholds(f_, 0, level, thr.data())); }
  svs_fast *handle() const { return f_; }

 private:
  bool run(const FrameDev &fr, int trials, std::vector<Corner> *corners) {
    const uint8_t *img[SVS_NUM_PYR_LEVELS]; int32_t stride[SVS_NUM_PYR_LEVELS]; size_t bs[SVS_NUM_PYR_LEVELS];
    for (int l = 0; l < n_levels_; ++l) { img[l] = fr.pyr(l); stride[l] = fr.stride(l); bs[l] = 0; }
    if (!ctx_.check(svs_fast_detect(f_, img, stride, bs, 1, trials))) return false;
    for (int l = 0; corners && l < n_levels_; ++l) {
      corners[l].resize((size_t)cap_);
      int32_t n = 0;
      if (!ctx_.check(svs_fast_download(f_, 0, l, &corners[l][0].x, cap_, &n, nullptr, nullptr, nullptr))) return false;
      corners[l].resize((size_t)n);
    }
    return true;
  }
  const Context &ctx_;
  svs_fast *f_;
  int n_levels_, cap_;
  svs_fastgrid grids_[SVS_NUM_PYR_LEVELS];
};

// GuidedMatcher<StereoCamera>::match (matcher.hpp:67-83).  keyframe_map / vertex_map look-ups and the
// append into TrackData stay on the host (hash-map bookkeeping); the per-point arithmetic runs on
// the device.  results[i].status == SVS_MATCH_OK <=> the reference appends an observation.
class GuidedMatcher {
 public:
  explicit GuidedMatcher(const Context &c) : ctx_(c) {}
  bool match(const std::vector<svs_keyframe> &keyframes /* device pyramids + T_anchor_from_w */,
             const double T_cur_from_actkey[12], const double T_actkey_from_w[12], const FrameDev &cur_frame,
             FastGrid &feature_tree, const svs_cam cam_vec[SVS_NUM_PYR_LEVELS],
             const std::vector<svs_candidate_point> &ap_map, int SEARCHRADIUS, int thr_mean, int thr_std,
             std::vector<svs_match_result> *results) {
    const size_t n = ap_map.size();
    results->assign(n, svs_match_result());
    if (n == 0) return true;
    double Tcw[12], Twk[12];
    mul(T_cur_from_actkey, T_actkey_from_w, Tcw);     // matcher.cpp:330
    inv(T_actkey_from_w, Twk);                        // matcher.cpp:328
    DeviceBuffer<svs_keyframe> d_kf(ctx_, keyframes.size());
    DeviceBuffer<svs_candidate_point> d_pts(ctx_, n);
    DeviceBuffer<double> d_T(ctx_, 24);
    DeviceBuffer<svs_match_result> d_out(ctx_, n);
    double TT[24];
    std::memcpy(TT, Tcw, sizeof Tcw); std::memcpy(TT + 12, Twk, sizeof Twk);
    if (!d_kf.upload(keyframes.data(), keyframes.size()) || !d_pts.upload(ap_map.data(), n) || !d_T.upload(TT, 24)) return false;
    svs_match_args a;
    std::memset(&a, 0, sizeof a);
    a.d_kfs = d_kf.get(); a.n_kf = (int32_t)keyframes.size(); a.d_pts = d_pts.get(); a.n_pts = (int32_t)n;
    a.d_T_cur_from_w = d_T.get(); a.d_T_w_from_actkey = d_T.get() + 12;
    for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l) { a.d_cur_pyr[l] = cur_frame.pyr(l); a.cur_stride[l] = cur_frame.stride(l); a.cam_vec[l] = cam_vec[l]; }
    a.d_disp = cur_frame.disp(); a.disp_stride = cur_frame.stride(0);
    a.search_radius = SEARCHRADIUS; a.thr_mean = thr_mean; a.thr_std = thr_std; a.n_batch = 1;
    if (!ctx_.check(svs_match(ctx_.get(), &a, feature_tree.handle(), d_out.get()))) return false;
    return d_out.download(results->data(), n);
  }

 private:
  static void mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 4; ++j) C[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j]; C[4 * i + 3] += A[4 * i + 3]; }
  }
  static void inv(const double *A, double *B) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[4 * i + j] = A[4 * j + i];
    for (int i = 0; i < 3; ++i) B[4 * i + 3] = -(B[4 * i] * A[3] + B[4 * i + 1] * A[7] + B[4 * i + 2] * A[11]);
  }
  const Context &ctx_;
};

// PoseOptimizer<SE3,6,IdObs<3>,3> (BA_SE3_XYZ_STEREO, pose_optimizer.h:486): calcFastMotionOnly over the matcher's
// TrackData (stereo_frontend.cpp:1058-1063).  `track` = the result records of one or more GuidedMatcher::match calls
// (entries with status != SVS_MATCH_OK are skipped, as they never reach obs_list).
struct PoseOptimizerParams {             // pose_optimizer.h:36-58
  PoseOptimizerParams(bool robust_kernel = true, double kernel_param = 1, int num_iter = 50, double initial_mu = -1)
      : robust_kernel(robust_kernel), kernel_param(kernel_param), num_iter(num_iter), initial_mu(initial_mu), tau(0.00001) {}
  bool robust_kernel; double kernel_param; int num_iter; double initial_mu; double tau;
};
class BA_SE3_XYZ_STEREO {
 public:
  explicit BA_SE3_XYZ_STEREO(const Context &c) : ctx_(c) {}
  bool calcFastMotionOnly(const std::vector<svs_match_result> &track, const svs_cam &cam, const PoseOptimizerParams &ba_params,
                          double T_cur_from_actkey[12], svs_pose_opt_stats *stats) {
    DeviceBuffer<svs_match_result> d_res(ctx_, track.size() ? track.size() : 1);
    DeviceBuffer<double> d_T(ctx_, 12);
    DeviceBuffer<svs_pose_opt_stats> d_st(ctx_, 1);
    if ((track.size() && !d_res.upload(track.data(), track.size())) || !d_T.upload(T_cur_from_actkey, 12)) return false;
    svs_pose_opt_params p;
    std::memset(&p, 0, sizeof p);
    p.robust_kernel = ba_params.robust_kernel; p.num_iter = ba_params.num_iter; p.kernel_param = ba_params.kernel_param;
    p.initial_mu = ba_params.initial_mu; p.tau = ba_params.tau;
    if (!ctx_.check(svs_motion_only(ctx_.get(), d_res.get(), (int)track.size(), track.size(), &cam, &p, d_T.get(), d_st.get(), 1))) return false;
    svs_pose_opt_stats st;
    if (!d_T.download(T_cur_from_actkey, 12) || !d_st.download(&st, 1)) return false;
    if (stats) *stats = st;
    return st.status == 0;      // 1: empty list (assert in the reference), 2: NaN residual (throw in the reference)
  }
  // StereoFrontend::processMatchedPoints (stereo_frontend.cpp:834-974), numeric part: reprojection gate at
  // T_cur_from_actkey, PointStatistics counters, level positions.  `points[i]` is the candidate point record i of
  // `track` was matched from; records below n_new_records stem from the new-feature match calls (:989-1030).
  // The caller then walks `gated` to fill new_point_list / track_point_list / point_tree (host containers).
  bool processMatchedPoints(const std::vector<svs_match_result> &track, const std::vector<svs_candidate_point> &points, int n_new_records,
                            const svs_cam &cam, const double T_cur_from_actkey[12], float max_reproj_error,
                            std::vector<svs_gated_point> *gated, svs_point_stats *stats) {
    if (track.size() != points.size() || !gated || !stats) return false;
    const size_t n = track.size();
    DeviceBuffer<svs_match_result> d_res(ctx_, n ? n : 1);
    DeviceBuffer<svs_candidate_point> d_pts(ctx_, n ? n : 1);
    DeviceBuffer<svs_gated_point> d_g(ctx_, n ? n : 1);
    DeviceBuffer<double> d_T(ctx_, 12);
    DeviceBuffer<svs_point_stats> d_st(ctx_, 1);
    if (n && (!d_res.upload(track.data(), n) || !d_pts.upload(points.data(), n))) return false;
    if (!d_T.upload(T_cur_from_actkey, 12)) return false;
    if (!ctx_.check(svs_process_matched_points(ctx_.get(), d_res.get(), d_pts.get(), (int)n, n, n, n_new_records, &cam, d_T.get(), max_reproj_error,
                                               d_g.get(), n, d_st.get(), 1))) return false;
    gated->resize(n);
    if (n && !d_g.download(gated->data(), n)) return false;
    return d_st.download(stats, 1);
  }

 private:
  const Context &ctx_;
};

// StereoFrontend::calcDisparityCpu (stereo_frontend.cpp:620-653): cv::StereoBM on the level-0 left image and the
// right image; the float disparity lands in the frame's disparity image (FrameDev::disp, -1 where filtered).
class StereoBM {
 public:
  StereoBM(const Context &c, int w, int h, int num_disp16 = 2) : ctx_(c), s_(0), w_(w), h_(h), right_(c, (size_t)((w + 63) / 64 * 64) * h) {
    svs_stereo_params p = {31, 7, 0, 16 * num_disp16, 10, 15, 100, 32, 1};      // state set at stereo_frontend.cpp:626-636
    ok_ = c.check(svs_stereo_create(c.get(), w, h, 1, &p, &s_));
  }
  ~StereoBM() { if (s_) svs_stereo_destroy(s_); }
  bool ok() const { return ok_; }
  bool operator()(const FrameDev &left, const Image8 &right, FrameDev *out) {
    const int stride = left.stride(0);
    std::vector<uint8_t> tmp((size_t)stride * h_);
    for (int y = 0; y < h_; ++y) std::memcpy(&tmp[(size_t)y * stride], right.data + (size_t)y * right.stride, (size_t)w_);
    if (!ok_ || !right_.upload(tmp.data(), tmp.size())) return false;
    return ctx_.check(svs_stereo_compute(s_, left.pyr(0), stride, 0, right_.get(), stride, 0, const_cast<float *>(out->disp()), stride, 0, 1));
  }

 private:
  StereoBM(const StereoBM &);
  StereoBM &operator=(const StereoBM &);
  const Context &ctx_;
  svs_stereo *s_;
  int w_, h_;
  bool ok_;
  DeviceBuffer<uint8_t> right_;
};

// DenseTracker, CPU-path semantics (dense_tracking.h:53-97, dense_tracking.cpp:222-423).
class DenseTracker {
 public:
  DenseTracker(const Context &c, const svs_cam cam_vec[SVS_NUM_PYR_LEVELS]) : ctx_(c), d_T_(c, 12), d_T_jac_(c, 36), d_passes_(c, 1) {
    for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l) {
      cam_[l] = cam_vec[l];
      const size_t n4 = (size_t)(cam_vec[l].w / 4) * (cam_vec[l].h / 4) * 4;
      cloud_[l] = DeviceBuffer<float>(c, n4);
      d_rimg_[l] = DeviceBuffer<float>(c, n4);
      residual_img[l].assign(n4, 0.f);                                   // setTo(cv::Scalar(0,0,0,1)), dense_tracking.cpp:54
      for (size_t i = 3; i < n4; i += 4) residual_img[l][i] = 1.f;
    }
  }
  std::vector<float> residual_img[SVS_NUM_PYR_LEVELS];                   // public member of the reference (dense_tracking.h:66), float4 per sample
  // void computeDensePointCloudCpu(const SE3& T_cur_from_actkey): ref_dense_points_ from the frame's disparity
  bool computeDensePointCloudCpu(const FrameDev &fr, const double T_cur_from_actkey[12]) {
    if (!d_T_.upload(T_cur_from_actkey, 12)) return false;
    for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l)
      if (!ctx_.check(svs_pointcloud_cpu_sem(ctx_.get(), fr.disp(), fr.stride(0), 0, &cam_[l], l, d_T_.get(), cloud_[l].get(), 0, 1))) return false;
    return svs_ctx_sync(ctx_.get()) == SVS_OK;
  }
  // ref_dense_points_[level] given by the host (state restored from a saved session; tests): float4 per quarter-grid sample, row-major
  bool setDensePointCloud(int level, const float *h_cloud4) { return cloud_[level].upload(h_cloud4, cloud_[level].size()); }
  bool getDensePointCloud(int level, float *h_cloud4) const { return cloud_[level].download(h_cloud4, cloud_[level].size()); }
  // void denseTrackingCpu(SE3* T_cur_from_actkey): in/out pose, whole LM loop in one device launch
  bool denseTrackingCpu(const FrameDev &prev, const FrameDev &cur, double T_cur_from_actkey[12]) {
    svs_dense_track_args a;
    std::memset(&a, 0, sizeof a);
    for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l) {
      a.d_cloud[l] = cloud_[l].get(); a.d_prev_u8[l] = prev.pyr(l); a.pstride[l] = prev.stride(l);
      a.d_cur[l] = cur.f32(l); a.d_dx[l] = cur.dx(l); a.d_dy[l] = cur.dy(l); a.fstride[l] = cur.stride(l); a.cam_vec[l] = cam_[l];
    }
    a.d_T_jac_out = d_T_jac_.get();
    if (!d_T_.upload(T_cur_from_actkey, 12)) return false;
    if (!ctx_.check(svs_dense_track_cpu_sem(ctx_.get(), &a, d_T_.get(), d_passes_.get(), 1))) return false;
    if (want_residual_img) {                                             // the reference always writes it; here it is opt-in (GUI only)
      for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l) {
        if (!ctx_.check(svs_dense_residual_image_cpu_sem(ctx_.get(), cloud_[l].get(), 0, prev.pyr(l), prev.stride(l), 0, cur.f32(l), cur.stride(l), 0,
                                                         nullptr, 0, 0, &cam_[l], d_T_jac_.get() + 12 * l, 36, d_rimg_[l].get(), 0, 1))) return false;
        if (!d_rimg_[l].download(residual_img[l].data(), residual_img[l].size())) return false;
      }
    }
    return d_T_.download(T_cur_from_actkey, 12);
  }
  bool want_residual_img = false;

 private:
  const Context &ctx_;
  svs_cam cam_[SVS_NUM_PYR_LEVELS];
  DeviceBuffer<float> cloud_[SVS_NUM_PYR_LEVELS], d_rimg_[SVS_NUM_PYR_LEVELS];
  DeviceBuffer<double> d_T_, d_T_jac_;
  DeviceBuffer<int32_t> d_passes_;
};


// GpuTracker (gpu/dense_tracking.cuh:281-342): the CUDA build's per-pass call surface, same argument lists (device pointers into the
// caller's images, strides in elements), blocking like the reference (each call ends with a device synchronisation there too).
struct GpuIntrinsics { float focal_length, pp_x, pp_y; void set(double fl, double px, double py) { focal_length = (float)fl; pp_x = (float)px; pp_y = (float)py; } };
struct GpuMatrix34 { float data_colmajor[12]; void set(const double *m_colmajor) { for (int i = 0; i < 12; ++i) data_colmajor[i] = (float)m_colmajor[i]; } };
struct GpuMatrix4 { float data_colmajor[16]; void set(const double *m_colmajor) { for (int i = 0; i < 16; ++i) data_colmajor[i] = (float)m_colmajor[i]; } };
struct GpuTrackingData { double hessian[21], jacobian_times_res[6]; };      // packed upper by column, as GpuSymMatrix6 (f64 sums here)
// The reference's own result layout (gpu/dense_tracking.cuh:40-277): 21 + 6 FLOATS with the copyTo members its caller uses
// (`tracking_data.hessian.copyTo(H.data())`, dense_tracking.cpp:100-104), for a host that keeps the reference's loop untouched.  The sums are
// still formed in f64 on the device and narrowed once at the end.
struct GpuVector6F32 {
  static const int NUM_ROWS = 6;
  float data[NUM_ROWS];
  void copyTo(double *vec) const { for (int i = 0; i < NUM_ROWS; ++i) vec[i] = data[i]; }
};
struct GpuSymMatrix6F32 {
  static const int NUM_ROWS = 6;
  float data[21];
  void copyTo(double *mat_colmajor) const {      // gpu/dense_tracking.cuh:118-132
    int i = 0;
    for (int c = 0; c < NUM_ROWS; ++c)
      for (int r = 0; r <= c; ++r) { const float v = data[i]; mat_colmajor[r + NUM_ROWS * c] = v; mat_colmajor[c + NUM_ROWS * r] = v; ++i; }
  }
};
struct GpuTrackingDataF32 { GpuSymMatrix6F32 hessian; GpuVector6F32 jacobian_times_res; };
inline bool computePointCloud(const Context &c, const GpuMatrix4 &TQ_actkey_from_cur, const float *d_disparities, int width, int height, int stride_in,
                              int stride_out, int factor, float *d_point_cloud4) {
  return c.check(svs_pointcloud_full(c.get(), TQ_actkey_from_cur.data_colmajor, d_disparities, width, height, stride_in, stride_out, factor, d_point_cloud4)) &&
         svs_ctx_sync(c.get()) == SVS_OK;
}
class GpuTracker {
 public:
  GpuTracker(const Context &c, int /*width*/, int /*height*/) : ctx_(c), d_sums_(c, 1), cur_(nullptr), dx_(nullptr), dy_(nullptr) {}
  void bindTexture(const float *d_img_cur, const float *d_dx_img_cur, const float *d_dy_img_cur, int, int, int) { cur_ = d_img_cur; dx_ = d_dx_img_cur; dy_ = d_dy_img_cur; }
  bool jacobianReduction(const float *d_img_prev, const float *d_point_cloud_prev4, const GpuMatrix34 &T_cur_from_prev, const GpuIntrinsics &K, int width, int height,
                         int stride_float_img, int stride_float4_img, GpuTrackingData *tracking_result) {
    svs_dense_sums s;
    if (!pass(d_img_prev, d_point_cloud_prev4, T_cur_from_prev, K, width, height, stride_float_img, stride_float4_img, 1, &s)) return false;
    for (int i = 0; i < 21; ++i) tracking_result->hessian[i] = s.H[i];
    for (int i = 0; i < 6; ++i) tracking_result->jacobian_times_res[i] = s.b[i];
    return true;
  }
  // the same call with the reference's result type (21 + 6 floats)
  bool jacobianReduction(const float *d_img_prev, const float *d_point_cloud_prev4, const GpuMatrix34 &T_cur_from_prev, const GpuIntrinsics &K, int width, int height,
                         int stride_float_img, int stride_float4_img, GpuTrackingDataF32 *tracking_result) {
    GpuTrackingData d;
    if (!jacobianReduction(d_img_prev, d_point_cloud_prev4, T_cur_from_prev, K, width, height, stride_float_img, stride_float4_img, &d)) return false;
    for (int i = 0; i < 21; ++i) tracking_result->hessian.data[i] = (float)d.hessian[i];
    for (int i = 0; i < 6; ++i) tracking_result->jacobian_times_res.data[i] = (float)d.jacobian_times_res[i];
    return true;
  }
  float chi2(const float *d_img_prev, const float *d_point_cloud_prev4, const GpuMatrix34 &T_cur_from_prev, const GpuIntrinsics &K, int width, int height,
             int stride_float_img, int stride_float4_img) {
    svs_dense_sums s;
    return pass(d_img_prev, d_point_cloud_prev4, T_cur_from_prev, K, width, height, stride_float_img, stride_float4_img, 0, &s) ? (float)s.chi2 : -1.f;
  }
  bool residualImage(const float *d_img_prev, const float *d_point_cloud_prev4, const GpuMatrix34 &T, const GpuIntrinsics &K, int width, int height,
                     int stride_float_img, int stride_float4_img, float *d_res_img4) {
    return ctx_.check(svs_dense_residual_image_full(ctx_.get(), d_point_cloud_prev4, width, height, stride_float4_img, d_img_prev, cur_, stride_float_img,
                                                    K.focal_length, K.pp_x, K.pp_y, T.data_colmajor, d_res_img4)) && svs_ctx_sync(ctx_.get()) == SVS_OK;
  }

 private:
  bool pass(const float *prev, const float *cloud4, const GpuMatrix34 &T, const GpuIntrinsics &K, int w, int h, int sf, int s4, int jac, svs_dense_sums *out) {
    return ctx_.check(svs_dense_pass_full(ctx_.get(), cloud4, w, h, s4, prev, cur_, dx_, dy_, sf, K.focal_length, K.pp_x, K.pp_y, T.data_colmajor, jac, d_sums_.get())) &&
           d_sums_.download(out, 1);
  }
  const Context &ctx_;
  DeviceBuffer<svs_dense_sums> d_sums_;
  const float *cur_, *dx_, *dy_;
};

// DenseTracker of the CUDA build: denseTrackingGpu(SE3*) as ONE launch (dense_tracking.cpp:60-193).  The caller hands over the device images of
// FrameData (gpu_pyr_float32 / _dx / _dy of the current frame, gpu_pyr_float32 of the previous one) and dev_ref_dense_points_.
class DenseTrackerGpu {
 public:
  explicit DenseTrackerGpu(const Context &c) : ctx_(c), d_T_(c, 12), d_passes_(c, 1) {}
  struct Levels {                          // per pyramid level, device pointers; strides in elements (float / float4)
    const float *cloud4[SVS_NUM_PYR_LEVELS], *prev[SVS_NUM_PYR_LEVELS], *cur[SVS_NUM_PYR_LEVELS], *dx[SVS_NUM_PYR_LEVELS], *dy[SVS_NUM_PYR_LEVELS];
    int stride_f4[SVS_NUM_PYR_LEVELS], stride_f[SVS_NUM_PYR_LEVELS];
    svs_cam cam_vec[SVS_NUM_PYR_LEVELS];
  };
  // void denseTrackingGpu(SE3 * T_cur_from_actkey): in/out pose.  passes (optional): fused sweeps executed
  bool denseTrackingGpu(const Levels &L, double T_cur_from_actkey[12], int *passes = nullptr) {
    svs_dense_track_full_args a;
    std::memset(&a, 0, sizeof a);
    for (int l = 0; l < SVS_NUM_PYR_LEVELS; ++l) {
      a.d_cloud4[l] = L.cloud4[l]; a.stride_f4[l] = L.stride_f4[l]; a.d_prev[l] = L.prev[l]; a.d_cur[l] = L.cur[l]; a.d_dx[l] = L.dx[l]; a.d_dy[l] = L.dy[l];
      a.stride_f[l] = L.stride_f[l]; a.w[l] = L.cam_vec[l].w; a.h[l] = L.cam_vec[l].h; a.f[l] = L.cam_vec[l].f; a.cx[l] = L.cam_vec[l].cx; a.cy[l] = L.cam_vec[l].cy;
    }
    if (!d_T_.upload(T_cur_from_actkey, 12)) return false;
    if (!ctx_.check(svs_dense_track_full(ctx_.get(), &a, d_T_.get(), d_passes_.get(), 1))) return false;
    int32_t p = 0;
    if (!d_passes_.download(&p, 1) || p < 0) return false;
    if (passes) *passes = p;
    return d_T_.download(T_cur_from_actkey, 12);
  }

 private:
  const Context &ctx_;
  DeviceBuffer<double> d_T_;
  DeviceBuffer<int32_t> d_passes_;
};

// StereoFrontend::processFrame(bool*) / processFirstFrame() (stereo_frontend.h:88-95): the data-parallel part of a frame in ONE blocking call with
// host buffers in and out; the caller keeps the reference's bookkeeping (keyframe switch / drop, list building) and uses the returned records.
class StereoFrontend {
 public:
  StereoFrontend(const Context &c, const svs_cam &cam, const svs_frontend_params &prm, int max_points, int max_keyframes) : ctx_(c), fe_(nullptr) {
    ok_ = c.check(svs_frontend_create(c.get(), &cam, &prm, max_points, max_keyframes, &fe_));
  }
  ~StereoFrontend() { if (fe_) svs_frontend_destroy(fe_); }
  StereoFrontend(const StereoFrontend &) = delete;
  StereoFrontend &operator=(const StereoFrontend &) = delete;
  // cuda_build: the reference compiled with SCAVISLAM_CUDA_SUPPORT (full-resolution denseTrackingGpu, matcher search radius 4, stereo_frontend.cpp:1043-1047);
  // n_levels: use_n_levels_in_frontent (stereo_frontend.cpp:68; the code default is 2, the shipped configurations set 3)
  static svs_frontend_params referenceParams(bool block_matching, bool cuda_build = false, int n_levels = 3) {
    svs_frontend_params p;
    std::memset(&p, 0, sizeof p);
    p.fast_trials = 6; p.search_radius = cuda_build ? 4 : 8; p.thr_mean = 22; p.thr_std = 10; p.max_reproj_error = 2.f; p.use_block_matching = block_matching ? 1 : 0;
    p.n_levels = n_levels; p.num_max_points = 300; p.min_matches = 20; p.cuda_build = cuda_build ? 1 : 0;
    p.pose_opt.robust_kernel = 1; p.pose_opt.num_iter = 15; p.pose_opt.kernel_param = 2.0; p.pose_opt.initial_mu = -1.0; p.pose_opt.tau = 1e-5;
    const svs_stereo_params sp = {31, 7, 0, 32, 10, 15, 100, 32, 1};
    p.stereo = sp;
    return p;
  }
  bool ok() const { return ok_; }
  bool processFirstFrame(Image8 left, const Image8 *right, const ImageF *disp) {
    return ctx_.check(svs_frontend_first_frame(fe_, left.data, left.stride, right ? right->data : nullptr, right ? right->stride : 0, disp ? disp->data : nullptr,
                                               disp ? disp->stride : 0));
  }
  bool keepKeyframe(int slot, const double T_kf_from_w[12]) { return ctx_.check(svs_frontend_keep_keyframe(fe_, slot, T_kf_from_w)); }
  bool setCandidates(const std::vector<svs_candidate_point> &ap_map, int n_new_records) {
    n_ = (int)ap_map.size();
    return ctx_.check(svs_frontend_set_candidates(fe_, ap_map.data(), n_, n_new_records));
  }
  // matchAndTrack's lists in the order it walks them (stereo_frontend.cpp:976-1050): group_end[0] = end of newpoint_map[actkey_id], one end per neighbour in
  // strength_to_neighbors order, the last = ap_map.size() = end of neighborhood_->point_list
  bool setCandidateLists(const std::vector<svs_candidate_point> &ap_map, const std::vector<int32_t> &group_end) {
    n_ = (int)ap_map.size();
    return ctx_.check(svs_frontend_set_candidates_grouped(fe_, 0, ap_map.data(), n_, group_end.data(), (int)group_end.size()));
  }
  // the grabber side of frame_grabber.hpp:93-155: hand the NEXT frame over while the current one is processed; processFrame / processFirstFrame then get no image
  bool prefetchFrame(Image8 left, const Image8 *right, const ImageF *disp) {
    return ctx_.check(svs_frontend_prefetch_frame(fe_, left.data, left.stride, right ? right->data : nullptr, right ? right->stride : 0, disp ? disp->data : nullptr,
                                                  disp ? disp->stride : 0));
  }
  bool processPrefetchedFrame(double T_cur_from_actkey[12], const double T_actkey_from_w[12], svs_frame_result *res, std::vector<svs_match_result> *matches,
                              std::vector<svs_gated_point> *gated) {
    matches->resize((size_t)n_); gated->resize((size_t)n_);
    if (!ctx_.check(svs_frontend_process_frame(fe_, nullptr, 0, nullptr, 0, nullptr, 0, T_cur_from_actkey, T_actkey_from_w, res, matches->data(), gated->data())))
      return false;
    for (int i = 0; i < 12; ++i) T_cur_from_actkey[i] = res->T_cur_from_actkey[i];
    return res->tracking_ok != 0;
  }
  // returns what matchAndTrack returns (enough features matched); *T_cur_from_actkey is in/out like the reference's member
  bool processFrame(Image8 left, const Image8 *right, const ImageF *disp, double T_cur_from_actkey[12], const double T_actkey_from_w[12], svs_frame_result *res,
                    std::vector<svs_match_result> *matches, std::vector<svs_gated_point> *gated) {
    matches->resize((size_t)n_); gated->resize((size_t)n_);
    if (!ctx_.check(svs_frontend_process_frame(fe_, left.data, left.stride, right ? right->data : nullptr, right ? right->stride : 0, disp ? disp->data : nullptr,
                                               disp ? disp->stride : 0, T_cur_from_actkey, T_actkey_from_w, res, matches->data(), gated->data())))
      return false;
    for (int i = 0; i < 12; ++i) T_cur_from_actkey[i] = res->T_cur_from_actkey[i];
    return res->tracking_ok != 0;
  }
  bool recomputeDensePointCloud(const double T_cur_from_actkey[12]) { return ctx_.check(svs_frontend_recompute_cloud(fe_, T_cur_from_actkey)); }
  // computeFastCorners' outputs of the frame processed last (stereo_frontend.cpp:656-679): the corner list of a level in the order FastGrid inserts it into the
  // quadtree (cells row-major, corners of a cell row-major), corners per cell, and the persistent thresholds as they stand now (the CellGrid2d snapshot a Frame keeps)
  bool fastCorners(int level, std::vector<Corner> *corners, std::vector<int32_t> *cell_counts, std::vector<int32_t> *thresholds) {
    svs_fast *f = nullptr;
    if (!ctx_.check(svs_frontend_device_view(fe_, 0, nullptr, nullptr, nullptr, nullptr, &f)) || !f) return false;
    corners->resize(16384); cell_counts->assign(SVS_MAX_CELLS, 0); thresholds->assign(SVS_MAX_CELLS, 0);
    int32_t n = 0;
    if (!ctx_.check(svs_fast_download(f, 0, level, &(*corners)[0].x, (int)corners->size(), &n, cell_counts->data(), nullptr, thresholds->data()))) return false;
    corners->resize((size_t)n);
    return true;
  }
  svs_frontend *handle() const { return fe_; }

 private:
  const Context &ctx_;
  svs_frontend *fe_;
  bool ok_;
  int n_ = 0;
};

// One rank of a landmark-sharded back-end: rank 0 calls uniqueId() and distributes it (MPI_Bcast, a TCP store, a file); every rank then constructs
// the communicator and attaches it to its SlamGraphBA with attach().  svs_ba_optimize all-reduces over RCCL on the context's stream.
class Communicator {
 public:
  static bool uniqueId(const Context &c, svs_unique_id *id) { return c.check(svs_comm_get_unique_id(c.get(), id)); }
  Communicator(const Context &c, const svs_unique_id &id, int rank, int world) : comm_(nullptr) { ok_ = c.check(svs_comm_create(c.get(), &id, rank, world, &comm_)); }
  ~Communicator() { if (comm_) svs_comm_destroy(comm_); }
  Communicator(const Communicator &) = delete;
  Communicator &operator=(const Communicator &) = delete;
  bool ok() const { return ok_; }
  svs_comm *get() const { return comm_; }

 private:
  svs_comm *comm_;
  bool ok_;
};

// SlamGraph::optimize(const OptParams&) (slam_graph.hpp:457-462): the caller marshals the double
// window exactly as copyDataToG2o does (slam_graph.cpp:983-1032) into flat arrays.
struct OptParams {                       // slam_graph.hpp:36-50
  OptParams(int num_iters, bool use_robust_kernel, double huber_kernel_width)
      : num_iters(num_iters), use_robust_kernel(use_robust_kernel), huber_kernel_width(huber_kernel_width) {}
  int num_iters; bool use_robust_kernel; double huber_kernel_width;
};
class SlamGraphBA {
 public:
  explicit SlamGraphBA(const Context &c) : ctx_(c), ba_(nullptr) { c.check(svs_ba_create(c.get(), &ba_)); }
  ~SlamGraphBA() { if (ba_) svs_ba_destroy(ba_); }
  SlamGraphBA(const SlamGraphBA &) = delete;
  SlamGraphBA &operator=(const SlamGraphBA &) = delete;
  // poses [P][12] and psi [L][3] are updated in place, like restoreDataFromG2o (slam_graph.cpp:1035-1058)
  bool optimize(const OptParams &opt, const svs_cam &cam, std::vector<double> *poses, std::vector<double> *psi,
                const std::vector<svs_ba_edge> &obs_edges, const std::vector<svs_ba_constraint> &constraints,
                svs_ba_stats *stats = nullptr, bool exact_self_edges = false) {
    svs_ba_params prm = params(opt, exact_self_edges);
    const int P = (int)(poses->size() / 12), L = (int)(psi->size() / 3);
    if (!ctx_.check(svs_ba_set_problem(ba_, P, poses->data(), L, psi->data(), (int)obs_edges.size(), obs_edges.data(),
                                       (int)constraints.size(), constraints.data(), &cam, &prm, 1))) return false;
    if (!ctx_.check(svs_ba_optimize(ba_, nullptr, nullptr, stats))) return false;
    return ctx_.check(svs_ba_get_state(ba_, poses->data(), psi->data()));
  }
  // ---- the marshalling of SlamGraph::copyDataToG2o / restoreDataFromG2o (slam_graph.cpp:983-1058) from records that keep the GRAPH'S OWN IDS ----
  // One record per addObsToG2o call (slam_graph-impl.cpp:44-97): the ImageFeature of vertex `pose_id`'s feature_table for point `point_id`
  struct ObsById { int point_id, pose_id, level; double center[3]; };
  // One record per addConstraintToG2o call (slam_graph-impl.cpp:99-126).  copyContraintsToG2o (slam_graph.cpp:937-981) walks ORDERED pairs of the double
  // window, so a marginalised edge with an OUTER end is handed over twice: (1, 2, T_2_from_1, Lambda_2_from_1) and (2, 1, T_1_from_2, Lambda_1_from_2) --
  // the caller passes both, as the reference does
  struct ConstraintById { int pose_id_1, pose_id_2; double T_2_from_1[12], Lambda_2_from_1[36]; };
  // pose_ids / poses [P][12]: the double window in the order of copyPosesToG2o; point_ids / anchor_ids / xyz_anchor [L][3]: the active points (Point::xyz_anchor,
  // Point::anchorframe_id) -- converted to the inverse-depth parameters the solver works on (invert_depth, slam_graph.cpp:916) and back (:1054);
  // observations from frames outside the window are skipped like slam_graph.cpp:1001-1006 does; information = diag(4^-level, 4^-level, 0.333^2) (:1010-1015).
  // poses and xyz_anchor are updated in place.  false: an anchor frame outside the window (the reference asserts), or a library error
  bool optimizeWindow(const OptParams &opt, const svs_cam &cam, const std::vector<int> &pose_ids, std::vector<double> *poses, const std::vector<int> &point_ids,
                      const std::vector<int> &anchor_ids, std::vector<double> *xyz_anchor, const std::vector<ObsById> &obs,
                      const std::vector<ConstraintById> &constraints, svs_ba_stats *stats = nullptr) {
    const size_t P = pose_ids.size(), L = point_ids.size();
    if (poses->size() != 12 * P || xyz_anchor->size() != 3 * L || anchor_ids.size() != L) return false;
    std::vector<std::pair<int, int> > pmap(P), lmap(L);                      // (id, index), sorted: the hash maps of the reference flattened
    for (size_t i = 0; i < P; ++i) pmap[i] = std::make_pair(pose_ids[i], (int)i);
    for (size_t i = 0; i < L; ++i) lmap[i] = std::make_pair(point_ids[i], (int)i);
    std::sort(pmap.begin(), pmap.end()); std::sort(lmap.begin(), lmap.end());
    std::vector<double> psi(3 * L);
    std::vector<int> anchor_idx(L);
    for (size_t i = 0; i < L; ++i) {
      const double *x = &(*xyz_anchor)[3 * i];
      psi[3 * i] = x[0] / x[2]; psi[3 * i + 1] = x[1] / x[2]; psi[3 * i + 2] = 1. / x[2];      // invert_depth (maths_utils.h:66-69)
      if ((anchor_idx[i] = find(pmap, anchor_ids[i])) < 0) return false;
    }
    std::vector<svs_ba_edge> edges;
    edges.reserve(obs.size());
    for (size_t k = 0; k < obs.size(); ++k) {
      const int pt = find(lmap, obs[k].point_id), po = find(pmap, obs[k].pose_id);
      if (pt < 0 || po < 0) continue;                                         // point not active / frame not in the double window
      svs_ba_edge e;
      std::memset(&e, 0, sizeof e);
      const double f = 1. / (double)(1 << obs[k].level);                     // pyrFromZero_d(1., level)
      for (int c = 0; c < 3; ++c) e.obs[c] = obs[k].center[c];
      e.info[0] = e.info[1] = f * f; e.info[2] = 0.333 * 0.333;               // Po2(...) (slam_graph.cpp:1010-1015)
      e.point = pt; e.pose = po; e.anchor = anchor_idx[pt];
      edges.push_back(e);
    }
    std::vector<svs_ba_constraint> cons(constraints.size());
    for (size_t k = 0; k < constraints.size(); ++k) {
      std::memcpy(cons[k].T_21, constraints[k].T_2_from_1, sizeof cons[k].T_21);
      std::memcpy(cons[k].info, constraints[k].Lambda_2_from_1, sizeof cons[k].info);
      cons[k].pose1 = find(pmap, constraints[k].pose_id_1); cons[k].pose2 = find(pmap, constraints[k].pose_id_2);
      if (cons[k].pose1 < 0 || cons[k].pose2 < 0) return false;
    }
    if (!optimize(opt, cam, poses, &psi, edges, cons, stats)) return false;
    for (size_t i = 0; i < L; ++i) {                                          // invert_depth again (slam_graph.cpp:1054)
      double *x = &(*xyz_anchor)[3 * i];
      const double *s = &psi[3 * i];
      x[0] = s[0] / s[2]; x[1] = s[1] / s[2]; x[2] = 1. / s[2];
    }
    return true;
  }
  // The same call for a window that slides: the library keeps the observations it has been given (svs_ba_window_update), a call brings the window by ids,
  // the current values and the observations made SINCE THE LAST CALL.  new_obs: ObsById records; everything else as optimizeWindow
  bool optimizeSlidingWindow(const OptParams &opt, const svs_cam &cam, const std::vector<int> &pose_ids, std::vector<double> *poses,
                             const std::vector<int> &point_ids, const std::vector<int> &anchor_ids, std::vector<double> *xyz_anchor,
                             const std::vector<ObsById> &new_obs, const std::vector<ConstraintById> &constraints, svs_ba_stats *stats = nullptr) {
    const size_t P = pose_ids.size(), L = point_ids.size();
    if (poses->size() != 12 * P || xyz_anchor->size() != 3 * L || anchor_ids.size() != L) return false;
    std::vector<double> psi(3 * L);
    for (size_t i = 0; i < L; ++i) { const double *x = &(*xyz_anchor)[3 * i]; psi[3 * i] = x[0] / x[2]; psi[3 * i + 1] = x[1] / x[2]; psi[3 * i + 2] = 1. / x[2]; }
    std::vector<svs_ba_edge> eo(new_obs.size());
    for (size_t k = 0; k < new_obs.size(); ++k) {
      std::memset(&eo[k], 0, sizeof eo[k]);
      const double f = 1. / (double)(1 << new_obs[k].level);
      for (int c = 0; c < 3; ++c) eo[k].obs[c] = new_obs[k].center[c];
      eo[k].info[0] = eo[k].info[1] = f * f; eo[k].info[2] = 0.333 * 0.333;
      eo[k].point = new_obs[k].point_id; eo[k].pose = new_obs[k].pose_id;     // ids: the library resolves them against the window
    }
    std::vector<svs_ba_constraint> cons(constraints.size());
    for (size_t k = 0; k < constraints.size(); ++k) {
      std::memcpy(cons[k].T_21, constraints[k].T_2_from_1, sizeof cons[k].T_21);
      std::memcpy(cons[k].info, constraints[k].Lambda_2_from_1, sizeof cons[k].info);
      cons[k].pose1 = constraints[k].pose_id_1; cons[k].pose2 = constraints[k].pose_id_2;
    }
    svs_ba_params prm = params(opt, false);
    if (!ctx_.check(svs_ba_window_update(ba_, (int)P, pose_ids.data(), poses->data(), (int)L, point_ids.data(), psi.data(), anchor_ids.data(), (int)eo.size(),
                                         eo.data(), (int)cons.size(), cons.data(), &cam, &prm)))
      return false;
    if (!ctx_.check(svs_ba_optimize(ba_, nullptr, nullptr, stats))) return false;
    if (!ctx_.check(svs_ba_get_state(ba_, poses->data(), psi.data()))) return false;
    for (size_t i = 0; i < L; ++i) { double *x = &(*xyz_anchor)[3 * i]; const double *s = &psi[3 * i]; x[0] = s[0] / s[2]; x[1] = s[1] / s[2]; x[2] = 1. / s[2]; }
    return true;
  }
  bool windowReset() { return ctx_.check(svs_ba_window_reset(ba_)); }
  // landmark-sharded operation: this rank passes only ITS landmarks' edges to optimize() (pose terms on exactly one rank: add_pose_terms)
  bool attach(const Communicator *comm) { return ctx_.check(svs_ba_set_comm(ba_, comm ? comm->get() : nullptr)); }
  svs_ba *get() const { return ba_; }

 private:
  static svs_ba_params params(const OptParams &opt, bool exact_self_edges) {
    svs_ba_params prm;
    prm.num_iters = opt.num_iters; prm.use_robust = opt.use_robust_kernel ? 1 : 0;
    prm.huber_delta = 1.0;               // huber_kernel_width is dead in the reference (slam_graph-impl.cpp:86-90)
    prm.lambda_init = 50.0;              // slam_graph.cpp:338
    prm.max_trials = 5;                  // slam_graph.cpp:1073
    prm.self_edge_mode = exact_self_edges ? 1 : 0;
    return prm;
  }
  static int find(const std::vector<std::pair<int, int> > &m, int id) {
    std::vector<std::pair<int, int> >::const_iterator it = std::lower_bound(m.begin(), m.end(), std::make_pair(id, -1));
    return it != m.end() && it->first == id ? it->second : -1;
  }
  const Context &ctx_;
  svs_ba *ba_;
};

}  // namespace scavislam_hip
#endif
