// Drives the C++ adaptor classes (include/scavislam_hip.hpp, the reference-named call surfaces) on the
// GPU and prints results as text; tests/test_gpu_cpp_adaptor.py compares them with the CPU oracle.
// Input: a raw u8 image file (w h then pixels) and a BA problem dump written by the test.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "scavislam_hip.hpp"

using namespace scavislam_hip;

template <class T>
static bool read_vec(FILE *f, std::vector<T> *v, size_t n) { v->resize(n); return n == 0 || std::fread(v->data(), sizeof(T), n, f) == n; }

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  Context ctx(0);
  if (!ctx.ok()) { std::puts("NODEVICE"); return 3; }
  // ---- FastGrid::detectAdaptively on a pyramid built by FrameDev::preprocessing --------------------
  FILE *f = std::fopen(argv[1], "rb");
  int wh[2];
  if (!f || std::fread(wh, sizeof(int), 2, f) != 2) return 4;
  std::vector<uint8_t> img;
  if (!read_vec(f, &img, (size_t)wh[0] * wh[1])) return 4;
  std::fclose(f);
  FrameDev fr(ctx, wh[0], wh[1]);
  Image8 view = {img.data(), wh[0], wh[1], wh[0]};
  if (!fr.preprocessing(view)) return 5;
  int32_t w[3], h[3];
  svs_fastgrid grids[3];
  for (int l = 0; l < 3; ++l) {
    w[l] = wh[0] >> l; h[l] = wh[1] >> l;
    int dim = 3 - (int)(l * 0.5); if (dim < 1) dim = 1;                 // stereo_frontend.cpp:73-88
    double inv = 1.0 / (1 << l);
    int total = (int)(2000 * inv * inv), per_cell = total / (dim * dim), bound = per_cell / 3 > 10 ? per_cell / 3 : 10;
    grids[l] = makeFastGrid(w[l], h[l], per_cell, bound, 25, dim, dim);
  }
  FastGrid fg(ctx, 3, w, h, grids);
  std::vector<Corner> corners[3];
  for (int it = 0; it < 3; ++it)
    if (!fg.detectAdaptively(fr, 6, corners)) return 6;
  for (int l = 0; l < 3; ++l) {
    std::printf("CORNERS %d %zu", l, corners[l].size());
    unsigned long long sum = 0;
    for (size_t i = 0; i < corners[l].size(); ++i) sum = sum * 1000003ull + (unsigned long long)(corners[l][i].x * 4096 + corners[l][i].y);
    std::printf(" %llu", sum);
    std::vector<int32_t> thr = fg.cell_grid2d(l);
    for (size_t i = 0; i < thr.size(); ++i) std::printf(" %d", thr[i]);
    std::printf("\n");
  }
  // ---- SlamGraphBA::optimize ------------------------------------------------------------------------
  f = std::fopen(argv[2], "rb");
  int hdr[4];
  double camd[6];
  if (!f || std::fread(hdr, sizeof(int), 4, f) != 4 || std::fread(camd, sizeof(double), 6, f) != 6) return 7;
  std::vector<double> poses, psi;
  std::vector<svs_ba_edge> edges;
  std::vector<svs_ba_constraint> cons;
  if (!read_vec(f, &poses, (size_t)hdr[0] * 12) || !read_vec(f, &psi, (size_t)hdr[1] * 3) || !read_vec(f, &edges, (size_t)hdr[2]) ||
      !read_vec(f, &cons, (size_t)hdr[3])) return 7;
  std::fclose(f);
  svs_cam cam = {camd[0], camd[1], camd[2], camd[3], (int32_t)camd[4], (int32_t)camd[5]};
  SlamGraphBA ba(ctx);
  svs_ba_stats st;
  const std::vector<double> poses0 = poses, psi0 = psi;
  if (!ba.optimize(OptParams(2, true, 3), cam, &poses, &psi, edges, cons, &st)) return 8;
  // ---- the same window through the id-based marshalling (SlamGraphBA::optimizeWindow: what a SlamGraph::optimize binding calls) and through the
  //      sliding-window form: graph ids instead of indices, xyz_anchor instead of psi, (center, level) instead of (obs, info); shuffled observation order,
  //      two observations from a frame outside the window
  {
    const size_t P = poses0.size() / 12, L = psi0.size() / 3;
    std::vector<int> pose_ids(P), point_ids(L), anchor_ids(L, -1);
    for (size_t i = 0; i < P; ++i) pose_ids[i] = 100 + 3 * (int)i;
    for (size_t i = 0; i < L; ++i) point_ids[i] = 7 + 5 * (int)i;
    std::vector<double> xyz(3 * L);
    for (size_t i = 0; i < L; ++i) { const double *s = &psi0[3 * i]; xyz[3 * i] = s[0] / s[2]; xyz[3 * i + 1] = s[1] / s[2]; xyz[3 * i + 2] = 1. / s[2]; }
    std::vector<SlamGraphBA::ObsById> obs(edges.size());
    for (size_t k = 0; k < edges.size(); ++k) {
      const size_t src = (k * 7919u) % edges.size();              // a permutation (7919 is prime and does not divide the edge count in the test)
      const svs_ba_edge &e = edges[src];
      int level = 0;
      while (level < 3 && e.info[0] < 1.0 / (double)(1 << (2 * level)) * 0.75) ++level;      // info = 4^-level
      obs[k].point_id = point_ids[e.point]; obs[k].pose_id = pose_ids[e.pose]; obs[k].level = level;
      for (int c = 0; c < 3; ++c) obs[k].center[c] = e.obs[c];
      anchor_ids[e.point] = pose_ids[e.anchor];
    }
    for (size_t i = 0; i < L; ++i) if (anchor_ids[i] < 0) anchor_ids[i] = pose_ids[0];
    SlamGraphBA::ObsById stray = obs[0];
    stray.pose_id = 9001; obs.push_back(stray); stray.pose_id = 4; obs.push_back(stray);
    std::vector<SlamGraphBA::ConstraintById> cid(cons.size());
    for (size_t k = 0; k < cons.size(); ++k) {
      cid[k].pose_id_1 = pose_ids[cons[k].pose1]; cid[k].pose_id_2 = pose_ids[cons[k].pose2];
      std::memcpy(cid[k].T_2_from_1, cons[k].T_21, sizeof cons[k].T_21); std::memcpy(cid[k].Lambda_2_from_1, cons[k].info, sizeof cons[k].info);
    }
    double worst[2] = {0, 0};
    for (int variant = 0; variant < 2; ++variant) {
      std::vector<double> pw = poses0, xw = xyz;
      svs_ba_stats sw;
      SlamGraphBA baw(ctx);
      const bool ok = variant == 0 ? baw.optimizeWindow(OptParams(2, true, 3), cam, pose_ids, &pw, point_ids, anchor_ids, &xw, obs, cid, &sw)
                                   : baw.optimizeSlidingWindow(OptParams(2, true, 3), cam, pose_ids, &pw, point_ids, anchor_ids, &xw, obs, cid, &sw);
      if (!ok) return 80 + variant;
      double dp = 0, dl = 0;
      for (size_t i = 0; i < pw.size(); ++i) dp = std::fmax(dp, std::fabs(pw[i] - poses[i]));
      for (size_t i = 0; i < L; ++i) {
        const double *x = &xw[3 * i];
        const double ps[3] = {x[0] / x[2], x[1] / x[2], 1. / x[2]};
        for (int c = 0; c < 3; ++c) dl = std::fmax(dl, std::fabs(ps[c] - psi[3 * i + c]));
      }
      std::printf("BAW %d %d %d %d %.17g %.17g\n", variant, sw.iterations, sw.trials, sw.accepted, dp, dl);
      worst[variant] = dp;
    }
  }
  std::printf("BA %d %d %d %.17g %.17g\n", st.iterations, st.trials, st.accepted, st.chi2_init, st.chi2_final);
  for (size_t i = 0; i < poses.size(); ++i) std::printf("P %.17g\n", poses[i]);
  for (size_t i = 0; i < psi.size(); ++i) std::printf("S %.17g\n", psi[i]);
  // ---- StereoBM (calcDisparityCpu) + BA_SE3_XYZ_STEREO::calcFastMotionOnly: optional third / fourth input files ------
  if (argc >= 4) {
    f = std::fopen(argv[3], "rb");
    std::vector<uint8_t> right;
    if (!f || !read_vec(f, &right, (size_t)wh[0] * wh[1])) return 9;
    std::fclose(f);
    StereoBM bm(ctx, wh[0], wh[1]);
    Image8 rview = {right.data(), wh[0], wh[1], wh[0]};
    std::vector<float> disp;
    if (!bm.ok() || !bm(fr, rview, &fr) || !fr.getDisparity(&disp)) return 9;
    double sum = 0; size_t valid = 0;
    for (size_t i = 0; i < disp.size(); ++i) if (disp[i] >= 0) { sum += disp[i]; ++valid; }
    std::printf("DISP %zu %.17g\n", valid, sum);
  }
  if (argc >= 5) {
    f = std::fopen(argv[4], "rb");
    int n;
    double T[12];
    std::vector<svs_match_result> track;
    if (!f || std::fread(&n, sizeof(int), 1, f) != 1 || std::fread(T, sizeof(double), 12, f) != 12 || !read_vec(f, &track, (size_t)n)) return 10;
    std::fclose(f);
    BA_SE3_XYZ_STEREO ba2(ctx);
    svs_pose_opt_stats ps;
    if (!ba2.calcFastMotionOnly(track, cam, PoseOptimizerParams(true, 2, 15), T, &ps)) return 10;
    std::printf("MOTION %d %.17g %.17g", ps.num_obs, ps.initial_chi2, ps.chi2);
    for (int i = 0; i < 12; ++i) std::printf(" %.17g", T[i]);
    std::printf("\n");
    // processMatchedPoints at the refined pose: every 3rd point on level 1, first third of the records "new"
    std::vector<svs_candidate_point> points(track.size());
    std::memset(points.data(), 0, points.size() * sizeof(svs_candidate_point));
    for (size_t i = 0; i < points.size(); ++i) points[i].anchor_level = (i % 3 == 0) ? 1 : 0;
    std::vector<svs_gated_point> gated;
    svs_point_stats pst;
    if (!ba2.processMatchedPoints(track, points, (int)(track.size() / 3), cam, T, 2.f, &gated, &pst)) return 11;
    int n_acc = 0, n_new = 0;
    for (size_t i = 0; i < gated.size(); ++i) { n_acc += gated[i].accepted; n_new += gated[i].is_new; }
    std::printf("GATE %d %d %d %d %.17g\n", pst.num_obs, pst.num_track_points, n_acc, n_new, pst.sum_track_length);
  }
  // ---- StereoFrontend::processFirstFrame / processFrame (one call per frame) and DenseTrackerGpu::denseTrackingGpu: the image shifted by
  //      two pixels plays the next frame, constant disparity 8 ---------------------------------------------------------------------------
  {
    std::vector<uint8_t> next(img.size());
    for (int y = 0; y < wh[1]; ++y) for (int x = 0; x < wh[0]; ++x) next[(size_t)y * wh[0] + x] = img[(size_t)y * wh[0] + (x + 2 < wh[0] ? x + 2 : wh[0] - 1)];
    std::vector<float> disp((size_t)wh[0] * wh[1], 8.f);
    svs_frontend_params prm = StereoFrontend::referenceParams(false);
    svs_cam fcam = cam;                 // the BA window's intrinsics with this image's size
    fcam.w = wh[0]; fcam.h = wh[1];
    StereoFrontend fe(ctx, fcam, prm, 64, 2);
    if (!fe.ok()) return 12;
    Image8 v0 = {img.data(), wh[0], wh[1], wh[0]}, v1 = {next.data(), wh[0], wh[1], wh[0]};
    ImageF dv = {disp.data(), wh[0], wh[1], wh[0]};
    const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    if (!fe.processFirstFrame(v0, nullptr, &dv) || !fe.keepKeyframe(0, I)) return 12;
    std::vector<svs_candidate_point> none;
    if (!fe.setCandidates(none, 0)) return 12;
    double T[12];
    for (int i = 0; i < 12; ++i) T[i] = I[i];
    svs_frame_result res;
    std::vector<svs_match_result> matches;
    std::vector<svs_gated_point> gated;
    fe.processFrame(v1, nullptr, &dv, T, I, &res, &matches, &gated);        // no candidates: returns false (fewer than 20 matches), the pose is tracked
    std::printf("FRAME %d %d", res.dense_passes, res.n_matched);
    for (int i = 0; i < 12; ++i) std::printf(" %.17g", res.T_cur_from_actkey[i]);
    std::printf("\n");
  }
  return 0;
}
