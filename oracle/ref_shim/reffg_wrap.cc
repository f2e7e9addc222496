// TEST INFRASTRUCTURE -- not part of the product.
// C entry points around the reference's OWN FastGrid (scavislam/fast_grid.{h,cpp}, compiled from where they lie together with this file
// against the stand-in headers in ref_shim/fake) and QuadTree<int>.  The corner detector behind cv::FastFeatureDetector is a hook that the
// test binds to the oracle's FAST-9/16 restatement: compiled from the reference and checked are the cell layout (fast_grid.cpp:23-58), the
// adaptive threshold state machine incl. the prev_thr / prev_prev_thr words shared by the cells of a grid row (:86-152), the static
// detect (:60-83), the corner offsets and the insertion into the quadtree as computeFastCorners does it (stereo_frontend.cpp:657-679).
#include "fast_grid.h"
namespace cv { svs_shim_fast_fn svs_shim_fast_hook = 0; }
using ScaViSLAM::FastGrid;
using ScaViSLAM::QuadTree;
using ScaViSLAM::QuadTreeElement;
typedef ScaViSLAM::ALIGNED<QuadTreeElement<int> >::list ElemList;
static int dump_tree(const QuadTree<int> &qt, int w, int h, int *out_xyc, int cap) {
  ElemList l;
  qt.query(ScaViSLAM::Rectangle(0, 0, w, h), &l);
  int n = 0;
  for (ElemList::const_iterator it = l.begin(); it != l.end() && n < cap; ++it, ++n) {
    out_xyc[3 * n] = (int)it->pos[0]; out_xyc[3 * n + 1] = (int)it->pos[1]; out_xyc[3 * n + 2] = it->content;
  }
  return (int)l.size();
}
extern "C" {
void svs_reffg_set_fast(void *fn) { cv::svs_shim_fast_hook = (cv::svs_shim_fast_fn)fn; }
void *svs_reffg_create(int w, int h, int n_per_cell, int boundary, int fast_thr, int gx, int gy, int fast_min, int fast_max) {
  return new FastGrid(cv::Size(w, h), n_per_cell, boundary, fast_thr, cv::Size(gx, gy), fast_min, fast_max);
}
void svs_reffg_destroy(void *g) { delete static_cast<FastGrid *>(g); }
// detectAdaptively into a fresh QuadTree<int>(Rectangle(0, 0, w, h), 1); returns the number of tree elements, fills (x, y, content) in
// the tree's query order over the whole image
int svs_reffg_detect_adaptively(void *g, const uint8_t *img, int stride, int w, int h, int trials, int *out_xyc, int cap) {
  QuadTree<int> qt(ScaViSLAM::Rectangle(0, 0, w, h), 1);
  cv::Mat m(h, w, CV_8U, const_cast<uint8_t *>(img), (size_t)stride);
  static_cast<FastGrid *>(g)->detectAdaptively(m, trials, &qt);
  return dump_tree(qt, w, h, out_xyc, cap);
}
int svs_reffg_detect(void *g, const uint8_t *img, int stride, int w, int h, int *out_xyc, int cap) {
  QuadTree<int> qt(ScaViSLAM::Rectangle(0, 0, w, h), 1);
  cv::Mat m(h, w, CV_8U, const_cast<uint8_t *>(img), (size_t)stride);
  FastGrid::detect(m, static_cast<FastGrid *>(g)->cell_grid2d(), &qt);
  return dump_tree(qt, w, h, out_xyc, cap);
}
// per cell, row-major: u range, v range, current threshold
int svs_reffg_cells(void *g, int *out, int cap_cells) {
  const ScaViSLAM::CellGrid2d &c = static_cast<FastGrid *>(g)->cell_grid2d();
  int n = 0;
  for (size_t j = 0; j < c.size(); ++j)
    for (size_t i = 0; i < c[j].size(); ++i, ++n)
      if (n < cap_cells) { out[5 * n] = c[j][i].urange.start; out[5 * n + 1] = c[j][i].urange.end; out[5 * n + 2] = c[j][i].vrange.start; out[5 * n + 3] = c[j][i].vrange.end; out[5 * n + 4] = c[j][i].fast_thr; }
  return n;
}
}
