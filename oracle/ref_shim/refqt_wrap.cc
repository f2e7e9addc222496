// TEST INFRASTRUCTURE -- not part of the product.
// C entry points around the reference's OWN QuadTree<int> (scavislam/quadtree.h, compiled from where it lies under /root/reference
// against the stand-in headers in ref_shim/fake): tests/test_ref_pin_cpu.py checks the oracle's restated quadtree (oracle/vision.c) and
// the quadrant-key order the HIP matcher uses for ties against it.
#include "quadtree.h"
using ScaViSLAM::QuadTree;
using ScaViSLAM::QuadTreeElement;
typedef ScaViSLAM::ALIGNED<QuadTreeElement<int> >::list ElemList;
extern "C" {
void *svs_refqt_create(double x, double y, double w, double h, double delta) { return new QuadTree<int>(ScaViSLAM::Rectangle(x, y, w, h), delta); }
void svs_refqt_destroy(void *t) { delete static_cast<QuadTree<int> *>(t); }
int svs_refqt_insert(void *t, double px, double py, int content) { return static_cast<QuadTree<int> *>(t)->insert(Eigen::Vector2d(px, py), content) ? 1 : 0; }
int svs_refqt_query(void *t, double wx, double wy, double ww, double wh, int *out_xyc, int cap) {
  ElemList l;
  static_cast<QuadTree<int> *>(t)->query(ScaViSLAM::Rectangle(wx, wy, ww, wh), &l);
  int n = 0;
  for (ElemList::const_iterator it = l.begin(); it != l.end() && n < cap; ++it, ++n) {
    out_xyc[3 * n] = (int)it->pos[0]; out_xyc[3 * n + 1] = (int)it->pos[1]; out_xyc[3 * n + 2] = it->content;
  }
  return (int)l.size();
}
int svs_refqt_is_window_empty(void *t, double wx, double wy, double ww, double wh) {
  return static_cast<QuadTree<int> *>(t)->isWindowEmpty(ScaViSLAM::Rectangle(wx, wy, ww, wh)) ? 1 : 0;
}
}
