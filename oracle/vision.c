/* vision.c -- oracle (TEST INFRASTRUCTURE ONLY; the reference's own code pinned by oracle/_ref, third-party arithmetic unpinned, see svs_oracle.h) for the
 * per-frame front-end: pyramid, f32+Sobel, FAST-9/16, FastGrid, QuadTree, GuidedMatcher,
 * DenseTracker.  Each function cites the /root/reference/scavislam file:line it follows. */
#include "svs_oracle.h"
#include "svs_math.h"
#include <stdlib.h>

/* ------------------------------------------------------------------------------------------
 * [3rd-party: OpenCV 2.4.2 cv::pyrDown for CV_8U, via cv::buildPyramid at
 *  frame_grabber.cpp:290-292].  Separable [1 4 6 4 1], BORDER_REFLECT_101, integer
 *  accumulate, (sum + 128) >> 8, dst size ((w+1)/2, (h+1)/2).  SURVEY.md A.2. */
static inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
  return i;
}
void svs_ref_pyr_down_u8(const uint8_t *src, int w, int h, int ss, uint8_t *dst, int ds) {
  int dw = (w + 1) / 2, dh = (h + 1) / 2;
  for (int y = 0; y < dh; ++y) {
    for (int x = 0; x < dw; ++x) {
      int acc = 0;
      static const int k[5] = {1, 4, 6, 4, 1};
      for (int r = -2; r <= 2; ++r) {
        const uint8_t *row = src + (size_t)reflect101(2 * y + r, h) * ss;
        int hsum = 0;
        for (int c = -2; c <= 2; ++c) hsum += k[c + 2] * row[reflect101(2 * x + c, w)];
        acc += k[r + 2] * hsum;
      }
      dst[(size_t)y * ds + x] = (uint8_t)((acc + 128) >> 8);
    }
  }
}

/* [3rd-party: convertTo(CV_32F, 1./255.) = (float)u8 * (float)(1/255.) (8u->32f scales in
 *  float), then cv::Sobel(dx=1,dy=0,ksize=1) = I(x+1)-I(x-1), BORDER_REFLECT_101, scale 1;
 *  frame_grabber.cpp:315-333].  The 1/2 is applied by the caller (dense_tracking.cpp:297-302). */
void svs_ref_convert_sobel(const uint8_t *src, int w, int h, int ss, float *img, float *dx,
                           float *dy, int fs) {
  const float sc = (float)(1. / 255.);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) img[(size_t)y * fs + x] = (float)src[(size_t)y * ss + x] * sc;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
      dx[(size_t)y * fs + x] = img[(size_t)y * fs + xp] - img[(size_t)y * fs + xm];
      dy[(size_t)y * fs + x] = img[(size_t)yp * fs + x] - img[(size_t)ym * fs + x];
    }
}

/* ------------------------------------------------------------------------------------------
 * [3rd-party: OpenCV 2.4.2 FAST(img, kps, threshold, nonmax=false), 16-px ring r=3, >= 9
 *  contiguous pixels all < v-t or all > v+t (strict); rows/cols 3..n-4; row-major output.
 *  Called at fast_grid.cpp:72-73,104-105 on the cell ROI.]  SURVEY.md A.1. */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

static int fast_is_corner(const uint8_t *p, int stride, int t) {
  int v = p[0];
  int ring[25];
  for (int k = 0; k < 16; ++k) ring[k] = p[RING_DY[k] * stride + RING_DX[k]];
  for (int k = 16; k < 25; ++k) ring[k] = ring[k - 16];
  int cnt = 0;
  for (int k = 0; k < 25; ++k) { if (ring[k] < v - t) { if (++cnt > 8) return 1; } else cnt = 0; }
  cnt = 0;
  for (int k = 0; k < 25; ++k) { if (ring[k] > v + t) { if (++cnt > 8) return 1; } else cnt = 0; }
  return 0;
}
int svs_ref_fast9_16(const uint8_t *img, int w, int h, int stride, int thr, int16_t *xy, int cap) {
  if (thr < 0) thr = 0;
  if (thr > 255) thr = 255;
  int n = 0;
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x)
      if (fast_is_corner(img + (size_t)y * stride + x, stride, thr)) {
        if (n < cap) { xy[2 * n] = (int16_t)x; xy[2 * n + 1] = (int16_t)y; }
        ++n;
      }
  return n;
}
/* score(p) = max(max_arcs min_k (v-x_k), max_arcs min_k (x_k-v)) - 1; corner at t <=> score>=t */
int svs_ref_fast_score(const uint8_t *img, int stride, int x, int y) {
  const uint8_t *p = img + (size_t)y * stride + x;
  int v = p[0], best = -256;
  for (int s = 0; s < 16; ++s) {
    int mnd = 1 << 20, mnb = 1 << 20;
    for (int k = 0; k < 9; ++k) {
      int r = p[RING_DY[(s + k) & 15] * stride + RING_DX[(s + k) & 15]];
      if (v - r < mnd) mnd = v - r;
      if (r - v < mnb) mnb = r - v;
    }
    if (mnd > best) best = mnd;
    if (mnb > best) best = mnb;
  }
  int sc = best - 1;
  return sc < -1 ? -1 : sc;
}

/* fast_grid.cpp:23-58 (constructor).  int members assigned from double => truncation. */
void svs_ref_fastgrid_init(svs_fastgrid *g, int img_w, int img_h, int n, int b, int fast_thr,
                           int gx, int gy, int fast_min, int fast_max) {
  g->gx = gx; g->gy = gy;
  g->min_inner = (int)(n - b * 0.33);
  g->min_outer = n - b;
  g->max_inner = (int)(n + b * 0.33);
  g->max_outer = n + b;
  g->cell_w = img_w / gx;
  g->cell_h = img_h / gy;
  g->fast_min = fast_min; g->fast_max = fast_max;
  for (int i = 0; i < SVS_MAX_CELLS; ++i) g->thr[i] = fast_thr;
}
/* stereo_frontend.cpp:73-88 */
void svs_ref_fastgrid_init_level(svs_fastgrid *g, int img_w, int img_h, int l) {
  int dim = 3 - (int)(l * 0.5);
  if (dim < 1) dim = 1;
  int num_cells = dim * dim;
  double inv_fac = 1.0 / (double)(1 << l);
  int total = (int)(2000 * inv_fac * inv_fac);
  int per_cell = total / num_cells;
  int bound = per_cell / 3 > 10 ? per_cell / 3 : 10;
  svs_ref_fastgrid_init(g, img_w, img_h, per_cell, bound, 25, dim, dim, 10, 40);
}

/* fast_grid.cpp:86-152, literally: re-runs FAST on the cell ROI for every trial. */
int svs_ref_fastgrid_detect_adaptively(svs_fastgrid *g, const uint8_t *img, int stride,
                                       int trials, int16_t *xy, int cap, int32_t *cell_count,
                                       int32_t *emit_thr) {
  int total = 0;
  int tmp_cap = g->cell_w * g->cell_h;
  int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)(tmp_cap > 0 ? tmp_cap : 1));
  for (int j = 0; j < g->gy; ++j) {
    int prev_thr = -1, prev_prev_thr = -2;            /* :93-94, shared by the row's cells */
    for (int i = 0; i < g->gx; ++i) {
      int *thr = &g->thr[j * g->gx + i];
      int u0 = i * g->cell_w, v0 = j * g->cell_h;
      const uint8_t *roi = img + (size_t)v0 * stride + u0;
      int num = 0, used = *thr;
      for (int trial = 0; trial < trials; ++trial) {
        used = *thr;
        num = svs_ref_fast9_16(roi, g->cell_w, g->cell_h, stride, *thr, tmp, tmp_cap);
        if (prev_prev_thr == *thr) { *thr = (*thr + prev_prev_thr) / 2; break; }
        prev_prev_thr = prev_thr;
        prev_thr = *thr;
        if (num < g->min_inner) {
          if (*thr <= g->fast_min) break;
          --*thr;
          if (num < g->min_outer) {
            if (*thr <= g->fast_min) break;
            --*thr;
            continue;
          }
        } else if (num > g->max_inner) {
          if (*thr >= g->fast_max) break;
          ++*thr;
          if (num > g->max_outer) {
            if (*thr >= g->fast_max) break;
            ++*thr;
            continue;
          }
        }
        break;
      }
      if (trials <= 0) num = 0;
      for (int k = 0; k < num; ++k) {
        if (total < cap) { xy[2 * total] = (int16_t)(tmp[2 * k] + u0); xy[2 * total + 1] = (int16_t)(tmp[2 * k + 1] + v0); }
        ++total;
      }
      if (cell_count) cell_count[j * g->gx + i] = num;
      if (emit_thr) emit_thr[j * g->gx + i] = used;
    }
  }
  free(tmp);
  return total;
}

/* fast_grid.cpp:60-83 */
int svs_ref_fastgrid_detect(const svs_fastgrid *g, const uint8_t *img, int stride, int16_t *xy,
                            int cap, int32_t *cell_count) {
  int total = 0;
  int tmp_cap = g->cell_w * g->cell_h;
  int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)(tmp_cap > 0 ? tmp_cap : 1));
  for (int j = 0; j < g->gy; ++j)
    for (int i = 0; i < g->gx; ++i) {
      int u0 = i * g->cell_w, v0 = j * g->cell_h;
      int num = svs_ref_fast9_16(img + (size_t)v0 * stride + u0, g->cell_w, g->cell_h, stride,
                                 g->thr[j * g->gx + i], tmp, tmp_cap);
      for (int k = 0; k < num; ++k) {
        if (total < cap) { xy[2 * total] = (int16_t)(tmp[2 * k] + u0); xy[2 * total + 1] = (int16_t)(tmp[2 * k + 1] + v0); }
        ++total;
      }
      if (cell_count) cell_count[j * g->gx + i] = num;
    }
  free(tmp);
  return total;
}

/* ------------------------------------------------------------------------------------------
 * QuadTree<int> (quadtree.h:510-544 Children::insert, :614-670 insert, :672-710 query).
 * A leaf holds one element; children order xy, xY, Xy, XY. */
typedef struct qt_node {
  double bx, by, bw, bh;
  struct qt_node *ch[4]; /* xy, xY, Xy, XY */
  int has_children, is_empty;
  double px, py; int content;
} qt_node;
struct svs_ref_qt { qt_node *root; double delta; };

static qt_node *qt_new(double x, double y, double w, double h) {
  qt_node *n = (qt_node *)calloc(1, sizeof(qt_node));
  n->bx = x; n->by = y; n->bw = w; n->bh = h; n->is_empty = 1;
  return n;
}
static void qt_free(qt_node *n) {
  if (!n) return;
  if (n->has_children) for (int i = 0; i < 4; ++i) qt_free(n->ch[i]);
  free(n);
}
static int qt_insert_node(qt_node *n, double px, double py, int c, double delta);
static int qt_children_insert(qt_node *n, double px, double py, int c, double delta) {
  /* quadtree.h:510-544 */
  double rel_x = 1 - (n->bx + n->bw - px) / n->bw;
  double rel_y = 1 - (n->by + n->bh - py) / n->bh;
  if (rel_x < 0.5 && rel_y < 0.5) return qt_insert_node(n->ch[0], px, py, c, delta);
  else if (rel_x >= 0.5 && rel_y < 0.5) return qt_insert_node(n->ch[2], px, py, c, delta);
  else if (rel_x < 0.5 && rel_y >= 0.5) return qt_insert_node(n->ch[1], px, py, c, delta);
  else return qt_insert_node(n->ch[3], px, py, c, delta);
}
static int qt_insert_node(qt_node *n, double px, double py, int c, double delta) {
  if (!n->has_children) {
    if (n->is_empty) { n->px = px; n->py = py; n->content = c; n->is_empty = 0; return 1; }
    double ddx = n->px - px, ddy = n->py - py;
    if (sqrt(ddx * ddx + ddy * ddy) < delta) return 0;          /* :631-634 */
    double x0 = n->bx, x1 = n->bx + n->bw * 0.5, y0 = n->by, y1 = n->by + n->bh * 0.5;
    double w = n->bw * 0.5, h = n->bh * 0.5;
    n->ch[0] = qt_new(x0, y0, w, h); n->ch[1] = qt_new(x0, y1, w, h);
    n->ch[2] = qt_new(x1, y0, w, h); n->ch[3] = qt_new(x1, y1, w, h);
    n->has_children = 1;
    qt_children_insert(n, n->px, n->py, n->content, delta);
    return qt_children_insert(n, px, py, c, delta);
  }
  return qt_children_insert(n, px, py, c, delta);
}
static int rect_intersects(const qt_node *a, double wx, double wy, double ww, double wh) {
  if (a->by + a->bh <= wy) return 0;
  if (a->by >= wy + wh) return 0;
  if (a->bx + a->bw <= wx) return 0;
  if (a->bx >= wx + ww) return 0;
  return 1;
}
static void qt_query_node(const qt_node *n, double wx, double wy, double ww, double wh,
                          int32_t *out, int cap, int *cnt) {
  if (!n->has_children) {
    if (!n->is_empty) {
      /* cv::Rect_::contains: x <= px < x+w */
      if (wx <= n->px && n->px < wx + ww && wy <= n->py && n->py < wy + wh) {
        if (*cnt < cap) { out[3 * *cnt] = (int32_t)n->px; out[3 * *cnt + 1] = (int32_t)n->py; out[3 * *cnt + 2] = n->content; }
        ++*cnt;
      }
    }
  } else {
    for (int i = 0; i < 4; ++i)
      if (rect_intersects(n->ch[i], wx, wy, ww, wh)) qt_query_node(n->ch[i], wx, wy, ww, wh, out, cap, cnt);
  }
}
svs_ref_qt *svs_ref_qt_create(double x, double y, double w, double h, double delta) {
  svs_ref_qt *q = (svs_ref_qt *)malloc(sizeof *q);
  q->root = qt_new(x, y, w, h); q->delta = delta;
  return q;
}
void svs_ref_qt_destroy(svs_ref_qt *q) { if (q) { qt_free(q->root); free(q); } }
int svs_ref_qt_insert(svs_ref_qt *q, double px, double py, int c) { return qt_insert_node(q->root, px, py, c, q->delta); }
void svs_ref_qt_insert_corners(svs_ref_qt *q, const int16_t *xy, const int32_t *cell_count, int n_cells) {
  int k = 0;
  for (int c = 0; c < n_cells; ++c)
    for (int idx = 0; idx < cell_count[c]; ++idx, ++k) qt_insert_node(q->root, (double)xy[2 * k], (double)xy[2 * k + 1], idx, q->delta);
}
int svs_ref_qt_query(const svs_ref_qt *q, double wx, double wy, double ww, double wh, int32_t *out, int cap) {
  int cnt = 0;
  qt_query_node(q->root, wx, wy, ww, wh, out, cap, &cnt);
  return cnt;
}

/* ------------------------------------------------------------------------------------------
 * GuidedMatcher (matcher.cpp).  [3rd-party: VisionTools LinearCamera map/unmap/isInFrame,
 * map(p)=f*p+c, unmap(uv)=(uv-c)/f, isInFrame(uv,b): b<=u<W-b && b<=v<H-b] */
static inline int in_frame(const svs_cam *c, int u, int v, int border) {
  return u >= border && v >= border && u < c->w - border && v < c->h - border;
}
static inline void warp_f(const double *T, double depth, const svs_cam *cam, double ku, double kv, double *o) {
  double p[3] = {depth * ((ku - cam->cx) / cam->f), depth * ((kv - cam->cy) / cam->f), depth * 1.0};
  double q[3];
  pose_act(T, p, q);
  o[0] = cam->f * (q[0] / q[2]) + cam->cx;
  o[1] = cam->f * (q[1] / q[2]) + cam->cy;
}
/* matcher.cpp:403-458 */
void svs_ref_warp_affine(const uint8_t *frame, int stride, const double *T, double depth,
                         const double *key_uv, const svs_cam *cam, int halfpatch, uint8_t *patch) {
  double f0[2], fu[2], fv[2];
  warp_f(T, depth, cam, key_uv[0], key_uv[1], f0);
  warp_f(T, depth, cam, key_uv[0] + 1, key_uv[1], fu);
  warp_f(T, depth, cam, key_uv[0], key_uv[1] + 1, fv);
  double a00 = fu[0] - f0[0], a01 = fu[1] - f0[1], a10 = fv[0] - f0[0], a11 = fv[1] - f0[1];
  /* Eigen 2x2 inverse: adj * (1/det) */
  double invdet = 1.0 / (a00 * a11 - a01 * a10);
  double i00 = a11 * invdet, i01 = -a01 * invdet, i10 = -a10 * invdet, i11 = a00 * invdet;
  int ps = halfpatch * 2;
  for (int ix = 0; ix < ps; ++ix)
    for (int iy = 0; iy < ps; ++iy) {
      double dx = ix - halfpatch, dy = iy - halfpatch;
      double r0 = (i00 * dx + i01 * dy) + key_uv[0];
      double r1 = (i10 * dx + i11 * dy) + key_uv[1];
      double x = floor(r0), y = floor(r1);
      uint8_t val;
      if (!(x >= 0) || !(y >= 0) || x + 1 >= cam->w || y + 1 >= cam->h) val = 0;
      else {
        double sx = r0 - x, sy = r1 - y;
        double wx0 = 1 - sx, wx1 = sx, wy0 = 1 - sy, wy1 = sy;
        int xi = (int)x, yi = (int)y;
        double v00 = frame[(size_t)yi * stride + xi], v01 = frame[(size_t)(yi + 1) * stride + xi];
        double v10 = frame[(size_t)yi * stride + xi + 1], v11 = frame[(size_t)(yi + 1) * stride + xi + 1];
        double s = (wx0 * wy0) * v00 + (wx0 * wy1) * v01 + (wx1 * wy0) * v10 + (wx1 * wy1) * v11;
        val = (uint8_t)(s < 255. ? s : 255.);
      }
      patch[iy * ps + ix] = val;
    }
}
/* matcher.cpp:42-74 (formula reproduced literally, C integer division truncates) */
int svs_ref_znssd(const uint8_t *key, const uint8_t *cur, int sumA, int sumAA) {
  uint32_t sB = 0, sBB = 0, sAB = 0;
  for (int r = 0; r < 64; ++r) { uint8_t c = cur[r]; sB += c; sBB += c * c; sAB += c * key[r]; }
  int sumB = (int)sB, sumBB = (int)sBB, sumAB = (int)sAB;
  return sumAA - 2 * sumAB - sumBB - (sumA * sumA - 2 * sumA * sumB - sumB * sumB) / 64;
}

/* matcher.cpp:144-181 matchCandidates: the candidates in the order the quadtree query returned them, strict '<' (the first of equal
   scores wins); out = {min_dist, index (content of the winner, -1: none), u, v} */
void svs_ref_match_candidates(const uint8_t *cur_img, int cur_stride, const svs_cam *cam, const int32_t *cand_xyc, int nc,
                              const uint8_t *key, int sumA, int sumAA, int init_dist, int *out) {
  int min_dist = init_dist, index = -1, bu = 0, bv = 0;
  for (int k = 0; k < nc; ++k) {
    int cu = cand_xyc[3 * k], cv = cand_xyc[3 * k + 1];
    if (!in_frame(cam, cu, cv, 6)) continue;
    uint8_t cur[64];
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c)
      cur[r * 8 + c] = cur_img[(size_t)(cv - 4 + r) * cur_stride + (cu - 4 + c)];
    int z = svs_ref_znssd(key, cur, sumA, sumAA);
    if (z < min_dist) { min_dist = z; index = cand_xyc[3 * k + 2]; bu = cu; bv = cv; }
  }
  out[0] = min_dist; out[1] = index; out[2] = bu; out[3] = bv;
}

/* matcher.cpp:312-398 for each point (+ computePrediction :98-142, matchCandidates :144-181,
 * returnBestMatch :183-214, createObervation matcher-impl.cpp:33-51). */
void svs_ref_match(const svs_keyframe *kfs, int n_kf, const double *T_cur_from_actkey,
                   const double *T_actkey_from_w, const uint8_t *const cur_pyr[3],
                   const int cur_stride[3], const float *disp, int disp_stride,
                   svs_ref_qt *const feature_tree[3], const svs_cam cam_vec[3],
                   const svs_candidate_point *pts, int n, int R, int thr_mean, int thr_std,
                   svs_match_result *out) {
  double T_w_from_actkey[12], T_cur_from_w[12];
  pose_inv(T_actkey_from_w, T_w_from_actkey);
  pose_mul(T_cur_from_actkey, T_actkey_from_w, T_cur_from_w);
  int qcap = (2 * R + 1) * (2 * R + 1);
  int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)qcap);
  for (int ip = 0; ip < n; ++ip) {
    const svs_candidate_point *ap = &pts[ip];
    svs_match_result *o = &out[ip];
    memset(o, 0, sizeof *o);
    o->znssd = thr_mean * thr_mean * 64;
    if (ap->kf_index < 0 || ap->kf_index >= n_kf) { o->status = SVS_MATCH_NO_ANCHOR; continue; }
    const svs_keyframe *kf = &kfs[ap->kf_index];
    int lvl = ap->anchor_level;
    const svs_cam *cam = &cam_vec[lvl];
    double T_w_from_anchor[12], T_cur_from_anchor[12], xyz_cur[3];
    pose_inv(kf->T_anchor_from_w, T_w_from_anchor);
    pose_mul(T_cur_from_w, T_w_from_anchor, T_cur_from_anchor);
    pose_act(T_cur_from_anchor, ap->xyz_anchor, xyz_cur);
    double uv0 = cam->f * (xyz_cur[0] / xyz_cur[2]) + cam->cx;
    double uv1 = cam->f * (xyz_cur[1] / xyz_cur[2]) + cam->cy;
    if (!in_frame(cam, (int)ap->anchor_obs_pyr[0], (int)ap->anchor_obs_pyr[1], 4)) { o->status = SVS_MATCH_BORDER; continue; }
    double depth_cur = 1. / xyz_cur[2], depth_anchor = 1. / ap->xyz_anchor[2];
    if (depth_cur > depth_anchor * 3 || depth_anchor > depth_cur * 3) { o->status = SVS_MATCH_DEPTH; continue; }
    if (!(fabs(uv0) < 1e9) || !(fabs(uv1) < 1e9)) { o->status = SVS_MATCH_NONE; continue; } /* cast<int> would be UB */
    int ui = (int)uv0, vi = (int)uv1;
    int nc = svs_ref_qt_query(feature_tree[lvl], ui - R, vi - R, 2 * R + 1, 2 * R + 1, cand, qcap);
    uint8_t patch10[100], key[64];
    double key_uv[2] = {ap->anchor_obs_pyr[0], ap->anchor_obs_pyr[1]};
    svs_ref_warp_affine(kf->pyr[lvl], kf->stride[lvl], T_cur_from_anchor, ap->xyz_anchor[2], key_uv, cam, 5, patch10);
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) key[r * 8 + c] = patch10[(r + 1) * 10 + (c + 1)];
    uint32_t sA = 0, sAA = 0;
    for (int r = 0; r < 64; ++r) { sA += key[r]; sAA += key[r] * key[r]; }
    int sumA = (int)sA, sumAA = (int)sAA;
    if (sumA * sumA - sumAA < (int)(thr_std * thr_std * 64)) { o->status = SVS_MATCH_TEXTURE; continue; }
    int best[4];
    svs_ref_match_candidates(cur_pyr[lvl], cur_stride[lvl], cam, cand, nc, key, sumA, sumAA, thr_mean * thr_mean * 64, best);
    int min_dist = best[0], index = best[1], bu = best[2], bv = best[3];
    double T_anchor_from_actkey[12], T_actkey_from_anchor[12];
    pose_mul(kf->T_anchor_from_w, T_w_from_actkey, T_anchor_from_actkey);
    pose_inv(T_anchor_from_actkey, T_actkey_from_anchor);
    pose_act(T_actkey_from_anchor, ap->xyz_anchor, o->xyz_actkey);
    o->znssd = min_dist; o->u = bu; o->v = bv;
    if (index < 0) { o->status = SVS_MATCH_NONE; continue; }
    double inv_factor = 1.0 / (double)(1 << lvl);
    double d = disp[(size_t)(bv << lvl) * disp_stride + (bu << lvl)] * inv_factor;
    if (d > 0) {
      double sc = (double)(1 << lvl);
      float fu = (float)bu, fv = (float)bv;
      o->obs[0] = fu * sc; o->obs[1] = fv * sc; o->obs[2] = (fu - d) * sc;
      o->status = SVS_MATCH_OK;
    } else o->status = SVS_MATCH_NO_DISP;
  }
  free(cand);
}

/* ------------------------------------------------------------------------------------------
 * DenseTracker, CPU path (dense_tracking.cpp:222-423). */
static inline float interp32f(const float *m, int stride, float u, float v) { /* maths_utils.cpp:46-65 */
  float x = floorf(u), y = floorf(v);
  float sx = u - x, sy = v - y;
  float wx0 = 1 - sx, wx1 = sx, wy0 = 1 - sy, wy1 = sy;
  int xi = (int)x, yi = (int)y;
  float v00 = m[(size_t)yi * stride + xi], v01 = m[(size_t)(yi + 1) * stride + xi];
  float v10 = m[(size_t)yi * stride + xi + 1], v11 = m[(size_t)(yi + 1) * stride + xi + 1];
  return (wx0 * wy0) * v00 + (wx0 * wy1) * v01 + (wx1 * wy0) * v10 + (wx1 * wy1) * v11;
}
/* transformations.h:117-139 */
static inline void frame_jac_xyz2uv(const double *p, double f, double *J /* 2x6 */) {
  double x = p[0], y = p[1], z = p[2], z2 = z * z;
  J[0] = -1. / z * f; J[1] = 0; J[2] = x / z2 * f; J[3] = x * y / z2 * f; J[4] = -(1 + (x * x / z2)) * f; J[5] = y / z * f;
  J[6] = 0; J[7] = -1. / z * f; J[8] = y / z2 * f; J[9] = (1 + y * y / z2) * f; J[10] = -x * y / z2 * f; J[11] = -x / z * f;
}
void svs_ref_dense_pass_cpu(const float *cloud, int cw, int ch, const uint8_t *prev_u8, int ps,
                            const float *cur, const float *dxi, const float *dyi, int fs,
                            const svs_cam *cam, const double *T, int do_jac, svs_dense_sums *out,
                            float *rimg) {
  double H[36] = {0}, b[6] = {0};
  float chi2 = 0;
  int64_t nv = 0;
  for (int v = 0; v < ch; ++v)
    for (int u = 0; u < cw; ++u) {
      const float *c4 = cloud + 4 * ((size_t)v * cw + u);
      float r4[4];
      if (c4[3] > 0) {
        double xp[3] = {c4[0], c4[1], c4[2]}, xc[3];
        pose_act(T, xp, xc);
        float uvx = (float)(cam->f * (xc[0] / xc[2]) + cam->cx);
        float uvy = (float)(cam->f * (xc[1] / xc[2]) + cam->cy);
        /* cast<int> of NaN/inf is UB in the reference; treat as out of frame */
        int ok = (fabsf(uvx) < 1e9f && fabsf(uvy) < 1e9f) && in_frame(cam, (int)uvx, (int)uvy, 2);
        if (ok) {
          float ip = (float)((1. / 255.) * prev_u8[(size_t)(v * 4) * ps + u * 4]);
          float ic = interp32f(cur, fs, uvx, uvy);
          float res = ip - ic;
          if (res > 0.1) res = 0.1;
          if (res < -0.1) res = -0.1;
          chi2 += res * res;
          ++nv;
          if (do_jac) {
            float gx = (float)(0.5 * interp32f(dxi, fs, uvx, uvy));
            float gy = (float)(0.5 * interp32f(dyi, fs, uvx, uvy));
            double fj[12], J[6];
            frame_jac_xyz2uv(xc, cam->f, fj);
            for (int k = 0; k < 6; ++k) J[k] = gx * fj[k] + gy * fj[6 + k];
            for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) H[6 * i + j] += J[i] * J[j]; b[i] += J[i] * res; }
            float vv = 1 - 50.f * res * res; if (vv < 0.f) vv = 0.f;
            r4[0] = r4[1] = r4[2] = vv; r4[3] = 1.f;
          }
        } else { r4[0] = 1.f; r4[1] = 0.f; r4[2] = 0.f; r4[3] = 1.f; }
      } else { r4[0] = 0.f; r4[1] = 1.f; r4[2] = 0.f; r4[3] = 1.f; }
      if (do_jac && rimg) memcpy(rimg + 4 * ((size_t)v * cw + u), r4, sizeof r4);
    }
  int k = 0;
  for (int c = 0; c < 6; ++c) for (int r = 0; r <= c; ++r) out->H[k++] = H[6 * r + c];
  for (int i = 0; i < 6; ++i) out->b[i] = b[i];
  out->chi2 = chi2; out->n_valid = nv;
}

int svs_ref_dense_tracking_cpu(const float *const cloud[3], const uint8_t *const prev_u8[3],
                               const int pstride[3], const float *const cur[3],
                               const float *const dx[3], const float *const dy[3],
                               const int fstride[3], const svs_cam cam_vec[3], double *T) {
  return svs_ref_dense_tracking_cpu_rimg(cloud, prev_u8, pstride, cur, dx, dy, fstride, cam_vec, T, 0);
}
/* same, also leaving DenseTracker::residual_img[level] as the reference does: every H,b pass rewrites it
   (dense_tracking.cpp:319-329), so it ends as the image of the LAST H,b pass of each level */
int svs_ref_dense_tracking_cpu_rimg(const float *const cloud[3], const uint8_t *const prev_u8[3],
                                    const int pstride[3], const float *const cur[3],
                                    const float *const dx[3], const float *const dy[3],
                                    const int fstride[3], const svs_cam cam_vec[3], double *T,
                                    float *const rimg[3]) {
  return svs_ref_dense_tracking_cpu_rec(cloud, prev_u8, pstride, cur, dx, dy, fstride, cam_vec, T, rimg, 0, 0, 0);
}
/* same + the accept / reject record: rec[4 k ..] = {level, accepted (1 / 0; 2 = the level's initial chi2), chi2, new_chi2} */
int svs_ref_dense_tracking_cpu_rec(const float *const cloud[3], const uint8_t *const prev_u8[3],
                                   const int pstride[3], const float *const cur[3],
                                   const float *const dx[3], const float *const dy[3],
                                   const int fstride[3], const svs_cam cam_vec[3], double *T,
                                   float *const rimg[3], double *rec, int rec_cap, int *n_rec) {
  int passes = 0, nr = 0;
#define SVS_REC(l, a, c0, c1) do { if (rec && nr < rec_cap) { rec[4 * nr] = (l); rec[4 * nr + 1] = (a); rec[4 * nr + 2] = (c0); rec[4 * nr + 3] = (c1); } ++nr; } while (0)
  for (int level = 2; level >= 0; --level) {
    const svs_cam *cam = &cam_vec[level];
    int cw = cam->w / 4, ch = cam->h / 4;
    svs_dense_sums s;
    svs_ref_dense_pass_cpu(cloud[level], cw, ch, prev_u8[level], pstride[level], cur[level], dx[level], dy[level], fstride[level], cam, T, 0, &s, 0);
    ++passes;
    float chi2 = (float)s.chi2;
    SVS_REC(level, 2, chi2, chi2);
    double nu = 2, mu = 0.01f; int stop = 0, trial = 0;
    (void)nu; (void)mu;
    for (int i = 0; i < 15; ++i) {
      double rho = 0;
      do {
        svs_ref_dense_pass_cpu(cloud[level], cw, ch, prev_u8[level], pstride[level], cur[level], dx[level], dy[level], fstride[level], cam, T, 1, &s, rimg ? rimg[level] : 0);
        ++passes;
        double Hf[36], nb[6], x[6], E[12], Tn[12];
        int k = 0;
        for (int c = 0; c < 6; ++c) for (int r = 0; r <= c; ++r) { Hf[6 * r + c] = s.H[k]; Hf[6 * c + r] = s.H[k]; ++k; }
        for (int q = 0; q < 6; ++q) nb[q] = -s.b[q];
        solve_small(6, Hf, nb, x);                      /* H.ldlt().solve(-Jres), no damping (:332) */
        se3_exp(x, E);
        pose_mul(E, T, Tn);
        svs_dense_sums s2;
        svs_ref_dense_pass_cpu(cloud[level], cw, ch, prev_u8[level], pstride[level], cur[level], dx[level], dy[level], fstride[level], cam, Tn, 0, &s2, 0);
        ++passes;
        float new_chi2 = (float)s2.chi2;
        rho = chi2 - new_chi2;
        SVS_REC(level, rho > 0 ? 1 : 0, chi2, new_chi2);
        if (rho > 0) {
          memcpy(T, Tn, sizeof(double) * 12);
          chi2 = new_chi2;
          double mx = -1; for (int q = 0; q < 6; ++q) if (fabs(x[q]) > mx) mx = fabs(x[q]);
          stop = mx <= 1e-10;
          trial = 0;
        } else {
          ++trial;
          if (trial == 2) stop = 1;
        }
      } while (!(rho > 0 || stop));
      if (stop) break;
    }
  }
#undef SVS_REC
  if (n_rec) *n_rec = nr;
  return passes;
}

/* dense_tracking.cpp:393-423 + stereo_camera.cpp:24-34 (Q) + maths_utils.cpp:36-44 */
void svs_ref_pointcloud_cpu(const float *disp, int ds, const svs_cam *cam, int level,
                            const double *T_cur_from_actkey, float *cloud) {
  double Ti[12];
  pose_inv(T_cur_from_actkey, Ti);
  /* TQ = T^-1(4x4) * Q ; Q rows: [1 0 0 -cx; 0 1 0 -cy; 0 0 0 f; 0 0 1/b 0] */
  double Q[16] = {1, 0, 0, -cam->cx, 0, 1, 0, -cam->cy, 0, 0, 0, cam->f, 0, 0, 1.0 / cam->b, 0};
  double TQ[16];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
    double s = 0;
    for (int k = 0; k < 4; ++k) { double tik = i < 3 ? Ti[4 * i + k] : (k == 3 ? 1.0 : 0.0); s += tik * Q[4 * k + j]; }
    TQ[4 * i + j] = s;
  }
  int cw = cam->w / 4, ch = cam->h / 4;
  double inv_factor = 1.0 / (double)(1 << level);
  for (int v = 0; v < ch; ++v)
    for (int u = 0; u < cw; ++u) {
      float d = (float)(disp[(size_t)((v * 4) << level) * ds + ((u * 4) << level)] * inv_factor);
      float *o = cloud + 4 * ((size_t)v * cw + u);
      if (d <= 0) { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = -1.f; }
      else {
        double uvd[4] = {(double)(u * 4), (double)(v * 4), d, 1.0}, r[4];
        for (int i = 0; i < 4; ++i) r[i] = TQ[4 * i] * uvd[0] + TQ[4 * i + 1] * uvd[1] + TQ[4 * i + 2] * uvd[2] + TQ[4 * i + 3] * uvd[3];
        o[0] = (float)(r[0] / r[3]); o[1] = (float)(r[1] / r[3]); o[2] = (float)(r[2] / r[3]); o[3] = 1.f;
      }
    }
}

/* ------------------------------------------------------------------------------------------
 * Full-resolution f32 variant (the CUDA build): gpu/dense_tracking.cu:24-80 (helpers), :172-263
 * (jacobianReduction_kernel), :376-453 (chi2_kernel), :495-541 (residualImage_kernel) and the host loop
 * DenseTracker::denseTrackingGpu (dense_tracking.cpp:60-193).
 *
 * PINNED against the reference itself: oracle/_ref/libsvs_ref_gpu.so is compiled from those two reference files
 * (oracle/Makefile) and tests/test_ref_pin_cpu.py checks this restatement against it bit for bit -- per pixel, per
 * 8x8 block and for whole images in SVS_SUM_F32_TREE mode (the reference's own f32 reduction order).
 *
 * Texture fetch: the reference samples tex2D(tex, uv.x + 0.5f, uv.y + 0.5f) with linear filtering (.cu:206-215).
 * The texture unit subtracts the 0.5 again (CUDA C Programming Guide, linear filtering: xB = x - 0.5), so the
 * sampling position is RN(uv + 0.5f) - 0.5f -- NOT always uv: the f32 addition rounds when uv + 0.5 crosses a binade
 * (uv in [2^k - 0.5, 2^k)).  That rounding is the reference's own arithmetic and is reproduced; the four taps are
 * combined as in the reference's software bilinear (maths_utils.cpp:46-65).  NVIDIA's 9-bit fixed-point weights are
 * device behaviour and are not reproduced (oracle/_ref can switch them on to measure their effect). */
static inline float tex2d_lin(const float *m, int stride, int w, int h, float x, float y) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fi = floorf(xb), fj = floorf(yb);
  const float sx = xb - fi, sy = yb - fj;
  const float wx0 = 1 - sx, wx1 = sx, wy0 = 1 - sy, wy1 = sy;
  int i0 = (int)fi, j0 = (int)fj, i1 = i0 + 1, j1 = j0 + 1;
  i0 = i0 < 0 ? 0 : (i0 >= w ? w - 1 : i0); i1 = i1 < 0 ? 0 : (i1 >= w ? w - 1 : i1);   /* clamp addressing */
  j0 = j0 < 0 ? 0 : (j0 >= h ? h - 1 : j0); j1 = j1 < 0 ? 0 : (j1 >= h ? h - 1 : j1);
  const float v00 = m[(size_t)j0 * stride + i0], v01 = m[(size_t)j1 * stride + i0];
  const float v10 = m[(size_t)j0 * stride + i1], v11 = m[(size_t)j1 * stride + i1];
  return (wx0 * wy0) * v00 + (wx0 * wy1) * v01 + (wx1 * wy0) * v10 + (wx1 * wy1) * v11;
}
/* one thread of jacobianReduction_kernel / chi2_kernel up to the reduction: returns 1 and fills res (and J if do_jac)
   when the pixel contributes (.cu:193-221 / :395-411) */
static inline int full_pixel(const float *cloud, int w, int h, int s4, const float *prev, const float *cur,
                             const float *dxi, const float *dyi, int fs, float f, float cx, float cy, const float *T,
                             int u, int v, int do_jac, float *res_out, float *J) {
  const float *p = cloud + 4 * ((size_t)v * s4 + u);
  if (!(p[3] > 0)) return 0;
  const float x = p[0] * T[0] + p[1] * T[3] + p[2] * T[6] + p[3] * T[9];       /* matTimesVec / dotStride3 */
  const float y = p[0] * T[1] + p[1] * T[4] + p[2] * T[7] + p[3] * T[10];
  const float z = p[0] * T[2] + p[1] * T[5] + p[2] * T[8] + p[3] * T[11];
  const float uu = f * x / z + cx, vv = f * y / z + cy;                        /* cameraProject */
  if (!(uu >= 1.f && vv >= 1.f && uu <= (float)(w - 2) && vv <= (float)(h - 2))) return 0;
  const float ut = uu + 0.5f, vt = vv + 0.5f;                                  /* uv_cur_texoffset */
  const float ip = prev[(size_t)v * fs + u];
  const float ic = tex2d_lin(cur, fs, w, h, ut, vt);
  const float res = ip - ic;
  *res_out = res;
  if (do_jac) {
    float gx = 0.5f * tex2d_lin(dxi, fs, w, h, ut, vt), gy = 0.5f * tex2d_lin(dyi, fs, w, h, ut, vt);
    const float zsq = z * z;                                                   /* frameJacobian, .cu:65-80 */
    gx *= f; gy *= f;
    J[0] = (float)(-gx * (1. / z));
    J[1] = (float)(-gy * 1. / z);
    J[2] = (gx * x / zsq + gy * y / zsq);
    J[3] = (gx * (x * y) / zsq + gy * (1.f + y * y / zsq));
    J[4] = (-gx * (1.f + (x * x / zsq)) - gy * (x * y) / zsq);
    J[5] = (gx * y / z - gy * x / z);
  }
  return 1;
}
/* per-pixel terms of jacobianReduction_kernel before its reduction: out[v * w + u] = {J0..J5, res, valid} (zeros where the
   pixel does not contribute) -- what the HIP path's svs_dense_pixel_terms_full must reproduce bit for bit */
void svs_ref_dense_pixel_terms_full(const float *cloud, int w, int h, int s4, const float *prev, const float *cur,
                                    const float *dxi, const float *dyi, int fs, float f, float cx, float cy, const float *T,
                                    float *out) {
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      float res = 0, J[6] = {0, 0, 0, 0, 0, 0};
      const int ok = full_pixel(cloud, w, h, s4, prev, cur, dxi, dyi, fs, f, cx, cy, T, u, v, 1, &res, J);
      float *o = out + 8 * ((size_t)v * w + u);
      for (int i = 0; i < 6; ++i) o[i] = ok ? J[i] : 0.f;
      o[6] = ok ? res : 0.f; o[7] = ok ? 1.f : 0.f;
    }
}
/* sum_mode SVS_SUM_F64 (0): f32 per-pixel products accumulated in f64, row-major (what the product is compared with);
   sum_mode SVS_SUM_F32_TREE (1): the reference's arithmetic to the last bit -- per 8x8 block a 64-slot f32 array
   reduced as `s[t] += s[t+off]`, off = 32,16,...,1, lanes t < 32 in lockstep and only lanes inside the image taking
   part (.cu:153-168,224-261), then the block results added sequentially in f32 on the host in block order (.cu:343-355,
   :480-485). */
void svs_ref_dense_pass_full_ex(const float *cloud, int w, int h, int s4, const float *prev,
                                const float *cur, const float *dxi, const float *dyi, int fs, float f,
                                float cx, float cy, const float *T, int do_jac, int sum_mode, svs_dense_sums *out) {
  int64_t nv = 0;
  if (sum_mode == 0) {
    double H[21] = {0}, b[6] = {0}, chi2 = 0;
    for (int v = 0; v < h; ++v)
      for (int u = 0; u < w; ++u) {
        float res, J[6];
        if (!full_pixel(cloud, w, h, s4, prev, cur, dxi, dyi, fs, f, cx, cy, T, u, v, do_jac, &res, J)) continue;
        chi2 += (double)(res * res);
        ++nv;
        if (do_jac) {
          int k = 0;
          for (int c = 0; c < 6; ++c) for (int r = 0; r <= c; ++r) H[k++] += (double)(J[c] * J[r]);
          for (int i = 0; i < 6; ++i) b[i] += (double)(J[i] * res);
        }
      }
    memcpy(out->H, H, sizeof H); memcpy(out->b, b, sizeof b); out->chi2 = chi2; out->n_valid = nv;
    return;
  }
  float Ht[21] = {0}, bt[6] = {0}, chi2t = 0.f;
  const int gx_ = (w + 7) / 8, gy_ = (h + 7) / 8;
  for (int by = 0; by < gy_; ++by)
    for (int bx = 0; bx < gx_; ++bx) {
      float s[64][28];           /* 21 H, 6 b, chi2 per thread */
      int active[64];
      memset(s, 0, sizeof s);
      for (int t = 0; t < 64; ++t) {
        const int u = bx * 8 + (t & 7), v = by * 8 + (t >> 3);
        active[t] = u < w && v < h;
        if (!active[t]) continue;
        float res, J[6];
        if (!full_pixel(cloud, w, h, s4, prev, cur, dxi, dyi, fs, f, cx, cy, T, u, v, do_jac, &res, J)) continue;
        ++nv;
        s[t][27] += res * res;
        if (do_jac) {
          int k = 0;
          for (int c = 0; c < 6; ++c) for (int r = 0; r <= c; ++r) s[t][k++] += J[c] * J[r];   /* addOuter, .cuh:163-203 */
          for (int i = 0; i < 6; ++i) s[t][21 + i] += J[i] * res;                              /* scaledAdd */
        }
      }
      for (int off = 32; off >= 1; off >>= 1)
        for (int t = 0; t < 32; ++t)            /* ascending t == lockstep: lane t reads slot t+off before lane t+off writes it */
          if (active[t]) for (int k = 0; k < 28; ++k) s[t][k] += s[t + off][k];
      for (int k = 0; k < 21; ++k) Ht[k] += s[0][k];
      for (int k = 0; k < 6; ++k) bt[k] += s[0][21 + k];
      chi2t += s[0][27];
    }
  for (int k = 0; k < 21; ++k) out->H[k] = Ht[k];
  for (int k = 0; k < 6; ++k) out->b[k] = bt[k];
  out->chi2 = chi2t; out->n_valid = nv;
}
void svs_ref_dense_pass_full(const float *cloud, int w, int h, int s4, const float *prev,
                             const float *cur, const float *dxi, const float *dyi, int fs, float f,
                             float cx, float cy, const float *T, int do_jac, svs_dense_sums *out) {
  svs_ref_dense_pass_full_ex(cloud, w, h, s4, prev, cur, dxi, dyi, fs, f, cx, cy, T, do_jac, 0, out);
}
/* gpu/dense_tracking.cu:495-541 residualImage_kernel */
void svs_ref_residual_image_full(const float *cloud, int w, int h, int s4, const float *prev,
                                 const float *cur, int fs, float f, float cx, float cy, const float *T,
                                 float *rimg) {
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      const float *p = cloud + 4 * ((size_t)v * s4 + u);
      float *o = rimg + 4 * ((size_t)v * s4 + u);
      if (!(p[3] > 0)) { o[0] = 0.f; o[1] = 1.f; o[2] = 0.f; o[3] = 1.f; continue; }
      float res, J[6];
      if (!full_pixel(cloud, w, h, s4, prev, cur, 0, 0, fs, f, cx, cy, T, u, v, 0, &res, J)) { o[0] = 1.f; o[1] = 0.f; o[2] = 0.f; o[3] = 1.f; continue; }
      float g = 1 - 50.f * res * res; if (g < 0.f) g = 0.f;
      o[0] = o[1] = o[2] = g; o[3] = 1.f;
    }
}

/* DenseTracker::denseTrackingGpu (dense_tracking.cpp:60-193): levels 2..0; chi2(); <= 15 iterations of
   { jacobianReduction; H += mu diag(H); x = H.ldlt().solve(-b); T_new = exp(x) T; new_chi2 = chi2(T_new);
     rho = chi2 - new_chi2 (float arithmetic); accept: stop = |b|_inf <= EPS, mu *= max(1/3, 1 - (2 rho - 1)^3), nu = 2,
     trial = 0 | reject: mu *= nu, nu *= 2, two rejections in a row stop } while (!(rho > 0 || stop)).
   Poses go to the kernels as GpuMatrix34 (f64 -> f32, column-major, :80-82,109-111,138-140).
   rec (optional, cap records of 4 doubles): {level, accepted (1/0; 2 = the level's initial chi2), chi2 before, chi2 of
   the trial}; T_jac (optional [3][12]): the pose of the last jacobianReduction of each level, which is the pose the
   reference renders residualImage with (:177-186 pass gpuT_cur_from_prev, not the final pose).  Returns the number of
   kernel passes the reference would have launched. */
/* force (optional, n_force entries indexed like rec): -1 = the loop's own decision, 0 / 1 = take this decision at that record instead (tests: a near-tie of
   `chi2 - new_chi2` may legitimately fall the other way in another summation order -- the test then follows that branch through the SAME loop) */
int svs_ref_dense_tracking_gpu_forced(const float *const cloud[3], const int stride4[3], const float *const prev[3],
                               const float *const cur[3], const float *const dx[3], const float *const dy[3],
                               const int fstride[3], const int w[3], const int h[3], const double f[3],
                               const double cx[3], const double cy[3], double *T, int sum_mode, double *rec, int rec_cap,
                               int *n_rec, double *T_jac, const int *force, int n_force) {
  int passes = 0, nr = 0;
#define SVS_REC(l, a, c0, c1) do { if (rec && nr < rec_cap) { rec[4 * nr] = (l); rec[4 * nr + 1] = (a); rec[4 * nr + 2] = (c0); rec[4 * nr + 3] = (c1); } ++nr; } while (0)
  for (int l = 2; l >= 0; --l) {
    const float fl = (float)f[l], cxl = (float)cx[l], cyl = (float)cy[l];       /* GpuIntrinsics::set, .cuh:30-38 */
    float Tf[12];
    svs_dense_sums s;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) Tf[3 * c + r] = (float)T[4 * r + c];
    svs_ref_dense_pass_full_ex(cloud[l], w[l], h[l], stride4[l], prev[l], cur[l], dx[l], dy[l], fstride[l], fl, cxl, cyl, Tf, 0, sum_mode, &s);
    ++passes;
    float chi2 = (float)s.chi2;
    SVS_REC(l, 2, chi2, chi2);
    double nu = 2, mu = 0.01f;
    int stop = 0, trial = 0;
    for (int i = 0; i < 15; ++i) {
      double rho = 0;
      do {
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) Tf[3 * c + r] = (float)T[4 * r + c];
        if (T_jac) memcpy(T_jac + 12 * l, T, sizeof(double) * 12);
        svs_ref_dense_pass_full_ex(cloud[l], w[l], h[l], stride4[l], prev[l], cur[l], dx[l], dy[l], fstride[l], fl, cxl, cyl, Tf, 1, sum_mode, &s);
        ++passes;
        double Hf[36], nb[6], x[6], E[12], Tn[12];
        int k = 0;
        for (int c = 0; c < 6; ++c) for (int r = 0; r <= c; ++r) { Hf[6 * r + c] = s.H[k]; Hf[6 * c + r] = s.H[k]; ++k; }   /* copyTo, .cuh:118-132 */
        for (int q = 0; q < 6; ++q) Hf[7 * q] += mu * Hf[7 * q];                                                       /* H += mu diag(H) */
        for (int q = 0; q < 6; ++q) nb[q] = -s.b[q];
        solve_small(6, Hf, nb, x);
        se3_exp(x, E);
        pose_mul(E, T, Tn);
        float Tnf[12];
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) Tnf[3 * c + r] = (float)Tn[4 * r + c];
        svs_dense_sums s2;
        svs_ref_dense_pass_full_ex(cloud[l], w[l], h[l], stride4[l], prev[l], cur[l], dx[l], dy[l], fstride[l], fl, cxl, cyl, Tnf, 0, sum_mode, &s2);
        ++passes;
        const float new_chi2 = (float)s2.chi2;
        rho = chi2 - new_chi2;                      /* float - float, then widened (:142) */
        if (force && nr < n_force && force[nr] >= 0) rho = force[nr] ? fabs(rho) + 1e-30 : -fabs(rho);      /* (tests only) */
        SVS_REC(l, rho > 0 ? 1 : 0, chi2, new_chi2);
        if (rho > 0) {
          memcpy(T, Tn, sizeof(double) * 12);
          chi2 = new_chi2;
          double mx = 0; for (int q = 0; q < 6; ++q) if (fabs(s.b[q]) > mx) mx = fabs(s.b[q]);
          stop = mx <= 1e-10;                       /* norm_max(b) <= EPS (global.h:106) */
          const double t = 2 * rho - 1;
          const double g = 1 - t * t * t;
          mu *= (1. / 3. > g ? 1. / 3. : g);
          nu = 2.;
          trial = 0;
        } else {
          mu *= nu;
          nu *= 2.;
          ++trial;
          if (trial == 2) stop = 1;
        }
      } while (!(rho > 0 || stop));
      if (stop) break;
    }
    ++passes;                                       /* residualImage(gpuT_cur_from_prev) (:177-186) */
  }
#undef SVS_REC
  if (n_rec) *n_rec = nr;
  return passes;
}
int svs_ref_dense_tracking_gpu(const float *const cloud[3], const int stride4[3], const float *const prev[3],
                               const float *const cur[3], const float *const dx[3], const float *const dy[3],
                               const int fstride[3], const int w[3], const int h[3], const double f[3],
                               const double cx[3], const double cy[3], double *T, int sum_mode, double *rec, int rec_cap,
                               int *n_rec, double *T_jac) {
  return svs_ref_dense_tracking_gpu_forced(cloud, stride4, prev, cur, dx, dy, fstride, w, h, f, cx, cy, T, sum_mode, rec, rec_cap, n_rec, T_jac, 0, 0);
}

/* FrameGrabber::preprocessing, CUDA build (frame_grabber.cpp:291-313): level 0 = gpu convertTo(CV_32F, 1/255.),
   levels 1, 2 = cv::gpu::pyrDown of the f32 level above, and on every level dx / dy = the gpu derivative filters
   created at :102-115 (ksize 1, cv::BORDER_REPLICATE).  OpenCV 2.4.2 gpu module is external -- ASSUMED semantics:
   convertTo multiplies in float; pyrDown = separable [1 4 6 4 1]/16 in f32 (weights .0625 .25 .375 .25 .0625),
   columns (vertical) first then rows, taps added in ascending order, BORDER_REFLECT_101, output ((w+1)/2, (h+1)/2)
   sampled at (2x, 2y); derivative = I(x+1) - I(x-1) with clamped coordinates, no scaling. */
static inline int refl101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; } return i; }
void svs_ref_pyr_down_f32(const float *src, int w, int h, int ss, float *dst, int ds) {
  const int ow = (w + 1) / 2, oh = (h + 1) / 2;
  const float k0 = 0.0625f, k1 = 0.25f, k2 = 0.375f;
  float *col = (float *)malloc(sizeof(float) * (size_t)w);
  for (int y = 0; y < oh; ++y) {
    const float *r0 = src + (size_t)refl101(2 * y - 2, h) * ss, *r1 = src + (size_t)refl101(2 * y - 1, h) * ss;
    const float *r2 = src + (size_t)refl101(2 * y, h) * ss, *r3 = src + (size_t)refl101(2 * y + 1, h) * ss;
    const float *r4 = src + (size_t)refl101(2 * y + 2, h) * ss;
    for (int x = 0; x < w; ++x) { float a = k0 * r0[x]; a = a + k1 * r1[x]; a = a + k2 * r2[x]; a = a + k1 * r3[x]; a = a + k0 * r4[x]; col[x] = a; }
    for (int x = 0; x < ow; ++x) {
      float a = k0 * col[refl101(2 * x - 2, w)]; a = a + k1 * col[refl101(2 * x - 1, w)]; a = a + k2 * col[refl101(2 * x, w)];
      a = a + k1 * col[refl101(2 * x + 1, w)]; a = a + k0 * col[refl101(2 * x + 2, w)];
      dst[(size_t)y * ds + x] = a;
    }
  }
  free(col);
}
void svs_ref_deriv_replicate(const float *img, int w, int h, int s, float *dx, float *dy, int ds) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1, ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
      dx[(size_t)y * ds + x] = img[(size_t)y * s + xp] - img[(size_t)y * s + xm];
      dy[(size_t)y * ds + x] = img[(size_t)yp * s + x] - img[(size_t)ym * s + x];
    }
}
void svs_ref_convert_f32(const uint8_t *src, int w, int h, int ss, float *dst, int ds) {
  const float sc = (float)(1. / 255.);
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) dst[(size_t)y * ds + x] = (float)src[(size_t)y * ss + x] * sc;
}
void svs_ref_pointcloud_full(const float *TQ, const float *disp, int w, int h, int si, int so,
                             int factor, float *cloud) {
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      float d = disp[(size_t)v * si + u * factor] * factor;   /* row NOT scaled: .cu:97-98 */
      float *o = cloud + 4 * ((size_t)v * so + u);
      if (d <= 0) { o[0] = o[1] = o[2] = 0.f; o[3] = -1.f; continue; }
      float q[4] = {(float)u, (float)v, d, 1.f}, r[4];
      for (int i = 0; i < 4; ++i) r[i] = q[0] * TQ[i] + q[1] * TQ[4 + i] + q[2] * TQ[8 + i] + q[3] * TQ[12 + i];
      o[0] = r[0] / r[3]; o[1] = r[1] / r[3]; o[2] = r[2] / r[3]; o[3] = 1.f;
    }
}

void svs_ref_se3_exp(const double *x, double *T) { se3_exp(x, T); }
void svs_ref_se3_log(const double *T, double *x) { se3_log(T, x); }
void svs_ref_se3_mul(const double *A, const double *B, double *C) { pose_mul(A, B, C); }
void svs_ref_se3_inv(const double *A, double *B) { pose_inv(A, B); }

/* ---- PoseOptimizer<SE3,6,IdObs<3>,3>::calcFastMotionOnly (pose_optimizer.h:134-298) -------------------
 * with the SE3XYZ_STEREO prediction (transformations.h:414-464): f = obs - map_uvu(T xyz), J_c = frameJac
 * (= d f / d delta), unweighted normal equations A = mu I + sum J^T J, B = -sum J^T (w f) with the
 * pseudo-Huber weight applied to f only (:169-175, :213-222), delta = A.ldlt().solve(B), T_new = exp(delta) T,
 * gain test on rho = chi2 - new_chi2 (:266), mu update mu *= max(1/3, 1 - (2 rho - 1)^3) (sic, rho is not
 * a ratio, :272), 5 consecutive rejections stop, stop also when |B|_max <= EPS (1e-10, global.h:106).
 * The observation list is the matcher's TrackData (status OK entries of `res`, in order):
 * stereo_frontend.cpp:1058-1063 calls it with PoseOptimizerParams(true, 2, 15). */
static double mo_kernel(double delta, double b) {          /* pose_optimizer.h:426-435 */
  const double a = fabs(delta);
  return a < b ? delta * delta : 2 * b * a - b * b;
}
static void mo_residual(const double *T, const double *xyz, const double *obs, const svs_cam *cam, double *f, double *J) {
  double p[3];
  pose_act(T, xyz, p);
  const double x = p[0], y = p[1], z = p[2], fl = cam->f;
  f[0] = obs[0] - (x / z * fl + cam->cx);                  /* StereoCamera::map_uvu, stereo_camera.cpp:37-44 */
  f[1] = obs[1] - (y / z * fl + cam->cy);
  f[2] = obs[2] - ((x - cam->b) / z * fl + cam->cx);
  if (J) {                                                  /* SE3XYZ_STEREO::frameJac, transformations.h:424-447 */
    const double ibz = 1. / z, ibz2 = 1. / (z * z);
    const double A = -fl * ibz, B = -fl * ibz, C = fl * x * ibz2, D = fl * y * ibz2, E = fl * (x - cam->b) * ibz2;
    const double Jv[18] = {A, 0, C, y * C, z * A - x * C, -y * A,
                           0, B, D, -z * B + y * D, -x * D, x * B,
                           A, 0, E, y * E, z * A - x * E, -y * A};
    memcpy(J, Jv, sizeof Jv);
  }
}
static double mo_weighted_sq(double *f, int robust, double b) {      /* f *= w; returns sqrW(f) after weighting */
  if (robust) {
    double nrm = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    if (nrm < 1e-10) nrm = 1e-10;
    const double w = sqrt(mo_kernel(nrm, b)) / nrm;
    f[0] *= w; f[1] *= w; f[2] *= w;
  }
  return f[0] * f[0] + f[1] * f[1] + f[2] * f[2];
}
int svs_ref_motion_only(const svs_match_result *res, int n, const svs_cam *cam, const svs_pose_opt_params *prm, double *T_io,
                        svs_pose_opt_stats *st) {
  const double EPS = 1e-10;
  double T[12], Tn[12], chi2 = 0, max_err = 0, norm_max_A = 0, nu = 2;
  int num_obs = 0, stop = 0, trial = 0;
  memcpy(T, T_io, sizeof T);
  for (int i = 0; i < n; ++i) {
    if (res[i].status != SVS_MATCH_OK) continue;
    double f[3], J[18];
    mo_residual(T, res[i].xyz_actkey, res[i].obs, cam, f, J);
    for (int c = 0; c < 6; ++c) {
      const double dgl = fabs(J[c] * J[c] + J[6 + c] * J[6 + c] + J[12 + c] * J[12 + c]);
      if (dgl > norm_max_A) norm_max_A = dgl;
    }
    chi2 += mo_weighted_sq(f, prm->robust_kernel, prm->kernel_param);
    ++num_obs;
    for (int k = 0; k < 3; ++k) if (fabs(f[k]) > max_err) max_err = fabs(f[k]);
  }
  st->initial_chi2 = chi2; st->num_obs = num_obs; st->status = 0;
  if (num_obs == 0) { st->chi2 = 0; st->max_err = 0; st->status = 1; return 1; }      /* assert(obs_list.size()>0) */
  double mu = prm->initial_mu == -1 ? prm->tau * norm_max_A : prm->initial_mu;
  for (int ig = 0; ig < prm->num_iter; ++ig) {
    double rho = 0;
    do {
      double A[36] = {0}, B[6] = {0}, delta[6];
      for (int c = 0; c < 6; ++c) A[7 * c] = mu;
      for (int i = 0; i < n; ++i) {
        if (res[i].status != SVS_MATCH_OK) continue;
        double f[3], J[18];
        mo_residual(T, res[i].xyz_actkey, res[i].obs, cam, f, J);
        mo_weighted_sq(f, prm->robust_kernel, prm->kernel_param);
        for (int r = 0; r < 6; ++r) {
          for (int c = 0; c < 6; ++c) A[6 * r + c] += J[r] * J[c] + J[6 + r] * J[6 + c] + J[12 + r] * J[12 + c];
          B[r] -= J[r] * f[0] + J[6 + r] * f[1] + J[12 + r] * f[2];
        }
      }
      solve_small(6, A, B, delta);
      double E[12];
      se3_exp(delta, E);
      pose_mul(E, T, Tn);
      double new_chi2 = 0, new_max_err = 0;
      for (int i = 0; i < n; ++i) {
        if (res[i].status != SVS_MATCH_OK) continue;
        double f[3];
        mo_residual(Tn, res[i].xyz_actkey, res[i].obs, cam, f, NULL);
        new_chi2 += mo_weighted_sq(f, prm->robust_kernel, prm->kernel_param);
        for (int k = 0; k < 3; ++k) if (fabs(f[k]) > new_max_err) new_max_err = fabs(f[k]);
      }
      if (isnan(new_chi2)) { st->status = 2; st->chi2 = chi2; st->max_err = max_err; return 2; }      /* throw runtime_error("Res is NaN!") */
      rho = chi2 - new_chi2;
      if (rho > 0) {
        memcpy(T, Tn, sizeof T);
        chi2 = new_chi2; max_err = new_max_err;
        double bm = -1;
        for (int c = 0; c < 6; ++c) if (fabs(B[c]) > bm) bm = fabs(B[c]);
        stop = bm <= EPS;
        const double q = 2 * rho - 1, sc = 1 - q * q * q;
        mu *= sc > 1. / 3. ? sc : 1. / 3.;
        nu = 2.; trial = 0;
      } else {
        mu *= nu; nu *= 2.; ++trial;
        if (trial == 5) stop = 1;
      }
    } while (!(rho > 0 || stop));
    if (stop) break;
  }
  memcpy(T_io, T, sizeof T);
  st->chi2 = chi2; st->max_err = max_err;
  return 0;
}

/* StereoFrontend::processMatchedPoints, stereo_frontend.cpp:834-974.  The list/hash-map/draw bookkeeping of the original
   is replaced by one flag record per matcher result; everything numeric follows the source line by line. */
void svs_ref_process_matched_points(const svs_match_result *res, const svs_candidate_point *pts, int n, int n_new_records,
                                    const svs_cam *cam, const double *T, float max_reproj_error, svs_gated_point *gated,
                                    svs_point_stats *stats) {
  memset(stats, 0, sizeof *stats);
  int half_width = (int)(cam->w * 0.5);                       /* :848-854 */
  int half_height = (int)(cam->h * 0.5);
  float third = 1. / 3.;
  int third_width = (int)(cam->w * third);
  int third_height = (int)(cam->h * third);
  int twothird_width = (int)(cam->w * 2 * third);
  int twothird_height = (int)(cam->h * 2 * third);
  double sum_track_length = 0.f;
  for (int k = 0; k < n; ++k) {
    memset(&gated[k], 0, sizeof gated[k]);
    if (res[k].status != 0) continue;                         /* not in obs_list */
    ++stats->num_obs;
    double diff[3];
    mo_residual(T, res[k].xyz_actkey, res[k].obs, cam, diff, 0);       /* uvu - se3xyz_stereo_.map(T, point), :863-866 */
    const double *uvu = res[k].obs;
    int level = pts[k].anchor_level;
    int factor = 1 << level;                                  /* zeroFromPyr_i(1, anchor_level) */
    if (fabs(diff[0]) < max_reproj_error * factor && fabs(diff[1]) < max_reproj_error * factor
        && fabs(diff[2]) < 3. * max_reproj_error) {           /* :869-871 */
      int i = 1, j = 1;
      if (uvu[0] < half_width) i = 0;
      if (uvu[1] < half_height) j = 0;
      ++stats->num_points_grid2x2[i * 2 + j];
      i = 2; j = 2;
      if (uvu[0] < third_width) i = 0; else if (uvu[0] < twothird_width) i = 1;
      if (uvu[1] < third_height) j = 0; else if (uvu[1] < twothird_height) j = 1;
      ++stats->num_points_grid3x3[i * 3 + j];
      ++stats->num_matched_points[level];
      const double *q = res[k].xyz_actkey;                    /* se3xyz.map(SE3(), point) = cam.map(project2d(point)) */
      double cu = q[0] / q[2] * cam->f + cam->cx, cv = q[1] / q[2] * cam->f + cam->cy;
      gated[k].accepted = 1;
      gated[k].is_new = k < n_new_records;                    /* id_obs.point_id < num_new_feat_matched, :920 */
      gated[k].curkey_uv_pyr[0] = cu / factor; gated[k].curkey_uv_pyr[1] = cv / factor;    /* pyrFromZero_2d */
      gated[k].uv_pyr[0] = uvu[0] / factor; gated[k].uv_pyr[1] = uvu[1] / factor;
      double dx = gated[k].uv_pyr[0] - gated[k].curkey_uv_pyr[0], dy = gated[k].uv_pyr[1] - gated[k].curkey_uv_pyr[1];
      sum_track_length += sqrt(dx * dx + dy * dy);            /* :925 / :955 */
      ++stats->num_track_points;
    }
  }
  stats->sum_track_length = sum_track_length;
}
