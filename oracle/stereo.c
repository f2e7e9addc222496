/*
 * stereo.c -- CPU restatement of the reference's block-matching disparity stage.
 * TEST INFRASTRUCTURE ONLY (see svs_oracle.h).  PARITY UNPINNED (cv::StereoBM is OpenCV 2.4.2, not under /root/reference).
 *
 * Reference call site: StereoFrontend::calcDisparityCpu, stereo_frontend.cpp:620-653:
 *   cv::StereoBM (default-constructed = BASIC preset: preFilterType XSOBEL, preFilterSize 9)
 *   preFilterCap 31, SADWindowSize 7, minDisparity 0, numberOfDisparities 32, textureThreshold 10,
 *   uniquenessRatio 15, speckleWindowSize 100, speckleRange 32, disp12MaxDiff 1; output CV_32F.
 * [3rd-party: OpenCV 2.4.2 modules/calib3d/src/stereobm.cpp (prefilterXSobel,
 *  findStereoCorrespondenceBM, FindStereoCorrespInvoker) and stereosgbm.cpp (validateDisparity,
 *  filterSpeckles); source not under /root/reference -- this restates the published algorithm.]
 *
 * Two places where OpenCV 2.4.2's behaviour is undefined are given a definition here:
 *   D1  the right-image read `rptr[d]` runs up to 3 bytes past the end of the prefiltered row for
 *       the last 3 window columns; here the column is clamped to width-1.
 *   D2  validateDisparity's first pass reads the cost of FILTERED pixels, which findStereoCorrespondenceBM
 *       never writes; here FILTERED pixels do not vote (as later OpenCV releases do).
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "svs_oracle.h"

#define DISP_SHIFT 4

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* prefilterXSobel: tab[v] saturates the 3x3 x-Sobel to [0, 2*cap]; rows are reflected (101) at the top and
 * bottom; columns 0 and w-1 and -- for odd heights -- the last row are the neutral value `cap`. */
void svs_ref_stereo_prefilter_xsobel(const uint8_t *src, int w, int h, int stride, int cap, uint8_t *dst) {
  for (int y = 0; y < h; ++y) {
    uint8_t *d = dst + (size_t)y * w;
    if ((h & 1) && y == h - 1) { memset(d, cap, (size_t)w); continue; }
    const int yp = y > 0 ? y - 1 : (h > 1 ? 1 : 0);
    const int yn = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
    const uint8_t *r0 = src + (size_t)yp * stride, *r1 = src + (size_t)y * stride, *r2 = src + (size_t)yn * stride;
    d[0] = d[w - 1] = (uint8_t)cap;
    for (int x = 1; x < w - 1; ++x) {
      const int v = (r0[x + 1] - r0[x - 1]) + 2 * (r1[x + 1] - r1[x - 1]) + (r2[x + 1] - r2[x - 1]);
      d[x] = (uint8_t)(v < -cap ? 0 : v > cap ? 2 * cap : v + cap);
    }
  }
}

/* findStereoCorrespondenceBM on prefiltered images: disp16 (4 fractional bits, FILTERED = (mindisp-1)*16) and
 * cost = SAD of the winner.  Window rows/columns are replicated at the image border (the clamped hsad/htext
 * rows and the MIN/MAX column clamps of the original). */
void svs_ref_stereo_bm_core(const uint8_t *lp, const uint8_t *rp, int w, int h, const svs_stereo_params *p, int16_t *disp16,
                            int32_t *cost) {
  const int wsz2 = p->sad_window / 2, ndisp = p->num_disparities, mindisp = p->min_disparity;
  const int lofs = ndisp - 1 + mindisp > 0 ? ndisp - 1 + mindisp : 0;
  const int rofs = ndisp - 1 + mindisp < 0 ? -(ndisp - 1 + mindisp) : 0;
  const int width1 = w - rofs - ndisp + 1, ftzero = p->prefilter_cap;
  const int16_t FILTERED = (int16_t)((mindisp - 1) << DISP_SHIFT);
  for (size_t i = 0; i < (size_t)w * h; ++i) { disp16[i] = FILTERED; cost[i] = 0; }
  if (width1 <= 0) return;
  int *sad = (int *)malloc(sizeof(int) * (size_t)(ndisp + 2));
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < width1; ++x) {
      int *s = sad + 1, tsum = 0;
      for (int d = 0; d < ndisp; ++d) s[d] = 0;
      for (int dy = -wsz2; dy <= wsz2; ++dy) {
        const int yy = clampi(y + dy, 0, h - 1);
        for (int dx = -wsz2; dx <= wsz2; ++dx) {
          const int lval = lp[(size_t)yy * w + clampi(x + dx, -lofs, w - lofs - 1) + lofs];
          const int rc = clampi(x + dx, -rofs, w - rofs - 1) + rofs;
          for (int d = 0; d < ndisp; ++d) s[d] += abs(lval - rp[(size_t)yy * w + (rc + d < w ? rc + d : w - 1)]);      /* D1 */
          tsum += abs(lval - ftzero);
        }
      }
      int minsad = INT_MAX, mind = -1;
      for (int d = 0; d < ndisp; ++d)
        if (s[d] < minsad) { minsad = s[d]; mind = d; }
      int16_t *out = disp16 + (size_t)y * w + x + lofs;
      if (tsum < p->texture_threshold) continue;
      if (p->uniqueness_ratio > 0) {
        const int thresh = minsad + (minsad * p->uniqueness_ratio / 100);
        int d;
        for (d = 0; d < ndisp; ++d)
          if (s[d] <= thresh && (d < mind - 1 || d > mind + 1)) break;
        if (d < ndisp) continue;
      }
      s[-1] = s[1]; s[ndisp] = s[ndisp - 2];
      const int pp = s[mind + 1], nn = s[mind - 1], dd = pp + nn - 2 * s[mind] + abs(pp - nn);
      *out = (int16_t)(((ndisp - mind - 1 + mindisp) * 256 + (dd != 0 ? (pp - nn) * 256 / dd : 0) + 15) >> 4);
      cost[(size_t)y * w + x + lofs] = s[mind];
    }
  free(sad);
}

/* validateDisparity (left-right consistency from the left cost volume winners), row by row */
void svs_ref_stereo_validate(int16_t *disp16, const int32_t *cost, int w, int h, const svs_stereo_params *p) {
  const int minD = p->min_disparity, maxD = minD + p->num_disparities;
  const int minX1 = maxD > 0 ? maxD : 0, maxX1 = w + (minD < 0 ? minD : 0);
  const int SCALE = 1 << DISP_SHIFT, INVALID = (minD - 1) * SCALE, maxdiff = p->disp12_max_diff * SCALE;
  int *d2 = (int *)malloc(sizeof(int) * 2 * (size_t)w), *c2 = d2 + w;
  for (int y = 0; y < h; ++y) {
    int16_t *dp = disp16 + (size_t)y * w;
    const int32_t *cp = cost + (size_t)y * w;
    for (int x = 0; x < w; ++x) { d2[x] = INVALID; c2[x] = INT_MAX; }
    for (int x = minX1; x < maxX1; ++x) {
      const int d = dp[x], c = cp[x];
      if (d == INVALID) continue;                                                       /* D2 */
      const int x2 = x - ((d + SCALE / 2) >> DISP_SHIFT);
      if (x2 < 0 || x2 >= w) continue;
      if (c2[x2] > c) { c2[x2] = c; d2[x2] = d; }
    }
    for (int x = minX1; x < maxX1; ++x) {
      const int d = dp[x];
      if (d == INVALID) continue;
      const int d0 = d >> DISP_SHIFT, d1 = (d + SCALE - 1) >> DISP_SHIFT;
      const int x0 = x - d0, x1 = x - d1;
      if ((0 <= x0 && x0 < w && d2[x0] > INVALID && abs(d2[x0] - d) > maxdiff) &&
          (0 <= x1 && x1 < w && d2[x1] > INVALID && abs(d2[x1] - d) > maxdiff))
        dp[x] = (int16_t)INVALID;
    }
  }
  free(d2);
}

/* filterSpeckles: 4-connected components of non-FILTERED pixels whose neighbouring values differ by <= max_diff;
 * components of <= max_size pixels become FILTERED */
void svs_ref_stereo_filter_speckles(int16_t *disp16, int w, int h, int new_val, int max_size, int max_diff) {
  const size_t n = (size_t)w * h;
  int32_t *label = (int32_t *)calloc(n, sizeof(int32_t));
  int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * n);
  uint8_t *small = (uint8_t *)calloc(n + 1, 1);
  int cur = 0;
  for (size_t i = 0; i < n; ++i) {
    if (disp16[i] == new_val) continue;
    if (label[i]) { if (small[label[i]]) disp16[i] = (int16_t)new_val; continue; }
    int sp = 0, count = 0;
    ++cur;
    label[i] = cur; stack[sp++] = (int32_t)i;
    while (sp) {
      const int32_t q = stack[--sp];
      const int qx = q % w, qy = q / w, dv = disp16[q];
      ++count;
      const int nb[4] = {qx < w - 1 ? q + 1 : -1, qx > 0 ? q - 1 : -1, qy < h - 1 ? q + w : -1, qy > 0 ? q - w : -1};
      for (int k = 0; k < 4; ++k) {
        const int t = nb[k];
        if (t >= 0 && !label[t] && disp16[t] != new_val && abs(dv - disp16[t]) <= max_diff) { label[t] = cur; stack[sp++] = t; }
      }
    }
    if (count <= max_size) { small[cur] = 1; disp16[i] = (int16_t)new_val; }
  }
  /* pixels of small components visited before their label was known to be small */
  for (size_t i = 0; i < n; ++i)
    if (label[i] && small[label[i]]) disp16[i] = (int16_t)new_val;
  free(label); free(stack); free(small);
}

/* cv::StereoBM::operator()(left, right, disp, CV_32F): float disparity, -1 (= (mindisp-1)) where filtered */
void svs_ref_stereo_bm(const uint8_t *left, const uint8_t *right, int w, int h, int stride, const svs_stereo_params *p, float *disp,
                       int dstride) {
  const size_t n = (size_t)w * h;
  uint8_t *lp = (uint8_t *)malloc(n), *rp = (uint8_t *)malloc(n);
  int16_t *d16 = (int16_t *)malloc(sizeof(int16_t) * n);
  int32_t *cost = (int32_t *)malloc(sizeof(int32_t) * n);
  svs_ref_stereo_prefilter_xsobel(left, w, h, stride, p->prefilter_cap, lp);
  svs_ref_stereo_prefilter_xsobel(right, w, h, stride, p->prefilter_cap, rp);
  svs_ref_stereo_bm_core(lp, rp, w, h, p, d16, cost);
  if (p->disp12_max_diff >= 0) svs_ref_stereo_validate(d16, cost, w, h, p);
  if (p->speckle_range >= 0 && p->speckle_window > 0)
    svs_ref_stereo_filter_speckles(d16, w, h, (p->min_disparity - 1) << DISP_SHIFT, p->speckle_window, p->speckle_range);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) disp[(size_t)y * dstride + x] = (float)d16[(size_t)y * w + x] * (1.f / (1 << DISP_SHIFT));
  free(lp); free(rp); free(d16); free(cost);
}
