// fast_view.h -- read-only device view of svs_fast's corner bitmaps, shared by fast.hip and match.hip.
#pragma once
#include "common.h"
struct FastView {
  // corner bitmap of a level (fast.hip, LevelDev): pixel (x, y) of cell column ci = bit x + bm_gap * ci of row y (rows of bm_stride bytes, slots of bm_bstride bytes)
  const uint8_t *bm[SVS_NUM_PYR_LEVELS]; int bm_stride[SVS_NUM_PYR_LEVELS]; size_t bm_bstride[SVS_NUM_PYR_LEVELS]; int bm_gap[SVS_NUM_PYR_LEVELS];
  const int *emit; int ncell_total; int cell_base[SVS_NUM_PYR_LEVELS];
  int gx[SVS_NUM_PYR_LEVELS], gy[SVS_NUM_PYR_LEVELS], cell_w[SVS_NUM_PYR_LEVELS], cell_h[SVS_NUM_PYR_LEVELS];
  int w[SVS_NUM_PYR_LEVELS], h[SVS_NUM_PYR_LEVELS]; int n_levels;
};
FastView svs_fast_view_internal(const svs_fast *f);
