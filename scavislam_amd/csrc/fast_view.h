// fast_view.h -- read-only device view of svs_fast's score maps, shared by fast.hip and match.hip.
#pragma once
#include "common.h"
struct FastView {
  const uint8_t *score[SVS_NUM_PYR_LEVELS]; int score_stride[SVS_NUM_PYR_LEVELS]; size_t score_bstride[SVS_NUM_PYR_LEVELS];
  const int *emit; int ncell_total; int cell_base[SVS_NUM_PYR_LEVELS];
  int gx[SVS_NUM_PYR_LEVELS], gy[SVS_NUM_PYR_LEVELS], cell_w[SVS_NUM_PYR_LEVELS], cell_h[SVS_NUM_PYR_LEVELS];
  int w[SVS_NUM_PYR_LEVELS], h[SVS_NUM_PYR_LEVELS]; int n_levels;
};
FastView svs_fast_view_internal(const svs_fast *f);
