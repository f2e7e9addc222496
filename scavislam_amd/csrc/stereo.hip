// stereo.hip -- block-matching disparity for gfx950: StereoFrontend::calcDisparityCpu (stereo_frontend.cpp:620-653)
// = cv::StereoBM (OpenCV 2.4.2, external) with XSOBEL prefilter, 7x7 SAD over 32 disparities, texture and
// uniqueness tests, sub-pixel interpolation, left-right check (validateDisparity) and speckle filter.
// Semantics as restated in oracle/stereo.c; results are bit-identical to it (integer pipeline).
//
// Kernels (all HBM/LDS-bound u8/u16 integer work, no MFMA):
//   stereo_prefilter_kernel  3x3 x-Sobel -> saturating table, written +1 (so 0 can act as the MQSAD mask) into
//                            column-padded rows (replicated borders = the MIN/MAX clamps of the original)
//   stereo_bm_kernel         one wave = 64 output columns x a strip of rows.  Per image row a lane forms the 32
//                            horizontal 7-tap SADs with 16 V_QSAD/V_MQSAD_PK_U16_U8 (4 disparities each), keeps
//                            the last 7 rows in an LDS ring and slides the vertical sum with packed-u16 adds;
//                            argmin / uniqueness / parabola per pixel
//   stereo_bm_edge_kernel    the 3 leftmost output columns, whose right-image window clamps before the shift
//   stereo_validate_kernel   validateDisparity, one workgroup per row
//   stereo_ccl_*             speckle filter = connected components (union-find with atomicMin) + size test,
//                            fused with the 1/16 float conversion
#include "common.h"

namespace {

constexpr int PADL = 16, PADR = 48;         // prefiltered rows are padded: pitch = w + PADL + PADR
constexpr int NDISP = 32, WSZ2 = 3;
constexpr int BM_STRIP = 40;                // output rows per wave
constexpr int DISP_SHIFT = 4;
constexpr int FILTERED16 = -(1 << DISP_SHIFT);      // (minDisparity - 1) << 4 with minDisparity 0

struct StereoDev {
  int w, h, pitch;
  int cap, texthr, uniq, speckle_window, speckle_range, disp12;
  uint8_t *lp, *rp;      // [batch][h][pitch], values + 1
  int16_t *disp16;       // [batch][h][w]
  uint16_t *cost;        // [batch][h][w]
  int32_t *label, *count;// [batch][h*w]
};

__device__ __forceinline__ int xsobel_tab(int v, int cap) { return v < -cap ? 0 : v > cap ? 2 * cap : v + cap; }

// grid: (ceil(pitch/64), h, 2*batch)   z = 2*b + (0 left | 1 right)
__global__ __launch_bounds__(64) void stereo_prefilter_kernel(StereoDev S, const uint8_t *__restrict__ left, int lstride, size_t l_bstride,
                                                              const uint8_t *__restrict__ right, int rstride, size_t r_bstride) {
  const int pc = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y, b = blockIdx.z >> 1, side = blockIdx.z & 1;
  if (pc >= S.pitch) return;
  const uint8_t *src = side ? right + (size_t)b * r_bstride : left + (size_t)b * l_bstride;
  const int stride = side ? rstride : lstride;
  const int w = S.w, h = S.h, cap = S.cap;
  const int x = min(max(pc - PADL, 0), w - 1);
  int v = cap;
  if (!((h & 1) && y == h - 1) && x > 0 && x < w - 1) {
    const int yp = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yn = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
    const uint8_t *r0 = src + (size_t)yp * stride, *r1 = src + (size_t)y * stride, *r2 = src + (size_t)yn * stride;
    v = xsobel_tab((r0[x + 1] - r0[x - 1]) + 2 * (r1[x + 1] - r1[x - 1]) + (r2[x + 1] - r2[x - 1]), cap);
  }
  (side ? S.rp : S.lp)[((size_t)b * h + y) * S.pitch + pc] = (uint8_t)(v + 1);
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
union Pk { uint64_t q; us2 h[2]; uint32_t u[2]; };

// winner selection of one pixel from its 32 window SADs (findStereoCorrespondenceBM inner loop)
__device__ __forceinline__ void bm_select(const uint32_t (&sad)[NDISP], int tsum, const StereoDev &S, int16_t &disp, uint16_t &cost) {
  disp = (int16_t)FILTERED16; cost = 0;
  uint32_t best = 0xffffffffu;
#pragma unroll
  for (int d = 0; d < NDISP; ++d) best = min(best, (sad[d] << 5) | (uint32_t)d);      // first minimum wins
  const int minsad = (int)(best >> 5), mind = (int)(best & 31);
  if (tsum < S.texthr) return;
  int p = 0, n = 0;      // sad[mind + 1], sad[mind - 1] with the mirrored ends sad[-1] = sad[1], sad[32] = sad[30]
  const int ip = mind == NDISP - 1 ? NDISP - 2 : mind + 1, in = mind == 0 ? 1 : mind - 1;
#pragma unroll
  for (int d = 0; d < NDISP; ++d) { p = d == ip ? (int)sad[d] : p; n = d == in ? (int)sad[d] : n; }
  if (S.uniq > 0) {
    const int thresh = minsad + (minsad * S.uniq / 100);
    int cnt = 0;
#pragma unroll
    for (int d = 0; d < NDISP; ++d) cnt += (int)sad[d] <= thresh;
    // the scan of the original stops at a d outside [mind-1, mind+1] with sad[d] <= thresh
    cnt -= 1 + (mind > 0 && n <= thresh) + (mind < NDISP - 1 && p <= thresh);
    if (cnt > 0) return;
  }
  const int dd = p + n - 2 * minsad + abs(p - n);
  disp = (int16_t)(((NDISP - mind - 1) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4);
  cost = (uint16_t)minsad;
}

// grid: (ceil((width1-3)/64), ceil(h/BM_STRIP), batch), block 64.  Output columns x in [3, width1), X = x + 31.
__global__ __launch_bounds__(64) void stereo_bm_kernel(StereoDev S) {
  __shared__ uint64_t s_ring[7][9][64];      // [row slot][8 x packed h(d) + texture][lane]
  const int lane = threadIdx.x, b = blockIdx.z;
  const int w = S.w, h = S.h, width1 = w - NDISP + 1;
  const int x = min(3 + blockIdx.x * 64 + lane, width1 - 1);      // lanes past the end redo the last column (no store)
  const bool store = 3 + blockIdx.x * 64 + lane < width1;
  const int y0 = blockIdx.y * BM_STRIP, y1 = min(y0 + BM_STRIP, h);
  const uint8_t *lp = S.lp + (size_t)b * h * S.pitch + PADL + x + (NDISP - 1) - WSZ2;
  const uint8_t *rp = S.rp + (size_t)b * h * S.pitch + PADL + x - WSZ2;
  const uint32_t ft4 = 0x01010101u * (uint32_t)(S.cap + 1);
  Pk sad[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) sad[g].q = 0;
  int tsum = 0, slot = 0;
  for (int r = y0 - WSZ2; r < y1 + WSZ2; ++r) {
    const int yy = min(max(r, 0), h - 1);
    const uint8_t *lrow = lp + (size_t)yy * S.pitch, *rrow = rp + (size_t)yy * S.pitch;
    const uint32_t l0 = load_u32_unaligned(lrow), l1 = load_u32_unaligned(lrow + 4) & 0x00ffffffu;      // 7 window bytes, 8th = mask
    uint32_t rw[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) rw[k] = load_u32_unaligned(rrow + 4 * k);
    const bool full = r - (y0 - WSZ2) >= 7;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      uint64_t hq = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)rw[g] | ((uint64_t)rw[g + 1] << 32), l0, 0ull);
      hq = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)rw[g + 1] | ((uint64_t)rw[g + 2] << 32), l1, hq);
      Pk hn, ho;
      hn.q = hq;
      ho.q = full ? s_ring[slot][g][lane] : 0ull;
      s_ring[slot][g][lane] = hq;
      sad[g].h[0] = sad[g].h[0] + hn.h[0] - ho.h[0];
      sad[g].h[1] = sad[g].h[1] + hn.h[1] - ho.h[1];
    }
    {
      const int t = (int)__builtin_amdgcn_sad_u8(l1 | (ft4 & 0xff000000u), ft4, __builtin_amdgcn_sad_u8(l0, ft4, 0u));
      const int to = full ? (int)s_ring[slot][8][lane] : 0;
      s_ring[slot][8][lane] = (uint64_t)t;
      tsum += t - to;
    }
    const int y = r - WSZ2;
    if (y >= y0 && store) {
      uint32_t s32[NDISP];
#pragma unroll
      for (int g = 0; g < 8; ++g) { s32[4 * g] = sad[g].u[0] & 0xffff; s32[4 * g + 1] = sad[g].u[0] >> 16; s32[4 * g + 2] = sad[g].u[1] & 0xffff; s32[4 * g + 3] = sad[g].u[1] >> 16; }
      int16_t d16; uint16_t c16;
      bm_select(s32, tsum, S, d16, c16);
      const size_t o = ((size_t)b * h + y) * w + x + (NDISP - 1);
      S.disp16[o] = d16; S.cost[o] = c16;
    }
    slot = slot == 6 ? 0 : slot + 1;
  }
}

// columns X < 31 (never searched) and x in [0,3): generic clamped evaluation.  grid: (h, batch), block 64
__global__ __launch_bounds__(64) void stereo_bm_edge_kernel(StereoDev S) {
  const int y = blockIdx.x, b = blockIdx.y, w = S.w, h = S.h, lane = threadIdx.x;
  const int16_t FILTERED = (int16_t)FILTERED16;
  for (int c = lane; c < NDISP - 1 && c < w; c += 64) { S.disp16[((size_t)b * h + y) * w + c] = FILTERED; S.cost[((size_t)b * h + y) * w + c] = 0; }
  const int width1 = w - NDISP + 1;
  if (lane >= 3 || lane >= width1) return;
  const int x = lane;
  const uint8_t *lp = S.lp + (size_t)b * h * S.pitch + PADL, *rp = S.rp + (size_t)b * h * S.pitch + PADL;
  uint32_t sad[NDISP];
#pragma unroll
  for (int d = 0; d < NDISP; ++d) sad[d] = 0;
  int tsum = 0;
  for (int dy = -WSZ2; dy <= WSZ2; ++dy) {
    const int yy = min(max(y + dy, 0), h - 1);
    for (int dx = -WSZ2; dx <= WSZ2; ++dx) {
      const int lval = lp[(size_t)yy * S.pitch + x + dx + NDISP - 1];      // x + dx + 31 >= 28: no clamp on the left image here
      const int rc = max(x + dx, 0);
#pragma unroll
      for (int d = 0; d < NDISP; ++d) sad[d] += (uint32_t)abs(lval - (int)rp[(size_t)yy * S.pitch + min(rc + d, w - 1)]);
      tsum += abs(lval - (S.cap + 1));
    }
  }
  int16_t d16; uint16_t c16;
  bm_select(sad, tsum, S, d16, c16);
  const size_t o = ((size_t)b * h + y) * w + x + (NDISP - 1);
  S.disp16[o] = d16; S.cost[o] = c16;
}

// validateDisparity (with D2): one workgroup per image row.  grid: (h, batch), block 256
__global__ __launch_bounds__(256) void stereo_validate_kernel(StereoDev S) {
  extern __shared__ int s_mem[];
  const int w = S.w, y = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  int *s_d = s_mem, *s_c = s_mem + w, *s_d2 = s_mem + 2 * w;
  const int SCALE = 1 << DISP_SHIFT, INVALID = -SCALE, maxdiff = S.disp12 * SCALE;
  int16_t *dp = S.disp16 + ((size_t)b * S.h + y) * w;
  const uint16_t *cp = S.cost + ((size_t)b * S.h + y) * w;
  for (int x = tid; x < w; x += 256) { s_d[x] = dp[x]; s_c[x] = cp[x]; }
  __syncthreads();
  const int minX1 = NDISP;
  for (int x2 = tid; x2 < w; x2 += 256) {
    int bc = 0x7fffffff, bd = INVALID;
    for (int k = 0; k <= NDISP; ++k) {      // sources x = x2 + round(d): ascending x, strictly smaller cost wins
      const int x = x2 + k;
      if (x < minX1 || x >= w) continue;
      const int d = s_d[x];
      if (d == INVALID || ((d + SCALE / 2) >> DISP_SHIFT) != k) continue;
      if (s_c[x] < bc) { bc = s_c[x]; bd = d; }
    }
    s_d2[x2] = bd;
  }
  __syncthreads();
  for (int x = minX1 + tid; x < w; x += 256) {
    const int d = s_d[x];
    if (d == INVALID) continue;
    const int x0 = x - (d >> DISP_SHIFT), x1 = x - ((d + SCALE - 1) >> DISP_SHIFT);
    const bool bad0 = x0 >= 0 && x0 < w && s_d2[x0] > INVALID && abs(s_d2[x0] - d) > maxdiff;
    const bool bad1 = x1 >= 0 && x1 < w && s_d2[x1] > INVALID && abs(s_d2[x1] - d) > maxdiff;
    if (bad0 && bad1) dp[x] = (int16_t)INVALID;
  }
}

// ---- speckle filter: connected components by union-find (labels = linear pixel index inside the frame) ----------
__device__ __forceinline__ int ccl_find(const int32_t *label, int x) {
  int p = label[x];
  while (p != x) { x = p; p = label[x]; }
  return x;
}
__device__ __forceinline__ void ccl_union(int32_t *label, int a, int b) {
  while (true) {
    a = ccl_find(label, a); b = ccl_find(label, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }      // a > b: hang the larger root under the smaller
    const int old = atomicMin(&label[a], b);
    if (old == a) return;
    a = old;
  }
}
// grid: (ceil(w*h/256), batch)
__global__ __launch_bounds__(256) void stereo_ccl_init_kernel(StereoDev S) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = S.w * S.h;
  if (i >= n) return;
  const size_t o = (size_t)blockIdx.y * n + i;
  S.label[o] = S.disp16[o] == (int16_t)FILTERED16 ? -1 : i;
  S.count[o] = 0;
}
__global__ __launch_bounds__(256) void stereo_ccl_merge_kernel(StereoDev S) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = S.w * S.h, w = S.w;
  if (i >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  const int16_t *d = S.disp16 + base;
  int32_t *label = S.label + base;
  const int FILTERED = FILTERED16, dv = d[i];
  if (dv == FILTERED) return;
  const int x = i % w;
  if (x + 1 < w) { const int dn = d[i + 1]; if (dn != FILTERED && abs(dv - dn) <= S.speckle_range) ccl_union(label, i, i + 1); }
  if (i + w < n) { const int dn = d[i + w]; if (dn != FILTERED && abs(dv - dn) <= S.speckle_range) ccl_union(label, i, i + w); }
}
__global__ __launch_bounds__(256) void stereo_ccl_count_kernel(StereoDev S) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = S.w * S.h;
  if (i >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  if (S.label[base + i] < 0) return;
  const int root = ccl_find(S.label + base, i);
  S.label[base + i] = root;      // flatten (roots stay roots, so concurrent finds remain valid)
  atomicAdd(&S.count[base + root], 1);
}
// disparity in pixels; components of <= speckle_window pixels are filtered.  use_ccl == 0: plain conversion
__global__ __launch_bounds__(256) void stereo_finish_kernel(StereoDev S, int use_ccl, float *__restrict__ out, int dstride, size_t d_bstride) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = S.w * S.h;
  if (i >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  int d = S.disp16[base + i];
  if (use_ccl && d != FILTERED16 && S.count[base + S.label[base + i]] <= S.speckle_window) d = FILTERED16;
  out[(size_t)blockIdx.y * d_bstride + (size_t)(i / S.w) * dstride + (i % S.w)] = (float)d * (1.f / (1 << DISP_SHIFT));
}

}  // namespace

struct svs_stereo {
  svs_ctx *ctx = nullptr;
  int w = 0, h = 0, max_batch = 0, pitch = 0;
  svs_stereo_params prm{};
  uint8_t *d_lp = nullptr, *d_rp = nullptr;
  int16_t *d_disp16 = nullptr;
  uint16_t *d_cost = nullptr;
  int32_t *d_label = nullptr, *d_count = nullptr;
};

extern "C" int svs_stereo_create(svs_ctx *ctx, int w, int h, int max_batch, const svs_stereo_params *prm, svs_stereo **out) {
  SVS_REQUIRE(ctx, ctx && prm && out && w > 0 && h > 0 && max_batch > 0);
  if (prm->sad_window != 7 || prm->min_disparity != 0 || prm->num_disparities != NDISP || prm->prefilter_cap < 1 || prm->prefilter_cap > 63 ||
      w < NDISP + 2 * WSZ2 || h < 2) {
    ctx->err = "svs_stereo: only SADWindowSize 7, minDisparity 0, numberOfDisparities 32, preFilterCap 1..63, w >= 38 are supported";
    return SVS_ERR_UNSUPPORTED;
  }
  svs_stereo *s = new svs_stereo();
  s->ctx = ctx; s->w = w; s->h = h; s->max_batch = max_batch; s->pitch = w + PADL + PADR; s->prm = *prm;
  const size_t n = (size_t)w * h * max_batch, np = (size_t)s->pitch * h * max_batch + 64;
  SVS_HIP(ctx, hipMalloc(&s->d_lp, np));
  SVS_HIP(ctx, hipMalloc(&s->d_rp, np));
  SVS_HIP(ctx, hipMalloc(&s->d_disp16, n * sizeof(int16_t)));
  SVS_HIP(ctx, hipMalloc(&s->d_cost, n * sizeof(uint16_t)));
  SVS_HIP(ctx, hipMalloc(&s->d_label, n * sizeof(int32_t)));
  SVS_HIP(ctx, hipMalloc(&s->d_count, n * sizeof(int32_t)));
  *out = s;
  return SVS_OK;
}

extern "C" int svs_stereo_destroy(svs_stereo *s) {
  if (!s) return SVS_OK;
  (void)hipStreamSynchronize(s->ctx->stream);
  if (s->d_lp) (void)hipFree(s->d_lp); if (s->d_rp) (void)hipFree(s->d_rp);
  if (s->d_disp16) (void)hipFree(s->d_disp16); if (s->d_cost) (void)hipFree(s->d_cost);
  if (s->d_label) (void)hipFree(s->d_label); if (s->d_count) (void)hipFree(s->d_count);
  delete s;
  return SVS_OK;
}

extern "C" int svs_stereo_compute(svs_stereo *s, const uint8_t *d_left, int lstride, size_t l_bstride, const uint8_t *d_right, int rstride,
                                  size_t r_bstride, float *d_disp, int dstride, size_t d_bstride, int n_batch) {
  svs_ctx *ctx = s ? s->ctx : nullptr;
  SVS_REQUIRE(ctx, s && d_left && d_right && d_disp && n_batch >= 1 && n_batch <= s->max_batch && lstride >= s->w && rstride >= s->w && dstride >= s->w);
  StereoDev S{};
  S.w = s->w; S.h = s->h; S.pitch = s->pitch;
  S.cap = s->prm.prefilter_cap; S.texthr = s->prm.texture_threshold; S.uniq = s->prm.uniqueness_ratio;
  S.speckle_window = s->prm.speckle_window; S.speckle_range = s->prm.speckle_range; S.disp12 = s->prm.disp12_max_diff;
  S.lp = s->d_lp; S.rp = s->d_rp; S.disp16 = s->d_disp16; S.cost = s->d_cost; S.label = s->d_label; S.count = s->d_count;
  const int w = s->w, h = s->h, width1 = w - NDISP + 1, n = w * h;
  hipLaunchKernelGGL(stereo_prefilter_kernel, dim3(div_up(s->pitch, 64), h, 2 * n_batch), dim3(64), 0, ctx->stream, S, d_left, lstride, l_bstride,
                     d_right, rstride, r_bstride);
  SVS_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(stereo_bm_edge_kernel, dim3(h, n_batch), dim3(64), 0, ctx->stream, S);
  SVS_LAUNCH_CHECK(ctx);
  if (width1 > 3) {
    hipLaunchKernelGGL(stereo_bm_kernel, dim3(div_up(width1 - 3, 64), div_up(h, BM_STRIP), n_batch), dim3(64), 0, ctx->stream, S);
    SVS_LAUNCH_CHECK(ctx);
  }
  if (s->prm.disp12_max_diff >= 0) {
    hipLaunchKernelGGL(stereo_validate_kernel, dim3(h, n_batch), dim3(256), sizeof(int) * 3 * (size_t)w, ctx->stream, S);
    SVS_LAUNCH_CHECK(ctx);
  }
  const bool ccl = s->prm.speckle_range >= 0 && s->prm.speckle_window > 0;
  const dim3 gp(div_up(n, 256), n_batch);
  if (ccl) {
    hipLaunchKernelGGL(stereo_ccl_init_kernel, gp, dim3(256), 0, ctx->stream, S); SVS_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(stereo_ccl_merge_kernel, gp, dim3(256), 0, ctx->stream, S); SVS_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(stereo_ccl_count_kernel, gp, dim3(256), 0, ctx->stream, S); SVS_LAUNCH_CHECK(ctx);
  }
  hipLaunchKernelGGL(stereo_finish_kernel, gp, dim3(256), 0, ctx->stream, S, ccl ? 1 : 0, d_disp, dstride, d_bstride);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
