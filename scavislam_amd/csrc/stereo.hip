// stereo.hip -- block-matching disparity for gfx950: StereoFrontend::calcDisparityCpu (stereo_frontend.cpp:620-653)
// = cv::StereoBM (OpenCV 2.4.2, external) with XSOBEL prefilter, 7x7 SAD over 32 disparities, texture and
// uniqueness tests, sub-pixel interpolation, left-right check (validateDisparity) and speckle filter.
// Semantics as restated in oracle/stereo.c; results are bit-identical to it (integer pipeline).
//
// Kernels (all HBM/LDS-bound u8/u16 integer work, no MFMA):
//   stereo_prefilter_kernel  3x3 x-Sobel -> saturating table (4 pixels per thread), written +1 (so 0 can act as the MQSAD mask) into
//                            column-padded rows (replicated borders = the MIN/MAX clamps of the original)
//   stereo_bm_kernel         one wave = 64 output columns x a strip of rows.  Per image row a lane forms the 32
//                            horizontal 7-tap SADs with 16 V_QSAD/V_MQSAD_PK_U16_U8 (4 disparities each), keeps
//                            the last 7 rows in an LDS ring and slides the vertical sum with packed-u16 adds;
//                            argmin / uniqueness / parabola per pixel
//   stereo_bm_edge_kernel    the 3 leftmost output columns, whose right-image window clamps before the shift
//   stereo_validate_kernel   validateDisparity, one workgroup per row (one LDS atomicMin of (cost, x) per source pixel)
//   stereo_ccl_*             speckle filter = connected components (horizontal runs, then union-find with atomicMin
//                            across rows) + saturating size count; the size test is fused with the 1/16 float conversion
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

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ int byte_of(uint32_t lo, uint32_t hi, int i) { return (int)((i < 4 ? lo >> (8 * i) : hi >> (8 * (i - 4))) & 0xffu); }

// dword at byte position q of a row of w bytes, read from the clamped position and shifted into place: bytes that fall
// outside the row come back as 0 (they only ever feed border columns, which are overwritten).  Branch-free on purpose:
// a per-byte border path makes every wave that holds an edge lane walk it, with a wait behind each load.
__device__ __forceinline__ uint32_t load_u32_clamped(const uint8_t *row, int q, int w) {
  const int qs = min(max(q, 0), w - 4), d = q - qs;
  const uint32_t raw = load_u32_unaligned(row + qs);
  return d == 0 ? raw : (d >= 4 || d <= -4) ? 0u : d > 0 ? raw >> (8 * d) : raw << (-8 * d);
}
// 4 padded columns per thread, 4 rows per workgroup.  grid: (ceil(pitch/256), ceil(h/4), 2*batch), block (64, 4);
// z = 2*b + (0 left | 1 right)
__global__ __launch_bounds__(256) void stereo_prefilter_kernel(StereoDev S, const uint8_t *__restrict__ left, int lstride, size_t l_bstride,
                                                               const uint8_t *__restrict__ right, int rstride, size_t r_bstride) {
  const int pc0 = 4 * (blockIdx.x * 64 + threadIdx.x), y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z >> 1, side = blockIdx.z & 1;
  const int w = S.w, h = S.h, cap = S.cap;
  if (pc0 >= S.pitch || y >= h) return;      // pitch is a multiple of 4
  const uint8_t *src = side ? right + (size_t)b * r_bstride : left + (size_t)b * l_bstride;
  const int stride = side ? rstride : lstride;
  const int x0 = pc0 - PADL;
  const uint32_t fill = 0x01010101u * (uint32_t)(cap + 1);      // pads, border columns, odd last row
  uint32_t out = fill;
  if (!((h & 1) && y == h - 1)) {
    const int yp = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yn = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
    const uint8_t *r0 = src + (size_t)yp * stride, *r1 = src + (size_t)y * stride, *r2 = src + (size_t)yn * stride;
    // bytes x0-1 .. x0+4 of the three rows (w >= 38, checked at create)
    const uint32_t a0 = load_u32_clamped(r0, x0 - 1, w), a1 = load_u32_clamped(r0, x0 + 3, w);
    const uint32_t b0 = load_u32_clamped(r1, x0 - 1, w), b1 = load_u32_clamped(r1, x0 + 3, w);
    const uint32_t c0 = load_u32_clamped(r2, x0 - 1, w), c1 = load_u32_clamped(r2, x0 + 3, w);
    out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = x0 + j;
      const int v = (byte_of(a0, a1, j + 2) - byte_of(a0, a1, j)) + 2 * (byte_of(b0, b1, j + 2) - byte_of(b0, b1, j)) +
                    (byte_of(c0, c1, j + 2) - byte_of(c0, c1, j));
      const uint32_t o = (x >= 1 && x <= w - 2) ? (uint32_t)(xsobel_tab(v, cap) + 1) : (uint32_t)(cap + 1);
      out |= o << (8 * j);
    }
  }
  *reinterpret_cast<uint32_t *>((side ? S.rp : S.lp) + ((size_t)b * h + y) * S.pitch + pc0) = out;
}


typedef unsigned short us2 __attribute__((ext_vector_type(2)));
union Pk { uint64_t q; us2 h[2]; uint32_t u[2]; };

// the window bytes one lane needs from one image row: 7 left bytes (8th masked) and 44 right bytes
struct RowWin { uint32_t l0, l1, r[11]; };
__device__ __forceinline__ RowWin load_rowwin(const uint8_t *lrow, const uint8_t *rrow) {
  RowWin wv;
  wv.l0 = load_u32_unaligned(lrow);
  wv.l1 = load_u32_unaligned(lrow + 4);
#pragma unroll
  for (int k = 0; k < 11; ++k) wv.r[k] = load_u32_unaligned(rrow + 4 * k);
  return wv;
}
// horizontal 7-tap SADs of one row for the 32 disparity indices (4 per V_QSAD/V_MQSAD pair) + texture term
__device__ __forceinline__ void row_sads(const RowWin &wv, uint32_t ft4, Pk (&hh)[8], int &t) {
  const uint32_t l1 = wv.l1 & 0x00ffffffu;      // byte 7 = 0 = masked out by MQSAD
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const uint64_t hq = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)wv.r[g] | ((uint64_t)wv.r[g + 1] << 32), wv.l0, 0ull);
    hh[g].q = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)wv.r[g + 1] | ((uint64_t)wv.r[g + 2] << 32), l1, hq);
  }
  t = (int)__builtin_amdgcn_sad_u8(l1 | (ft4 & 0xff000000u), ft4, __builtin_amdgcn_sad_u8(wv.l0, ft4, 0u));
}

// per-lane LDS scratch of bm_select; a union, so that the halfword and dword views may alias
union SelScr { uint32_t u[17]; unsigned short s[34]; };
// winner selection of one pixel from its 32 window SADs (findStereoCorrespondenceBM inner loop).  Dynamic indexing
// (sad[mind +- 1], masking the winner's neighbourhood) goes through the per-lane LDS scratch.
__device__ __forceinline__ void bm_select(const Pk (&sad)[8], int tsum, const StereoDev &S, SelScr &scr, int16_t &disp, uint16_t &cost) {
  disp = (int16_t)FILTERED16; cost = 0;
  uint32_t best = 0xffffffffu;
#pragma unroll
  for (int g = 0; g < 8; ++g) {      // first minimum wins: key = sad * 32 + d
    best = min(best, ((sad[g].u[0] & 0xffffu) << 5) | (uint32_t)(4 * g));
    best = min(best, ((sad[g].u[0] >> 16) << 5) | (uint32_t)(4 * g + 1));
    best = min(best, ((sad[g].u[1] & 0xffffu) << 5) | (uint32_t)(4 * g + 2));
    best = min(best, ((sad[g].u[1] >> 16) << 5) | (uint32_t)(4 * g + 3));
  }
  const int minsad = (int)(best >> 5), mind = (int)(best & 31);
  if (tsum < S.texthr) return;
#pragma unroll
  for (int g = 0; g < 8; ++g) { scr.u[2 * g] = sad[g].u[0]; scr.u[2 * g + 1] = sad[g].u[1]; }
  unsigned short *s16 = scr.s;
  // sad[mind + 1], sad[mind - 1] with the mirrored ends sad[-1] = sad[1], sad[32] = sad[30]
  const int p = s16[mind == NDISP - 1 ? NDISP - 2 : mind + 1], n = s16[mind == 0 ? 1 : mind - 1];
  if (S.uniq > 0) {
    // the scan of the original stops at a d outside [mind-1, mind+1] with sad[d] <= thresh: mask those three, take the min
    const int thresh = minsad + (minsad * S.uniq / 100);
    s16[mind] = 0xffff;
    s16[mind == 0 ? 32 : mind - 1] = 0xffff;            // slots 32/33 = spare halfword pair
    s16[mind == NDISP - 1 ? 33 : mind + 1] = 0xffff;
    us2 m = {0xffff, 0xffff};
#pragma unroll
    for (int k = 0; k < 16; ++k) { Pk t; t.u[0] = scr.u[k]; m = __builtin_elementwise_min(m, t.h[0]); }
    if ((int)min(m.x, m.y) <= thresh) return;
  }
  const int dd = p + n - 2 * minsad + abs(p - n);
  disp = (int16_t)(((NDISP - mind - 1) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4);
  cost = (uint16_t)minsad;
}

// grid: (ceil((width1-3)/64), ceil(h/BM_STRIP), batch), block 64.  Output columns x in [3, width1), X = x + 31.
// The vertical 7-row sum slides down the strip: the entering row's SADs are added, the leaving row's are recomputed and
// subtracted (cheaper than keeping a ring of 7 rows x 32 SADs: no LDS traffic, and occupancy is not LDS-bound); the
// next step's two row windows are loaded before the current step's arithmetic.
__global__ __launch_bounds__(64) void stereo_bm_kernel(StereoDev S) {
  __shared__ SelScr s_scr[64];
  const int lane = threadIdx.x, b = blockIdx.z;
  const int w = S.w, h = S.h, width1 = w - NDISP + 1;
  const int x = min(3 + blockIdx.x * 64 + lane, width1 - 1);      // lanes past the end redo the last column (no store)
  const bool store = 3 + blockIdx.x * 64 + lane < width1;
  const int y0 = blockIdx.y * BM_STRIP, y1 = min(y0 + BM_STRIP, h);
  const uint8_t *lp = S.lp + (size_t)b * h * S.pitch + PADL + x + (NDISP - 1) - WSZ2;
  const uint8_t *rp = S.rp + (size_t)b * h * S.pitch + PADL + x - WSZ2;
  const uint32_t ft4 = 0x01010101u * (uint32_t)(S.cap + 1);
  auto win = [&](int r) { const size_t o = (size_t)min(max(r, 0), h - 1) * S.pitch; return load_rowwin(lp + o, rp + o); };
  Pk sad[8], hh[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) sad[g].q = 0;
  int tsum = 0, t;
  for (int r = y0 - WSZ2; r < y0 + WSZ2; ++r) {      // rows y0-3 .. y0+2 of the first window
    const RowWin wv = win(r);
    row_sads(wv, ft4, hh, t);
#pragma unroll
    for (int g = 0; g < 8; ++g) { sad[g].h[0] += hh[g].h[0]; sad[g].h[1] += hh[g].h[1]; }
    tsum += t;
  }
  RowWin wa = win(y0 + WSZ2), ws = win(y0 - WSZ2);
  for (int y = y0; y < y1; ++y) {
    const RowWin na = win(y + WSZ2 + 1), ns = win(y - WSZ2 + 1);      // next step's rows, in flight during this step
    row_sads(wa, ft4, hh, t);
#pragma unroll
    for (int g = 0; g < 8; ++g) { sad[g].h[0] += hh[g].h[0]; sad[g].h[1] += hh[g].h[1]; }
    tsum += t;
    int16_t d16; uint16_t c16;
    bm_select(sad, tsum, S, s_scr[lane], d16, c16);
    if (store) {
      const size_t o = ((size_t)b * h + y) * w + x + (NDISP - 1);
      S.disp16[o] = d16; S.cost[o] = c16;
    }
    row_sads(ws, ft4, hh, t);
#pragma unroll
    for (int g = 0; g < 8; ++g) { sad[g].h[0] -= hh[g].h[0]; sad[g].h[1] -= hh[g].h[1]; }
    tsum -= t;
    wa = na; ws = ns;
  }
}

// columns X < 31 (never searched) and x in [0,3), whose right-image window clamps at column 0 BEFORE the disparity
// shift (so it is not a contiguous byte window).  Half a wave per pixel: lane = disparity index d.
// grid: (ceil(3*h/2), batch), block 64
__global__ __launch_bounds__(64) void stereo_bm_edge_kernel(StereoDev S) {
  const int b = blockIdx.y, w = S.w, h = S.h, lane = threadIdx.x, d = lane & 31;
  const int width1 = w - NDISP + 1, ncol = min(3, width1);
  {   // the never-searched left border, FILTERED
    const int n = (NDISP - 1) * h;
    for (int i = blockIdx.x * 64 + lane; i < n; i += gridDim.x * 64) {
      const size_t o = ((size_t)b * h + i / (NDISP - 1)) * w + i % (NDISP - 1);
      S.disp16[o] = (int16_t)FILTERED16; S.cost[o] = 0;
    }
  }
  const int pix = blockIdx.x * 2 + (lane >> 5);
  const bool act = pix < ncol * h;
  const int y = act ? pix / ncol : 0, x = act ? pix % ncol : 0;
  const uint8_t *lp = S.lp + (size_t)b * h * S.pitch + PADL, *rp = S.rp + (size_t)b * h * S.pitch + PADL;
  int sad = 0, tsum = 0;
  for (int dy = -WSZ2; dy <= WSZ2; ++dy) {
    const int yy = min(max(y + dy, 0), h - 1);
#pragma unroll
    for (int dx = -WSZ2; dx <= WSZ2; ++dx) {
      const int lval = lp[(size_t)yy * S.pitch + x + dx + NDISP - 1];      // x + dx + 31 >= 28: no clamp on the left image here
      sad += abs(lval - (int)rp[(size_t)yy * S.pitch + min(max(x + dx, 0) + d, w - 1)]);
      tsum += abs(lval - (S.cap + 1));
    }
  }
  // selection across the 32 lanes of the pixel (same rules as bm_select)
  uint32_t key = ((uint32_t)sad << 5) | (uint32_t)d;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o, 64));
  const int minsad = (int)(key >> 5), mind = (int)(key & 31), base = lane & 32;
  const int ip = mind == NDISP - 1 ? NDISP - 2 : mind + 1, in = mind == 0 ? 1 : mind - 1;
  const int p = __shfl(sad, base + ip, 64), n = __shfl(sad, base + in, 64);
  const int thresh = minsad + (minsad * S.uniq / 100);
  const unsigned long long rivals = __ballot(sad <= thresh && (d < mind - 1 || d > mind + 1));
  const bool rival = ((rivals >> base) & 0xffffffffull) != 0;
  if (!act || d != 0) return;
  int16_t d16 = (int16_t)FILTERED16; uint16_t c16 = 0;
  if (tsum >= S.texthr && !(S.uniq > 0 && rival)) {
    const int dd = p + n - 2 * minsad + abs(p - n);
    d16 = (int16_t)(((NDISP - mind - 1) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4);
    c16 = (uint16_t)minsad;
  }
  const size_t o = ((size_t)b * h + y) * w + x + (NDISP - 1);
  S.disp16[o] = d16; S.cost[o] = c16;
}

// validateDisparity (with D2): one workgroup per image row.  The reference's sequential first pass ("a strictly smaller
// cost wins", ascending x => first x on ties) is a minimum over (cost, x): every valid source pixel does one LDS atomicMin
// of the packed key (cost << 16 | x) on its target column.  grid: (h, batch), block 256, dynamic LDS = 2 * w ints
__global__ __launch_bounds__(256) void stereo_validate_kernel(StereoDev S) {
  extern __shared__ int s_mem[];
  const int w = S.w, y = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  int *s_d = s_mem;
  unsigned *s_key = reinterpret_cast<unsigned *>(s_mem + w);
  const int SCALE = 1 << DISP_SHIFT, INVALID = -SCALE, maxdiff = S.disp12 * SCALE;
  int16_t *dp = S.disp16 + ((size_t)b * S.h + y) * w;
  const uint16_t *cp = S.cost + ((size_t)b * S.h + y) * w;
  for (int x = tid; x < w; x += 256) { s_d[x] = dp[x]; s_key[x] = 0xffffffffu; }
  __syncthreads();
  const int minX1 = NDISP;
  for (int x = minX1 + tid; x < w; x += 256) {
    const int d = s_d[x];
    if (d == INVALID) continue;
    const int x2 = x - ((d + SCALE / 2) >> DISP_SHIFT);
    if (x2 >= 0 && x2 < w) atomicMin(&s_key[x2], ((unsigned)cp[x] << 16) | (unsigned)x);
  }
  __syncthreads();
  for (int x = minX1 + tid; x < w; x += 256) {
    const int d = s_d[x];
    if (d == INVALID) continue;
    const int x0 = x - (d >> DISP_SHIFT), x1 = x - ((d + SCALE - 1) >> DISP_SHIFT);
    bool bad0 = false, bad1 = false;
    if (x0 >= 0 && x0 < w) { const unsigned k = s_key[x0]; bad0 = k != 0xffffffffu && abs(s_d[k & 0xffffu] - d) > maxdiff; }
    if (x1 >= 0 && x1 < w) { const unsigned k = s_key[x1]; bad1 = k != 0xffffffffu && abs(s_d[k & 0xffffu] - d) > maxdiff; }
    if (bad0 && bad1) dp[x] = (int16_t)INVALID;
  }
}

// ---- speckle filter: connected components by union-find (labels = linear pixel index inside the frame) ----------
// Labels: -1 = filtered pixel, CCL_BIG = member of a component already known to exceed the speckle window (any
// component containing a horizontal run longer than the window), otherwise the index of a pixel of the same set
// (roots point at themselves).  CCL_BIG is smaller than every index, so "hang the larger root under the smaller"
// (atomicMin) makes it absorb whatever gets connected to it.
constexpr int CCL_BIG = -2;
__device__ __forceinline__ int ccl_find(const int32_t *label, int x) {
  int p = label[x];
  while (p != x && p >= 0) { x = p; p = label[x]; }
  return p < 0 ? CCL_BIG : x;
}
__device__ __forceinline__ void ccl_union(int32_t *label, int a, int b) {      // a, b: node indices or CCL_BIG
  while (true) {
    if (a >= 0) a = ccl_find(label, a);
    if (b >= 0) b = ccl_find(label, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }      // a > b >= CCL_BIG: hang the larger root under the smaller
    const int old = atomicMin(&label[a], b);
    if (old == a) return;
    a = old;                                            // a was no longer a root: continue from its new parent
  }
}
__device__ __forceinline__ bool ccl_linked(int a, int b, int range) { return a != FILTERED16 && b != FILTERED16 && abs(a - b) <= range; }

// Horizontal runs: label = index of the first pixel of the maximal run of horizontally linked pixels, so the union-find
// only has to stitch rows together and its chains stay short; runs longer than the speckle window are labelled CCL_BIG
// right away -- in a real disparity map that is most valid pixels, and their vertical links cost nothing later.
// One workgroup per image row, one wave per 64-pixel segment: "last run start at or left of me" and "next run boundary
// right of me" come from two ballots per segment plus a carry over the (few) segments of the row -- three barriers per
// row.  The start pixel of every small run gets count = 0 and its length in `rlen` (the cost plane, free after the
// LR check; 0 everywhere else): the later passes find the small runs through it and nothing else needs clearing.
// grid: (h, batch), block 256, dynamic LDS = (2 w + 2 nseg) ints
__global__ __launch_bounds__(256) void stereo_ccl_runs_kernel(StereoDev S) {
  extern __shared__ int s_mem[];
  const int w = S.w, y = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t base = (size_t)blockIdx.y * w * S.h + (size_t)y * w;
  const int16_t *d = S.disp16 + base;
  const int nseg = (w + 63) >> 6;
  int *s_st = s_mem, *s_len = s_mem + w, *s_segl = s_len + w, *s_segf = s_segl + nseg;
  for (int seg = wave; seg < nseg; seg += 4) {
    const int x = seg * 64 + lane;
    const bool in = x < w;
    const int dv = in ? d[x] : FILTERED16, dl = (in && x > 0) ? d[x - 1] : FILTERED16;
    const bool filt = dv == FILTERED16;                                        // lanes beyond the row end act as a boundary
    const bool start = !filt && !(x > 0 && ccl_linked(dv, dl, S.speckle_range));
    const unsigned long long sb = __ballot(start), bb = __ballot(start || filt);
    const unsigned long long lower = sb & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    if (in) {
      s_st[x] = filt ? -2 : (lower ? seg * 64 + 63 - __clzll((long long)lower) : -1);      // -1: the run started in an earlier segment
      const unsigned long long above = lane == 63 ? 0ull : (bb >> (lane + 1));
      s_len[x] = above ? x + __ffsll((long long)above) : -1;                               // next boundary; -1: in a later segment
    }
    if (lane == 0) {
      s_segl[seg] = sb ? seg * 64 + 63 - __clzll((long long)sb) : -1;
      s_segf[seg] = bb ? seg * 64 + __ffsll((long long)bb) - 1 : 0x7fffffff;
    }
  }
  __syncthreads();
  if (tid == 0) { int run = -1; for (int k = 0; k < nseg; ++k) { const int v = s_segl[k]; s_segl[k] = run; run = max(run, v); } }              // exclusive prefix max
  if (tid == 64) { int run = w; for (int k = nseg - 1; k >= 0; --k) { const int v = s_segf[k]; s_segf[k] = run; run = min(run, v); } }          // exclusive suffix min
  __syncthreads();
  for (int x = tid; x < w; x += 256) {
    const int seg = x >> 6;
    int st = s_st[x];
    if (st == -1) { st = s_segl[seg]; s_st[x] = st; }
    if (st == x) { int nb = s_len[x]; if (nb < 0) nb = s_segf[seg]; s_len[x] = min(nb, w) - x; }
  }
  __syncthreads();
  uint16_t *rlen = S.cost + base;
  for (int x = tid; x < w; x += 256) {
    const int st = s_st[x];
    int lab = -1, rl = 0;
    if (st >= 0) {
      const int len = s_len[st];
      const bool big = len > S.speckle_window;
      lab = big ? CCL_BIG : y * w + st;
      if (!big && st == x) { rl = len; S.count[base + x] = 0; }
    }
    S.label[base + x] = lab;
    rlen[x] = (uint16_t)rl;
  }
}
// stitch vertically linked pixels; one union per pair of overlapping runs (the leftmost linked column of the overlap).
// grid: (ceil(w*h/256), batch) -- scalar variant for widths that are not a multiple of 4
__global__ __launch_bounds__(256) void stereo_ccl_merge_kernel(StereoDev S) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = S.w * S.h, w = S.w;
  if (i + w >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  const int16_t *d = S.disp16 + base;
  int32_t *label = S.label + base;
  const int la = label[i], lb = label[i + w];
  if (la == -1 || lb == -1 || (la == CCL_BIG && lb == CCL_BIG)) return;      // filtered, or both sides already known to be big
  if (!ccl_linked(d[i], d[i + w], S.speckle_range)) return;
  if (i % w > 0 && label[i - 1] == la && label[i + w - 1] == lb && ccl_linked(d[i - 1], d[i + w - 1], S.speckle_range)) return;
  ccl_union(label, la, lb);      // run labels are only ever replaced by other members of the same set (or CCL_BIG)
}
// same, four horizontally adjacent pixels per lane (w % 4 == 0): the labels of both rows arrive as two 16-byte loads and
// most lanes leave right there (filtered, or big above big); grid: (ceil(w*h/1024), batch)
__global__ __launch_bounds__(256) void stereo_ccl_merge4_kernel(StereoDev S) {
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4, n = S.w * S.h, w = S.w;
  if (i0 + w >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  const int16_t *d = S.disp16 + base;
  int32_t *label = S.label + base;
  const int4 a4 = *reinterpret_cast<const int4 *>(label + i0), b4 = *reinterpret_cast<const int4 *>(label + i0 + w);
  const int la[4] = {a4.x, a4.y, a4.z, a4.w}, lb[4] = {b4.x, b4.y, b4.z, b4.w};
  bool cand[4], any = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) { cand[k] = !(la[k] == -1 || lb[k] == -1 || (la[k] == CCL_BIG && lb[k] == CCL_BIG)); any |= cand[k]; }
  if (!any) return;
  const uint2 da2 = *reinterpret_cast<const uint2 *>(d + i0), db2 = *reinterpret_cast<const uint2 *>(d + i0 + w);
  const int da[4] = {(int16_t)(da2.x & 0xffff), (int16_t)(da2.x >> 16), (int16_t)(da2.y & 0xffff), (int16_t)(da2.y >> 16)};
  const int db[4] = {(int16_t)(db2.x & 0xffff), (int16_t)(db2.x >> 16), (int16_t)(db2.y & 0xffff), (int16_t)(db2.y >> 16)};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!cand[k] || !ccl_linked(da[k], db[k], S.speckle_range)) continue;
    bool covered;
    if (k > 0) covered = la[k - 1] == la[k] && lb[k - 1] == lb[k] && ccl_linked(da[k - 1], db[k - 1], S.speckle_range);
    else covered = i0 % w > 0 && label[i0 - 1] == la[0] && label[i0 + w - 1] == lb[0] && ccl_linked(d[i0 - 1], d[i0 + w - 1], S.speckle_range);
    if (!covered) ccl_union(label, la[k], lb[k]);
  }
}
// component sizes, saturating: the test is "size <= speckle_window", so a root that is already beyond the window is left
// alone.  Only the start pixels of small runs carry a length (rlen != 0): eight pixels per lane, most lanes see zeros.
// Members of CCL_BIG need no count.  grid: (ceil(w*h/2048), batch)
__global__ __launch_bounds__(256) void stereo_ccl_count_kernel(StereoDev S) {
  const int n = S.w * S.h, i0 = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (i0 >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  const uint16_t *rlen = S.cost + base;
  int len[8];
  if (((base | (size_t)i0) & 7) == 0 && i0 + 8 <= n) {
    const uint4 r = *reinterpret_cast<const uint4 *>(rlen + i0);
    if ((r.x | r.y | r.z | r.w) == 0) return;
    len[0] = r.x & 0xffff; len[1] = r.x >> 16; len[2] = r.y & 0xffff; len[3] = r.y >> 16;
    len[4] = r.z & 0xffff; len[5] = r.z >> 16; len[6] = r.w & 0xffff; len[7] = r.w >> 16;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) len[k] = i0 + k < n ? rlen[i0 + k] : 0;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (len[k] == 0) continue;
    const int root = ccl_find(S.label + base, i0 + k);
    if (root != i0 + k) S.label[base + i0 + k] = root;      // flatten: the pixels of this run reach the root (or CCL_BIG) in one hop;
                                                            // roots stay roots, so concurrent finds through this node remain valid
    if (root >= 0 && S.count[base + root] <= S.speckle_window) atomicAdd(&S.count[base + root], len[k]);
  }
}
// disparity in pixels; components of <= speckle_window pixels are filtered.  use_ccl == 0: plain conversion
__global__ __launch_bounds__(256) void stereo_finish_kernel(StereoDev S, int use_ccl, float *__restrict__ out, int dstride, size_t d_bstride) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = S.w * S.h;
  if (i >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  int d = S.disp16[base + i];
  if (use_ccl && d != FILTERED16) {
    const int l = S.label[base + i];                     // start pixel of the pixel's run, or CCL_BIG
    if (l >= 0) {
      const int root = S.label[base + l];                // flattened by the count pass
      if (root >= 0 && S.count[base + root] <= S.speckle_window) d = FILTERED16;
    }
  }
  out[(size_t)blockIdx.y * d_bstride + (size_t)(i / S.w) * dstride + (i % S.w)] = (float)d * (1.f / (1 << DISP_SHIFT));
}

// same, four pixels per lane (w % 4 == 0, so the four share a row): one 8-byte disparity load, one 16-byte label load;
// grid: (ceil(w*h/1024), batch)
__global__ __launch_bounds__(256) void stereo_finish4_kernel(StereoDev S, int use_ccl, float *__restrict__ out, int dstride, size_t d_bstride) {
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4, n = S.w * S.h;
  if (i0 >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  const uint2 d2 = *reinterpret_cast<const uint2 *>(S.disp16 + base + i0);
  int d[4] = {(int16_t)(d2.x & 0xffff), (int16_t)(d2.x >> 16), (int16_t)(d2.y & 0xffff), (int16_t)(d2.y >> 16)};
  if (use_ccl) {
    const int4 l4 = *reinterpret_cast<const int4 *>(S.label + base + i0);
    const int l[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (d[k] != FILTERED16 && l[k] >= 0) {
        const int root = S.label[base + l[k]];           // flattened by the count pass
        if (root >= 0 && S.count[base + root] <= S.speckle_window) d[k] = FILTERED16;
      }
  }
  float *o = out + (size_t)blockIdx.y * d_bstride + (size_t)(i0 / S.w) * dstride + (i0 % S.w);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (float)d[k] * (1.f / (1 << DISP_SHIFT));
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
      w < NDISP + 2 * WSZ2 || w > 65535 || h < 2 || prm->speckle_window > 65535) {
    ctx->err = "svs_stereo: only SADWindowSize 7, minDisparity 0, numberOfDisparities 32, preFilterCap 1..63, w >= 38 are supported";
    return SVS_ERR_UNSUPPORTED;
  }
  svs_stereo *s = new svs_stereo();
  s->ctx = ctx; s->w = w; s->h = h; s->max_batch = max_batch; s->pitch = (w + PADL + PADR + 3) & ~3; s->prm = *prm;
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
  hipLaunchKernelGGL(stereo_prefilter_kernel, dim3(div_up(s->pitch, 256), div_up(h, 4), 2 * n_batch), dim3(64, 4), 0, ctx->stream, S, d_left, lstride,
                     l_bstride, d_right, rstride, r_bstride);
  SVS_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(stereo_bm_edge_kernel, dim3(div_up(3 * h, 2), n_batch), dim3(64), 0, ctx->stream, S);
  SVS_LAUNCH_CHECK(ctx);
  if (width1 > 3) {
    hipLaunchKernelGGL(stereo_bm_kernel, dim3(div_up(width1 - 3, 64), div_up(h, BM_STRIP), n_batch), dim3(64), 0, ctx->stream, S);
    SVS_LAUNCH_CHECK(ctx);
  }
  if (s->prm.disp12_max_diff >= 0) {
    hipLaunchKernelGGL(stereo_validate_kernel, dim3(h, n_batch), dim3(256), sizeof(int) * 2 * (size_t)w, ctx->stream, S);
    SVS_LAUNCH_CHECK(ctx);
  }
  const bool ccl = s->prm.speckle_range >= 0 && s->prm.speckle_window > 0;
  const dim3 gp(div_up(n, 256), n_batch);
  if (ccl) {
    hipLaunchKernelGGL(stereo_ccl_runs_kernel, dim3(h, n_batch), dim3(256), sizeof(int) * (2 * (size_t)w + 2 * (size_t)((w + 63) / 64)), ctx->stream, S);
    SVS_LAUNCH_CHECK(ctx);
    if (w % 4 == 0) hipLaunchKernelGGL(stereo_ccl_merge4_kernel, dim3(div_up(n, 1024), n_batch), dim3(256), 0, ctx->stream, S);
    else hipLaunchKernelGGL(stereo_ccl_merge_kernel, gp, dim3(256), 0, ctx->stream, S);
    SVS_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(stereo_ccl_count_kernel, dim3(div_up(n, 2048), n_batch), dim3(256), 0, ctx->stream, S); SVS_LAUNCH_CHECK(ctx);
  }
  if (w % 4 == 0) hipLaunchKernelGGL(stereo_finish4_kernel, dim3(div_up(n, 1024), n_batch), dim3(256), 0, ctx->stream, S, ccl ? 1 : 0, d_disp, dstride, d_bstride);
  else hipLaunchKernelGGL(stereo_finish_kernel, gp, dim3(256), 0, ctx->stream, S, ccl ? 1 : 0, d_disp, dstride, d_bstride);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
