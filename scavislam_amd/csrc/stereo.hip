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
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <vector>

namespace {

constexpr int PADL = 16, PADR = 48;         // prefiltered rows are padded: pitch = w + PADL + PADR
constexpr int NDISP = 32, WSZ2 = 3;
constexpr int BM_STRIP = 96;                // output rows per wave (the six rows above a strip are summed without producing output)
constexpr int DISP_SHIFT = 4;
constexpr int FILTERED16 = -(1 << DISP_SHIFT);      // (minDisparity - 1) << 4 with minDisparity 0

struct StereoDev {
  int w, h, pitch;
  int cap, texthr, uniq, speckle_window, speckle_range, disp12;
  uint8_t *lp, *rp;      // [batch][h][pitch], values + 1
  int16_t *disp16;       // [batch][h][w]
  uint16_t *cost;        // [batch][h][w]
  int32_t *label, *count;// [batch][h*w]
  // strip speckle filter: components still undecided at strip boundaries
  int strip_rows, n_strips;
  int32_t *brow;         // [batch][2 * n_strips][w]: component id of the first / last row of every strip (-1 filtered, -2 big, else node)
  int2 *pend;            // [batch][h*w]: (pixel, node) of the pixels of undecided components, a list per strip (at the strip's first pixel)
  int32_t *npend;        // [batch][n_strips]
  int timing;            // SVS_STEREO_DEBUG: phase stamps of one workgroup behind the error word
  int *err;              // [1] set when a bounded walk of the strip path gave up (never expected)
  int swz;               // workgroups in XCD-contiguous order (common.h: xcd_contiguous)
};

__device__ __forceinline__ void wave_sync_lds() { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }
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
// 4 padded columns x 4 rows per thread (the six input rows of a 4-row group are read once: 12 dword loads per 16 pixels instead of 24; the
// horizontal differences of an input row serve three output rows), 16 rows per workgroup.
// grid: (ceil(pitch/256), ceil(h/16), 2*batch), block (64, 4); z = 2*b + (0 left | 1 right)
__global__ __launch_bounds__(256) void stereo_prefilter_kernel(StereoDev S, const uint8_t *__restrict__ left, int lstride, size_t l_bstride,
                                                               const uint8_t *__restrict__ right, int rstride, size_t r_bstride) {
  const int pc0 = 4 * (blockIdx.x * 64 + threadIdx.x), y0 = (blockIdx.y * 4 + threadIdx.y) * 4, b = blockIdx.z >> 1, side = blockIdx.z & 1;
  const int w = S.w, h = S.h, cap = S.cap;
  if (pc0 >= S.pitch || y0 >= h) return;      // pitch is a multiple of 4
  const uint8_t *src = side ? right + (size_t)b * r_bstride : left + (size_t)b * l_bstride;
  const int stride = side ? rstride : lstride;
  const int x0 = pc0 - PADL;
  const uint32_t fill = 0x01010101u * (uint32_t)(cap + 1);      // pads, border columns, odd last row
  // horizontal differences byte(j + 2) - byte(j), j = 0..3, of the six input rows y0-1 .. y0+4 (rows outside the image: the row the
  // original's yp / yn pick, i.e. 1 above the top and h-2 below the bottom)
  int d[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int r = y0 - 1 + i;
    r = r < 0 ? (h > 1 ? 1 : 0) : (r > h - 1 ? (h > 1 ? h - 2 : 0) : r);
    r = min(max(r, 0), h - 1);                                  // (rows far below the image only feed rows that are not written)
    const uint8_t *row = src + (size_t)r * stride;
    const uint32_t a0 = load_u32_clamped(row, x0 - 1, w), a1 = load_u32_clamped(row, x0 + 3, w);
#pragma unroll
    for (int j = 0; j < 4; ++j) d[i][j] = byte_of(a0, a1, j + 2) - byte_of(a0, a1, j);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = y0 + k;
    if (y >= h) break;
    uint32_t out = fill;
    if (!((h & 1) && y == h - 1)) {
      out = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = x0 + j;
        const int v = d[k][j] + 2 * d[k + 1][j] + d[k + 2][j];
        const uint32_t o = (x >= 1 && x <= w - 2) ? (uint32_t)(xsobel_tab(v, cap) + 1) : (uint32_t)(cap + 1);
        out |= o << (8 * j);
      }
    }
    *reinterpret_cast<uint32_t *>((side ? S.rp : S.lp) + ((size_t)b * h + y) * S.pitch + pc0) = out;
  }
}


// The same filter for rows of 16 n pixels on dword-aligned sources (every camera of the reference): 16 padded columns x 4 rows per thread.  The six input rows are
// read as 24 bytes each (one dwordx4 + the dword on either side), the arithmetic runs on pairs of pixels in packed 16-bit lanes (V_PERM_B32 picks the byte pairs:
// the pair (b[4m-1], b[4m]) is the minus operand of pixel pair 4m and the plus operand of pair 4m-2, (b[4m+1], b[4m+2]) the plus operand of 4m and the minus operand
// of 4m+2 -- nine shuffles and eight packed subtractions per input row), and a row of 16 results leaves as one 16-byte store.  Threads are dealt linearly over
// (row group, column chunk), so no lane idles on the 44 chunks of a 704-byte row.  grid: (ceil(n_chunks * n_rowgroups / 256), 1, 2 * batch), block 256
typedef short ps2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ps2 as_ps2(uint32_t v) { ps2 r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ uint32_t as_u32(ps2 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
__global__ __launch_bounds__(256) void stereo_prefilter16_kernel(StereoDev S, const uint8_t *__restrict__ left, int lstride, size_t l_bstride,
                                                                 const uint8_t *__restrict__ right, int rstride, size_t r_bstride) {
  const int w = S.w, h = S.h, cap = S.cap, nct = S.pitch >> 4, nrg = (h + 3) >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.z >> 1, side = blockIdx.z & 1;
  if (t >= nct * nrg) return;
  const int rg = t / nct, ct = t - rg * nct, y0 = 4 * rg, x0 = 16 * ct - PADL;
  const uint32_t fill = 0x01010101u * (uint32_t)(cap + 1);      // pads, border columns, odd last row
  uint8_t *dst = (side ? S.rp : S.lp) + ((size_t)b * h + y0) * S.pitch + 16 * ct;
  const bool inside = x0 >= 0 && x0 < w;                        // (w is a multiple of 16: a chunk is inside the image or inside a pad)
  uint4 out[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = make_uint4(fill, fill, fill, fill);
  if (inside) {
    const uint8_t *src = side ? right + (size_t)b * r_bstride : left + (size_t)b * l_bstride;
    const int stride = side ? rstride : lstride;
    const bool has_l = x0 >= 4, has_r = x0 + 20 <= w;
    ps2 d[6][8];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      int r = y0 - 1 + i;      // rows outside the image: the row the original's yp / yn pick, i.e. 1 above the top and h-2 below the bottom
      r = r < 0 ? (h > 1 ? 1 : 0) : (r > h - 1 ? (h > 1 ? h - 2 : 0) : r);
      r = min(max(r, 0), h - 1);                                // (rows far below the image only feed rows that are not written)
      const uint8_t *row = src + (size_t)r * stride + x0;
      const uint4 a = *reinterpret_cast<const uint4 *>(row);
      uint32_t W[6];
      W[0] = has_l ? *reinterpret_cast<const uint32_t *>(row - 4) : 0u;      // bytes outside the row only feed border columns, which are overwritten
      W[1] = a.x; W[2] = a.y; W[3] = a.z; W[4] = a.w;
      W[5] = has_r ? *reinterpret_cast<const uint32_t *>(row + 16) : 0u;
      ps2 E[5], O[4];
#pragma unroll
      for (int m = 0; m < 5; ++m) E[m] = as_ps2(__builtin_amdgcn_perm(W[m + 1], W[m], 0x0c040c03u));      // (b[4m-1], b[4m])
#pragma unroll
      for (int m = 0; m < 4; ++m) O[m] = as_ps2(__builtin_amdgcn_perm(0u, W[m + 1], 0x0c020c01u));         // (b[4m+1], b[4m+2])
#pragma unroll
      for (int m = 0; m < 4; ++m) { d[i][2 * m] = O[m] - E[m]; d[i][2 * m + 1] = E[m + 1] - O[m]; }
    }
    const ps2 lo = {(short)-cap, (short)-cap}, hi = {(short)cap, (short)cap}, off = {(short)(cap + 1), (short)(cap + 1)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ps2 v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int p = 2 * q + e;
          const ps2 s3 = d[k][p] + d[k + 1][p] + d[k + 1][p] + d[k + 2][p];
          v[e] = __builtin_elementwise_min(__builtin_elementwise_max(s3, lo), hi) + off;
        }
        o[q] = __builtin_amdgcn_perm(as_u32(v[1]), as_u32(v[0]), 0x06040200u);
      }
      if (x0 == 0) o[0] = (o[0] & 0xffffff00u) | (fill & 0x000000ffu);               // column 0
      if (x0 + 16 == w) o[3] = (o[3] & 0x00ffffffu) | (fill & 0xff000000u);          // column w-1
      out[k] = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = y0 + k;
    if (y >= h) break;
    *reinterpret_cast<uint4 *>(dst + (size_t)k * S.pitch) = ((h & 1) && y == h - 1) ? make_uint4(fill, fill, fill, fill) : out[k];
  }
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
// (p - n) * 256 / dd of the sub-pixel step (C division: towards zero) without the 30-instruction integer division: |p - n| * 256 < 2^24 and dd < 2^15 are exact
// floats, the product with the reciprocal is within 0.4 of the quotient, one step up or down on the integer remainder makes it exact
__device__ __forceinline__ int div_trunc_small(int num, int den) {      // |num| < 2^24, 0 < den < 2^15
  const int a = abs(num);
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)den));
  const int rem = a - q * den;
  q += rem >= den ? 1 : (rem < 0 ? -1 : 0);
  return num < 0 ? -q : q;
}
template <bool SMALL>      // SMALL: every SAD < 4096 (prefilter cap <= 41)
__device__ __forceinline__ void bm_select(const Pk (&sad)[8], int tsum, const StereoDev &S, SelScr &scr, int16_t &disp, uint16_t &cost) {
  disp = (int16_t)FILTERED16; cost = 0;
  int minsad, mind;
  if (SMALL) {      // 49 taps x |difference| <= 2 cap: every SAD < 4096, so a 16-bit KEY holds sad << 4 | dword index and both halves of the 16 dwords are searched
    us2 m = {0xffff, 0xffff};      // at once; first minimum wins within a half, d = 2 * index + half.  The kernel keeps its running sums AS keys (the index rides in the four
#pragma unroll                   // low bits, which the shifted row SADs never touch): the search is sixteen packed minima
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int j = 0; j < 2; ++j) m = __builtin_elementwise_min(m, sad[g].h[j]);
    const int s0 = m.x >> 4, d0 = 2 * (m.x & 15), s1 = m.y >> 4, d1 = 2 * (m.y & 15) + 1;
    const bool first = s0 < s1 || (s0 == s1 && d0 < d1);
    minsad = first ? s0 : s1; mind = first ? d0 : d1;
  } else {
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int g = 0; g < 8; ++g) {      // first minimum wins: key = sad * 32 + d
      best = min(best, ((sad[g].u[0] & 0xffffu) << 5) | (uint32_t)(4 * g));
      best = min(best, ((sad[g].u[0] >> 16) << 5) | (uint32_t)(4 * g + 1));
      best = min(best, ((sad[g].u[1] & 0xffffu) << 5) | (uint32_t)(4 * g + 2));
      best = min(best, ((sad[g].u[1] >> 16) << 5) | (uint32_t)(4 * g + 3));
    }
    minsad = (int)(best >> 5); mind = (int)(best & 31);
  }
  if (tsum < S.texthr) return;
#pragma unroll
  for (int g = 0; g < 8; ++g) { scr.u[2 * g] = sad[g].u[0]; scr.u[2 * g + 1] = sad[g].u[1]; }
  unsigned short *s16 = scr.s;
  constexpr int KS = SMALL ? 4 : 0;      // the scratch holds keys (SMALL) or plain sums
  // sad[mind + 1], sad[mind - 1] with the mirrored ends sad[-1] = sad[1], sad[32] = sad[30]
  const int p = s16[mind == NDISP - 1 ? NDISP - 2 : mind + 1] >> KS, n = s16[mind == 0 ? 1 : mind - 1] >> KS;
  if (S.uniq > 0) {
    // the scan of the original stops at a d outside [mind-1, mind+1] with sad[d] <= thresh: mask those three, take the min
    const int thresh = minsad + (minsad * S.uniq / 100);
    s16[mind] = 0xffff;
    s16[mind == 0 ? 32 : mind - 1] = 0xffff;            // slots 32/33 = spare halfword pair
    s16[mind == NDISP - 1 ? 33 : mind + 1] = 0xffff;
    us2 m = {0xffff, 0xffff};
#pragma unroll
    for (int k = 0; k < 16; ++k) { Pk t; t.u[0] = scr.u[k]; m = __builtin_elementwise_min(m, t.h[0]); }
    if ((int)(min(m.x, m.y) >> KS) <= thresh) return;      // (keys: a masked slot reads 4095 >= every real sum, and 29 real sums are always in the minimum)
  }
  const int dd = p + n - 2 * minsad + abs(p - n);
  disp = (int16_t)(((NDISP - mind - 1) * 256 + (dd != 0 ? (SMALL ? div_trunc_small((p - n) * 256, dd) : (p - n) * 256 / dd) : 0) + 15) >> 4);
  cost = (uint16_t)minsad;
}

// grid: (ceil((width1-3)/64), ceil(h/BM_STRIP), batch), block 64.  Output columns x in [3, width1), X = x + 31.
// The vertical 7-row sum slides down the strip: the entering row's 32 SADs are added, the leaving row's are subtracted.  The six rows
// of the window live in a register ring (7 slots x 16 VGPRs, slot = row mod 7: the entering row y+3 is computed straight into the slot
// the row that left one step earlier freed; the loop is unrolled by seven so that the slots are static) -- one V_QSAD / V_MQSAD pass and one row window load per output row
// instead of two (round 2 recomputed the leaving row); the next step's row window is loaded as soon as the current one is consumed.
template <bool SMALL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void stereo_bm_kernel(StereoDev S) {
  __shared__ SelScr s_scr[64];
  // column neighbours share 44 of their right-image bytes per row, strip neighbours six rows: an XCD works through whole frames (xcd_contiguous), so those lines
  // are fetched into one L2
  unsigned wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  if (S.swz) wg = xcd_contiguous(wg, gridDim.x * gridDim.y * gridDim.z);
  const int bx = wg % gridDim.x, by = (wg / gridDim.x) % gridDim.y;
  const int lane = threadIdx.x, b = wg / (gridDim.x * gridDim.y);
  const int w = S.w, h = S.h, width1 = w - NDISP + 1;
  const int x = min(3 + bx * 64 + lane, width1 - 1);      // lanes past the end redo the last column (no store)
  const bool store = 3 + bx * 64 + lane < width1;
  const int y0 = by * BM_STRIP, y1 = min(y0 + BM_STRIP, h);
  const uint8_t *lp = S.lp + (size_t)b * h * S.pitch + PADL + (NDISP - 1) - WSZ2;      // uniform bases + one 32-bit lane offset for both images
  const uint8_t *rp = S.rp + (size_t)b * h * S.pitch + PADL - WSZ2;
  const uint32_t ft4 = 0x01010101u * (uint32_t)(S.cap + 1);
  auto win = [&](int r) { const uint32_t o = (uint32_t)min(max(r, 0), h - 1) * (uint32_t)S.pitch + (uint32_t)x; return load_rowwin(lp + o, rp + o); };
  constexpr int NR = 2 * WSZ2 + 1;      // ring slots = window rows
  Pk sad[8], ring[NR][8];
  int tring[NR];
  // SMALL: the running sums are kept as the selection's KEYS, sum << 4 | dword index (bm_select): the index is put in once, the row SADs enter and leave shifted
  // (one packed shift per dword of the entering row instead of a shift and an OR per dword of every selection)
#pragma unroll
  for (int g = 0; g < 8; ++g) { sad[g].u[0] = SMALL ? 0x00010001u * (uint32_t)(2 * g) : 0u; sad[g].u[1] = SMALL ? 0x00010001u * (uint32_t)(2 * g + 1) : 0u; }
  auto as_keys = [](Pk (&r)[8]) __attribute__((always_inline)) {
    if (SMALL) {
#pragma unroll
      for (int g = 0; g < 8; ++g) { r[g].h[0] <<= 4; r[g].h[1] <<= 4; }
    }
  };
  int tsum = 0;
#pragma unroll
  for (int j = 0; j < NR - 1; ++j) {      // rows y0-3 .. y0+2 of the first window
    const RowWin wv = win(y0 - WSZ2 + j);
    row_sads(wv, ft4, ring[j], tring[j]);
    as_keys(ring[j]);
#pragma unroll
    for (int g = 0; g < 8; ++g) { sad[g].h[0] += ring[j][g].h[0]; sad[g].h[1] += ring[j][g].h[1]; }
    tsum += tring[j];
  }
  RowWin wa = win(y0 + WSZ2);
  for (int yb = y0; yb < y1; yb += NR) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {      // step k: row y+3 enters into the free slot (k + 6) % 7, row y-3 leaves from slot k, which is free then
      const int y = yb + k;
      if (y < y1) {      // (uniform)
        const int in = (k + NR - 1) % NR;
        row_sads(wa, ft4, ring[in], tring[in]);
        as_keys(ring[in]);
        wa = win(y + WSZ2 + 1);      // next step's row: issued as soon as this step's is consumed, in flight during the selection
#pragma unroll
        for (int g = 0; g < 8; ++g) { sad[g].h[0] += ring[in][g].h[0]; sad[g].h[1] += ring[in][g].h[1]; }
        tsum += tring[in];
        int16_t d16; uint16_t c16;
        bm_select<SMALL>(sad, tsum, S, s_scr[lane], d16, c16);
        if (store) {
          const size_t o = ((size_t)b * h + y) * w + x + (NDISP - 1);
          S.disp16[o] = d16; S.cost[o] = c16;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) { sad[g].h[0] -= ring[k][g].h[0]; sad[g].h[1] -= ring[k][g].h[1]; }
        tsum -= tring[k];
      }
    }
  }
}

// columns X < 31 (never searched) and x in [0,3), whose right-image window clamps at column 0 BEFORE the disparity
// shift (so it is not a contiguous byte window).  Half a wave per pixel: lane = disparity index d.
// grid: (ceil(h/EDGE_ROWS), batch), block 256
constexpr int EDGE_THREADS = 256, EDGE_ROWS = 64, EDGE_REC = 13;      // rows per workgroup; dwords staged per image row
__global__ __launch_bounds__(EDGE_THREADS) void stereo_bm_edge_kernel(StereoDev S) {
  // what the three columns of EDGE_ROWS rows read: right bytes 0..39 and left bytes 28..39 of the rows ybase-3 .. ybase+EDGE_ROWS+2 -- 52 bytes per row, staged once
  // with aligned dword loads (the per-lane form issued 28 unaligned dword loads per pixel and disparity: 10 M wave-level loads per 512 frames, the whole kernel time)
  __shared__ uint32_t s_rec[(EDGE_ROWS + 2 * WSZ2) * EDGE_REC];
  const int b = blockIdx.y, w = S.w, h = S.h, tid = threadIdx.x, lane = tid & 63, d = lane & 31;
  const int width1 = w - NDISP + 1, ncol = min(3, width1);
  const int ybase = blockIdx.x * EDGE_ROWS, nrow = min(EDGE_ROWS, h - ybase);
  const uint8_t *lp = S.lp + (size_t)b * h * S.pitch + PADL, *rp = S.rp + (size_t)b * h * S.pitch + PADL;      // (PADL and the pitch are multiples of 4)
  for (int i = tid; i < (nrow + 2 * WSZ2) * EDGE_REC; i += EDGE_THREADS) {
    const int r = i / EDGE_REC, c = i - r * EDGE_REC, yy = min(max(ybase - WSZ2 + r, 0), h - 1);
    s_rec[i] = *reinterpret_cast<const uint32_t *>((c < 10 ? rp + 4 * c : lp + 28 + 4 * (c - 10)) + (size_t)yy * S.pitch);
  }
  {   // the never-searched left border, FILTERED
    const int n = (NDISP - 1) * nrow;
    for (int i = tid; i < n; i += EDGE_THREADS) {
      const size_t o = ((size_t)b * h + ybase + i / (NDISP - 1)) * w + i % (NDISP - 1);
      S.disp16[o] = (int16_t)FILTERED16; S.cost[o] = 0;
    }
  }
  __syncthreads();
  const uint32_t ft4 = 0x01010101u * (uint32_t)(S.cap + 1);
  const int q = d >> 2, sh = d & 3;
  for (int pix = tid >> 5; pix < ncol * nrow; pix += EDGE_THREADS / 32) {      // half a wave per pixel: lane = disparity index d
    const int yl = pix / ncol, x = pix - yl * ncol, y = ybase + yl;
    // A window row: left bytes x+28 .. x+34 (8th byte masked); right bytes max(x + dx, 0) + d for dx = -3..3 -- the clamp repeats column d, so the seven taps are a
    // byte shuffle (V_PERM) of the bytes d .. d+7 (x <= 2, d <= 31: index <= 36 < w, no upper clamp)
    const uint32_t selA = x == 0 ? 0x00000000u : x == 1 ? 0x01000000u : 0x02010000u;      // taps dx = -3..0
    const uint32_t selB = x == 0 ? 0x0c030201u : x == 1 ? 0x0c040302u : 0x0c050403u;      // taps dx = 1..3, then a zero byte
    uint32_t sad_u = 0, tsum_u = 0;
#pragma unroll
    for (int k = 0; k <= 2 * WSZ2; ++k) {
      const uint32_t *rec = s_rec + (yl + k) * EDGE_REC;
      const uint32_t w0 = rec[q], w1 = rec[q + 1], w2 = rec[q + 2], a0 = rec[10], a1 = rec[11], a2 = rec[12];
      const uint32_t r0 = __builtin_amdgcn_alignbyte(w1, w0, sh), r1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
      const uint32_t l0 = __builtin_amdgcn_alignbyte(a1, a0, x), l1 = __builtin_amdgcn_alignbyte(a2, a1, x) & 0x00ffffffu;
      const uint32_t ra = __builtin_amdgcn_perm(r1, r0, selA), rb = __builtin_amdgcn_perm(r1, r0, selB);
      sad_u = __builtin_amdgcn_sad_u8(l1, rb, __builtin_amdgcn_sad_u8(l0, ra, sad_u));
      tsum_u = __builtin_amdgcn_sad_u8(l1 | (ft4 & 0xff000000u), ft4, __builtin_amdgcn_sad_u8(l0, ft4, tsum_u));
    }
    const int sad = (int)sad_u, tsum = (int)tsum_u;
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
    if (d == 0) {
      int16_t d16 = (int16_t)FILTERED16; uint16_t c16 = 0;
      if (tsum >= S.texthr && !(S.uniq > 0 && rival)) {
        const int dd = p + n - 2 * minsad + abs(p - n);
        d16 = (int16_t)(((NDISP - mind - 1) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4);
        c16 = (uint16_t)minsad;
      }
      const size_t o = ((size_t)b * h + y) * w + x + (NDISP - 1);
      S.disp16[o] = d16; S.cost[o] = c16;
    }
  }
}

// validateDisparity (with D2): one WAVE per image row, four rows per workgroup (a row is 640 pixels: a 256-lane workgroup per row spent
// its time in dispatch and two workgroup barriers; a wave orders its own LDS traffic without them).  The reference's sequential first
// pass ("a strictly smaller cost wins", ascending x => first x on ties) is a minimum over (cost, x): every valid source pixel does one
// LDS atomicMin of the packed key (cost << 16 | x) on its target column.  grid: (ceil(h/R), batch), block 64 R, dynamic LDS = R * 2 * w ints
// (R = 4 rows per workgroup; 1 for rows wider than 2048 pixels)
__global__ __launch_bounds__(256) void stereo_validate_kernel(StereoDev S) {
  extern __shared__ int s_mem[];
  const int w = S.w, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, y = blockIdx.x * (blockDim.x >> 6) + wave, b = blockIdx.y;
  if (y >= S.h) return;                                     // wave-uniform; no workgroup barrier below
  int *s_d = s_mem + wave * 2 * w;
  unsigned *s_key = reinterpret_cast<unsigned *>(s_d + w);
  const int SCALE = 1 << DISP_SHIFT, INVALID = -SCALE, maxdiff = S.disp12 * SCALE;
  int16_t *dp = S.disp16 + ((size_t)b * S.h + y) * w;
  const uint16_t *cp = S.cost + ((size_t)b * S.h + y) * w;
  // the row into LDS, one word per pixel: cost << 16 | disparity (16 bits each).  Rows of 4 n pixels: four pixels per lane and load (8 bytes of
  // each array; 2-byte loads made the wave wait on twenty narrow loads per row)
  if ((w & 3) == 0) {
    for (int x = 4 * lane; x < w; x += 256) {
      const uint2 dv = *reinterpret_cast<const uint2 *>(dp + x), cv = *reinterpret_cast<const uint2 *>(cp + x);
      int4 e;
      e.x = (int)((dv.x & 0xffffu) | (cv.x << 16)); e.y = (int)((dv.x >> 16) | (cv.x & 0xffff0000u));
      e.z = (int)((dv.y & 0xffffu) | (cv.y << 16)); e.w = (int)((dv.y >> 16) | (cv.y & 0xffff0000u));
      *reinterpret_cast<int4 *>(s_d + x) = e;
      *reinterpret_cast<uint4 *>(s_key + x) = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    }
  } else {
    for (int x = lane; x < w; x += 64) { s_d[x] = (int)(((unsigned)(uint16_t)dp[x]) | ((unsigned)cp[x] << 16)); s_key[x] = 0xffffffffu; }
  }
  wave_sync_lds();
  const int minX1 = NDISP;
  for (int x = minX1 + lane; x < w; x += 64) {
    const int e = s_d[x], d = (int16_t)(e & 0xffff);
    if (d == INVALID) continue;
    const int x2 = x - ((d + SCALE / 2) >> DISP_SHIFT);
    if (x2 >= 0 && x2 < w) atomicMin(&s_key[x2], ((unsigned)e & 0xffff0000u) | (unsigned)x);
  }
  wave_sync_lds();
  for (int x = minX1 + lane; x < w; x += 64) {
    const int d = (int16_t)(s_d[x] & 0xffff);
    if (d == INVALID) continue;
    const int x0 = x - (d >> DISP_SHIFT), x1 = x - ((d + SCALE - 1) >> DISP_SHIFT);
    bool bad0 = false, bad1 = false;
    if (x0 >= 0 && x0 < w) { const unsigned k = s_key[x0]; bad0 = k != 0xffffffffu && abs((int16_t)(s_d[k & 0xffffu] & 0xffff) - d) > maxdiff; }
    if (x1 >= 0 && x1 < w) { const unsigned k = s_key[x1]; bad1 = k != 0xffffffffu && abs((int16_t)(s_d[k & 0xffffu] & 0xffff) - d) > maxdiff; }
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
// bounded variants for the strip path (nodes of undecided components only)
__device__ __forceinline__ int ccl_find_b(const int32_t *label, int x, int *err) {
  int p = label[x], steps = 0;
  while (p != x && p >= 0) { x = p; p = label[x]; if (++steps > (1 << 16)) { atomicOr(err, 8); return CCL_BIG; } }
  return p < 0 ? CCL_BIG : x;
}
__device__ __forceinline__ void ccl_union_b(int32_t *label, int a, int b, int *err) {
  for (int tries = 0; tries < (1 << 16); ++tries) {
    if (a >= 0) a = ccl_find_b(label, a, err);
    if (b >= 0) b = ccl_find_b(label, b, err);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&label[a], b);
    if (old == a) return;
    a = old;
  }
  atomicOr(err, 16);
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


// ---- speckle filter by strips ------------------------------------------------------------------------------------------
// filterSpeckles needs the SIZE of every 4-connected component, but only up to speckle_window (100 pixels in the reference's
// configuration): nearly every component is either far larger or a speck of a few pixels, and both kinds are decided inside a strip
// of rows held in LDS.  One workgroup per (strip, frame):
//   runs      one wave per row: a pixel's label is the strip index of the first pixel of its horizontal run (two ballots per 64-pixel
//             segment, the last run start carried across segments in an SGPR); the run's last pixel knows its length -- runs longer
//             than the window make their start a member of BIG at once;
//   stitch    vertically linked pixels unite their runs (lock-free union-find on LDS words: a root holds ~(pixels | touch << 24), i.e.
//             a negative word, every other node the index of its parent; the larger root index is hung under the smaller with a CAS);
//   count     the last pixel of every run adds its length to the root (saturating: the test is size <= window) and marks the root if
//             the run lies on a row that has a neighbour strip;
//   decide    more than window pixels, or BIG: kept.  Small and not touching a neighbour strip: filtered.  Small and touching: UNDECIDED
//             -- the pixels are kept for now and listed (pixel, node) with node = the frame index of the root pixel; the first / last
//             row of the strip leaves its node ids in `brow`.  The float conversion of the stage is fused in here.
// Three small kernels finish the undecided ones: unite nodes across strip boundaries (global union-find, the same routines as the
// whole-frame path below), count the listed pixels per united root, filter the pixels of the roots that stay <= window.
// HBM traffic of the whole filter: disparity read once (2 B/px), float written once (4 B/px); the whole-frame path moved 46 B/px.
constexpr int SPK_THREADS = 1024;
constexpr int SPK_NS = 10;                         // segments of a row handled in registers at once
constexpr int SPK_MAX_SEG = 40;                   // 64-pixel segments per row: w <= 2560
constexpr int SPK_BIG = (int)0x80000000;           // root value of the BIG set; no real root value is that negative
constexpr int SPK_TOUCH = 1 << 24;
// LDS bytes per strip row: label (int) + disparity (int16) per pixel, and for rows of up to SPK_NS segments the four pass masks per segment
__host__ __device__ constexpr int spk_row_bytes(int w) { return ((w + 3) & ~3) * 6 + ((w + 63) / 64 <= SPK_NS ? 4 * SPK_NS * 8 : 0); }
// (every loop of the union-find routines is bounded: a parent index is smaller than its child, so a walk takes fewer steps than the strip
//  has pixels; past that something is broken and the walk gives up -- the library never hangs the GPU -- leaving a mark in *err)
constexpr int SPK_MAX_STEPS = 1 << 16;
__device__ __forceinline__ int spk_find(const int *lab, int x, int *err) {      // -> root index, or -1 for BIG
  int v = lab[x], steps = 0;
  while (v >= 0) { x = v; v = lab[x]; if (++steps > SPK_MAX_STEPS) { atomicOr(err, 1); return -1; } }
  return v == SPK_BIG ? -1 : x;
}
// same, halving the path on the way (a parent word is only ever replaced by an ancestor: safe next to concurrent unions and finds)
__device__ __forceinline__ int spk_find_halve(int *lab, int x, int *err) {
  int v = lab[x], steps = 0;
  while (v >= 0) {
    const int vv = lab[v];
    if (vv >= 0) lab[x] = vv;
    x = v; v = vv;
    if (++steps > SPK_MAX_STEPS) { atomicOr(err, 2); return -1; }
  }
  return v == SPK_BIG ? -1 : x;
}
__device__ __forceinline__ void spk_union(int *lab, int a, int b, int *err) {   // a, b: nodes of the strip
  for (int tries = 0;; ++tries) {
    if (tries > SPK_MAX_STEPS) { atomicOr(err, 4); return; }
    if (a >= 0) a = spk_find_halve(lab, a, err);              // (-1 = BIG stays BIG: it is not an index)
    if (b >= 0) b = spk_find_halve(lab, b, err);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }                     // a > b >= -1: hang root a under b (b == -1: a joins BIG)
    if (atomicCAS(&lab[a], ~0, b < 0 ? SPK_BIG : b) == ~0) return;    // (a root before the counting pass is ~0: zero pixels, no mark)
    // a stopped being a root in the meantime: look again from the top
  }
}
// grid: (n_strips, batch), block SPK_THREADS, dynamic LDS = strip_rows * round_up(w, 4) * 6 bytes (labels: int, disparities: int16).
// Four horizontally adjacent pixels per lane and step (one 8-byte disparity read, one 16-byte label read); a pixel of a run longer than
// the window carries SPK_BIG itself, so the common cases -- filtered, or big above big -- leave after those two reads.
__device__ __forceinline__ void spk_unpack4(uint2 v, int (&d)[4]) {
  d[0] = (int16_t)(v.x & 0xffff); d[1] = (int16_t)(v.x >> 16); d[2] = (int16_t)(v.y & 0xffff); d[3] = (int16_t)(v.y >> 16);
}
// acc with lane `sel` replaced by the wave-uniform `val` (V_WRITELANE_B32; this compiler has no builtin for it).  `sel` is a constant after unrolling at every call
// site: the switch folds to the one instruction with the lane select as an inline constant.
__device__ __forceinline__ unsigned spk_writelane(unsigned acc, unsigned val, int sel) {
#define SPK_WL(n) case n: asm("v_writelane_b32 %0, %1, " #n : "+v"(acc) : "s"(val)); break;
  switch (sel) {
    SPK_WL(0) SPK_WL(1) SPK_WL(2) SPK_WL(3) SPK_WL(4) SPK_WL(5) SPK_WL(6) SPK_WL(7) SPK_WL(8) SPK_WL(9)
    SPK_WL(10) SPK_WL(11) SPK_WL(12) SPK_WL(13) SPK_WL(14) SPK_WL(15) SPK_WL(16) SPK_WL(17) SPK_WL(18) SPK_WL(19)
    SPK_WL(20) SPK_WL(21) SPK_WL(22) SPK_WL(23) SPK_WL(24) SPK_WL(25) SPK_WL(26) SPK_WL(27) SPK_WL(28) SPK_WL(29)
    SPK_WL(30) SPK_WL(31) SPK_WL(32) SPK_WL(33) SPK_WL(34) SPK_WL(35) SPK_WL(36) SPK_WL(37) SPK_WL(38) SPK_WL(39)
    default: break;
  }
#undef SPK_WL
  return acc;
}
__device__ __forceinline__ unsigned long long spk_readlane64(unsigned long long v, int l) {      // value of lane l (l wave-uniform)
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
__global__ __launch_bounds__(SPK_THREADS) void stereo_speckle_strip_kernel(StereoDev S, float *__restrict__ out, int dstride, size_t d_bstride) {
  extern __shared__ int s_mem[];
  __shared__ unsigned long long s_bb[SPK_THREADS / 64][SPK_MAX_SEG];
  __shared__ int s_npend;
  if (threadIdx.x == 0) s_npend = 0;
  const int w = S.w, h = S.h, wp = (w + 3) & ~3, gpr = wp >> 2;
  const int b = blockIdx.y, strip = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y0 = strip * S.strip_rows, y1 = min(h, y0 + S.strip_rows), nr = y1 - y0, ng = nr * gpr;
#define SPK_STAMP(k) do { if (S.timing && tid == 0 && strip == S.n_strips / 2 && b == 0) reinterpret_cast<long long *>(S.err + 8)[k] = (long long)wall_clock64(); } while (0)
  SPK_STAMP(0);
  int *lab = s_mem;
  int16_t *dsp = reinterpret_cast<int16_t *>(s_mem + S.strip_rows * wp);
  const size_t fbase = (size_t)b * w * h, sbase = fbase + (size_t)y0 * w;
  const int range = S.speckle_range, window = S.speckle_window;
  const bool vec = (w & 3) == 0;                               // then every row of the frame starts 8-byte aligned
  const uint32_t f2 = (uint32_t)(uint16_t)FILTERED16 * 0x00010001u;
  for (int g = tid; g < ng; g += SPK_THREADS) {
    const int row = g / gpr, x0 = (g - row * gpr) * 4;
    const int16_t *src = S.disp16 + sbase + (size_t)row * w + x0;
    uint2 v;
    if (vec) v = *reinterpret_cast<const uint2 *>(src);
    else {
      uint32_t e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = x0 + k < w ? (uint32_t)(uint16_t)src[k] : (uint32_t)(uint16_t)FILTERED16;
      v.x = e[0] | (e[1] << 16); v.y = e[2] | (e[3] << 16);
    }
    *reinterpret_cast<uint2 *>(dsp + row * wp + x0) = v;
  }
  __syncthreads();
  SPK_STAMP(1);
  // ---- runs: one wave per row.  Left to right: the start of every pixel's run; right to left: its end, hence its length.
  //      Rows of up to SPK_NS segments (640 pixels) are handled in registers: all the row's loads in flight at once, the ballots of its
  //      segments in scalar registers, one label store per pixel -- the stepwise form below paid three LDS round trips per segment.
  const int nseg = (w + 63) >> 6;
  // Rows of up to SPK_NS segments also leave four bit masks per segment in LDS (`msk`, [row][4][SPK_NS] ballots) that the three later passes scan instead of the
  // pixels: ST = "unites its run with the one below" (vertically linked and the leftmost linked column of the two runs' overlap), CN = "last pixel of a run that
  // is not big", CL = "valid pixel of a run that is not big", S = "starts a run".  Their loops then cost two V_READLANE and a scalar branch per 64 pixels
  // where they read four disparities and evaluated four link tests per pixel (those passes were bound by VALU issue on the strip's one CU).  "Big" is the
  // state after THIS pass (a single-pixel run that joins BIG while rows are stitched stays listed: its find returns BIG and the entry is dropped).
  unsigned long long *const msk = reinterpret_cast<unsigned long long *>(dsp + S.strip_rows * wp);
  constexpr int MSK_ROW = 4 * SPK_NS;
  const bool use_msk = nseg <= SPK_NS;
  if (nseg <= SPK_NS) {
    for (int row = wave; row < nr; row += SPK_THREADS / 64) {
      const int16_t *d = dsp + row * wp;
      int *l = lab + row * wp;
      const bool has_b = row + 1 < nr;
      int dv[SPK_NS], dl[SPK_NS], db[SPK_NS], dbl[SPK_NS];
#pragma unroll
      for (int sg = 0; sg < SPK_NS; ++sg) {
        const int x = sg * 64 + lane;
        dv[sg] = x < w ? d[x] : FILTERED16;
        dl[sg] = (x < w && x > 0) ? d[x - 1] : FILTERED16;
        db[sg] = (has_b && x < w) ? d[wp + x] : FILTERED16;
        dbl[sg] = (has_b && x < w && x > 0) ? d[wp + x - 1] : FILTERED16;
      }
      // the row's 4 x SPK_NS mask words are assembled in the lanes that will store them (lane = mask * SPK_NS + segment): V_WRITELANE from the scalar ballots
      unsigned mlo = 0u, mhi = 0u;
      auto put = [&](int word, unsigned long long v) { mlo = spk_writelane(mlo, (unsigned)v, word); mhi = spk_writelane(mhi, (unsigned)(v >> 32), word); };
      unsigned long long sb[SPK_NS], bb[SPK_NS];
      unsigned long long vm_prev = 0ull;
#pragma unroll
      for (int sg = 0; sg < SPK_NS; ++sg) {
        const bool filt = dv[sg] == FILTERED16, start = !filt && !ccl_linked(dv[sg], dl[sg], range);
        sb[sg] = __ballot(start); bb[sg] = __ballot(start || filt);
        const unsigned long long vm = __ballot(ccl_linked(dv[sg], db[sg], range));        // vertically linked to the pixel below
        const unsigned long long hb = __ballot(ccl_linked(db[sg], dbl[sg], range));       // the pixel below is linked to its left neighbour
        const unsigned long long vsh = (vm << 1) | (vm_prev >> 63);                        // the left neighbours are vertically linked
        put(0 * SPK_NS + sg, vm & ~(vsh & ~bb[sg] & hb));                                  // (~bb: valid and linked to the left neighbour; 0 past the row end)
        put(3 * SPK_NS + sg, sb[sg]);
        vm_prev = vm;
      }
      int carry[SPK_NS], nbc[SPK_NS];                         // last run start left of segment sg / first boundary right of it
      { int c = 0; 
#pragma unroll
        for (int sg = 0; sg < SPK_NS; ++sg) { carry[sg] = c; if (sb[sg]) c = sg * 64 + 63 - __clzll((long long)sb[sg]); } }
      { int c = w;
#pragma unroll
        for (int sg = SPK_NS - 1; sg >= 0; --sg) { nbc[sg] = c; if (bb[sg]) c = sg * 64 + __ffsll((long long)bb[sg]) - 1; } }
#pragma unroll
      for (int sg = 0; sg < SPK_NS; ++sg) {
        const int x = sg * 64 + lane;
        const unsigned long long lower = sb[sg] & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        const unsigned long long above = lane == 63 ? 0ull : (bb[sg] >> (lane + 1));
        const int st = lower ? sg * 64 + 63 - __clzll((long long)lower) : carry[sg];
        const int nbx = above ? x + __ffsll((long long)above) : nbc[sg];
        const bool filt = dv[sg] == FILTERED16, isbig = !filt && nbx - st > window;
        const unsigned long long big = __ballot(isbig && x < w);
        if (x < w) l[x] = filt ? SPK_BIG + 1 : isbig ? SPK_BIG : st == x ? ~0 : row * wp + st;
        const unsigned long long cl = (sb[sg] | ~bb[sg]) & ~big;                                                   // valid (starts a run or is linked to the left) and not big
        const unsigned long long hnext = (~bb[sg] >> 1) | (sg + 1 < SPK_NS ? ~bb[sg + 1 < SPK_NS ? sg + 1 : sg] << 63 : 0ull);      // pixel x + 1 is linked to x
        put(1 * SPK_NS + sg, cl & ~hnext);
        put(2 * SPK_NS + sg, cl);
      }
      for (int x = w + lane; x < wp; x += 64) l[x] = SPK_BIG + 1;
      if (lane < MSK_ROW) msk[(size_t)row * MSK_ROW + lane] = ((unsigned long long)mhi << 32) | mlo;
    }
  } else {
    for (int row = wave; row < nr; row += SPK_THREADS / 64) {
      const int16_t *d = dsp + row * wp;
      int *l = lab + row * wp;
      int carry = 0;                                            // last run start of the earlier segments
      for (int seg = 0; seg < nseg; ++seg) {
        const int x = seg * 64 + lane;
        const bool in = x < w;
        const int dv = in ? d[x] : FILTERED16, dl = (in && x > 0) ? d[x - 1] : FILTERED16;
        const bool filt = dv == FILTERED16, start = !filt && !ccl_linked(dv, dl, range);
        const unsigned long long sb = __ballot(start), bb = __ballot(start || filt);
        const unsigned long long lower = sb & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        if (in) l[x] = filt ? -1 : lower ? seg * 64 + 63 - __clzll((long long)lower) : carry;
        if (lane == 0) s_bb[wave][seg] = bb;
        if (sb) carry = seg * 64 + 63 - __clzll((long long)sb);
      }
      int nb_carry = w;                                         // first boundary (run start, filtered pixel, row end) of the later segments
      for (int seg = nseg - 1; seg >= 0; --seg) {
        const int x = seg * 64 + lane;
        const unsigned long long bb = s_bb[wave][seg];
        const unsigned long long above = lane == 63 ? 0ull : (bb >> (lane + 1));
        const int nbx = above ? x + __ffsll((long long)above) : nb_carry;
        if (x < w) {
          const int st = l[x];
          l[x] = st < 0 ? SPK_BIG + 1 : (nbx - st > window) ? SPK_BIG : st == x ? ~0 : row * wp + st;
        }
        if (bb) nb_carry = seg * 64 + __ffsll((long long)bb) - 1;
      }
      for (int x = w + lane; x < wp; x += 64) l[x] = SPK_BIG + 1;
    }
  }
  __syncthreads();
  SPK_STAMP(2);
  // ---- stitch / count / classify share one shape: a wave takes a row, lane = column, all the row's operands in flight at once; the few
  //      lanes with real work (a union, a run to count, a pixel of a small component) do not do it in place -- every such block would
  //      cost the whole wave three dependent LDS round trips per segment -- but append their column to a queue, and the queue is worked
  //      off with one entry per lane.
  int *const q = reinterpret_cast<int *>(s_bb[wave]);        // 64 entries (the ballots of the stepwise runs pass are done with)
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  // queue entry: column (12 bits) | "starts its run" << 12 | row << 16.  The queue is worked off when it is full and once after the wave's last row: a flush is a
  // chain of dependent LDS round trips whatever the number of entries, so it is not paid per row.
  int qn = 0;
  auto enqueue = [&](unsigned long long m, unsigned long long flag_m, int x0, int row, auto &&flush) {      // the set lanes of m append column x0 + lane
    if (m == 0) return;
    const int c = __popcll(m);
    if (qn + c > 64) { flush(qn); qn = 0; }
    if ((m >> lane) & 1ull) q[qn + __popcll(m & lt_mask)] = (x0 + lane) | ((int)((flag_m >> lane) & 1ull) << 12) | (row << 16);
    qn += c;
  };
  auto row_masks = [&](int row) { return lane < MSK_ROW ? msk[(size_t)row * MSK_ROW + lane] : 0ull; };      // the row's masks: one LDS read, then lane broadcasts
  // ---- stitch rows.  A pixel unites its run with the one below where it is the LEFTMOST linked column of the two runs' overlap:
  //      vertically linked, and not (left neighbours vertically linked + both horizontal links)
  {
    auto flush = [&](int n_) {
      if (lane < n_) {
        const int e = q[lane], x = e & 0xfff, row = e >> 16;
        const int a = lab[row * wp + x], c = lab[(row + 1) * wp + x];
        // node of a pixel = its run start (the pixel itself where it starts a run or carries SPK_BIG: its word is a root / parent word then)
        if (!(a == SPK_BIG && c == SPK_BIG)) spk_union(lab, a >= 0 ? a : row * wp + x, c >= 0 ? c : (row + 1) * wp + x, S.err);
      }
    };
    for (int row = wave; row < nr - 1; row += SPK_THREADS / 64) {
      if (use_msk) {
        const unsigned long long mv = row_masks(row);
#pragma unroll
        for (int k = 0; k < SPK_NS; ++k) enqueue(spk_readlane64(mv, 0 * SPK_NS + k), 0ull, k * 64, row, flush);
      } else {
        const int16_t *da = dsp + row * wp, *db = da + wp;
        for (int seg = 0; seg < nseg; ++seg) {
          const int x = seg * 64 + lane;
          const bool in = x < w;
          const int va = in ? da[x] : FILTERED16, vb = in ? db[x] : FILTERED16;
          const int pa = (in && x > 0) ? da[x - 1] : FILTERED16, pb = (in && x > 0) ? db[x - 1] : FILTERED16;
          const bool lk = ccl_linked(va, vb, range);
          const bool covered = ccl_linked(pa, pb, range) && ccl_linked(pa, va, range) && ccl_linked(pb, vb, range);
          enqueue(__ballot(lk && !covered), 0ull, seg * 64, row, flush);
        }
      }
    }
    flush(qn); qn = 0;
  }
  __syncthreads();
  SPK_STAMP(3);
  // ---- count: the last pixel of a run adds the run to its root
  const bool nb_top = y0 > 0, nb_bot = y1 < h;
  {
    auto flush = [&](int n_) {
      if (lane < n_) {
        const int e = q[lane], x = e & 0xfff, row = e >> 16;
        const int st = ((e >> 12) & 1) ? row * wp + x : lab[row * wp + x];      // (the word of a start pixel is a root / parent word, not a start index)
        const int root = spk_find(lab, st, S.err);
        if (root >= 0) {
          if ((~lab[root] & (SPK_TOUCH - 1)) <= window) atomicSub(&lab[root], row * wp + x - st + 1);      // ~(V + len) = ~V - len
          if ((row == 0 && nb_top) || (row == nr - 1 && nb_bot)) atomicAnd(&lab[root], ~SPK_TOUCH);            // sets the mark in V
        }
      }
    };
    for (int row = wave; row < nr; row += SPK_THREADS / 64) {
      if (use_msk) {
        const unsigned long long mv = row_masks(row);
#pragma unroll
        for (int k = 0; k < SPK_NS; ++k) enqueue(spk_readlane64(mv, 1 * SPK_NS + k), spk_readlane64(mv, 3 * SPK_NS + k), k * 64, row, flush);
      } else {
        const int16_t *d = dsp + row * wp;
        const int *l = lab + row * wp;
        for (int seg = 0; seg < nseg; ++seg) {
          const int x = seg * 64 + lane;
          const bool in = x < w;
          const int dv = in ? d[x] : FILTERED16, lv = in ? l[x] : SPK_BIG + 1;
          const int dl = (in && x > 0) ? d[x - 1] : FILTERED16, dr = (in && x + 1 < w) ? d[x + 1] : FILTERED16;
          const bool on = dv != FILTERED16 && lv != SPK_BIG && !ccl_linked(dv, dr, range);      // last pixel of a run that is not big
          enqueue(__ballot(on), __ballot(!ccl_linked(dv, dl, range)), seg * 64, row, flush);
        }
      }
    }
    flush(qn); qn = 0;
  }
  __syncthreads();
  SPK_STAMP(4);
  // ---- classify the pixels of the components that are not big: filtered (their disparity in LDS becomes FILTERED), kept, or undecided
  //      (kept for now and listed; on the strip's first / last row they publish their node)
  __shared__ unsigned s_pmask[2][SPK_MAX_SEG * 2];            // undecided pixels of the first / last row
  for (int i = tid; i < 2 * SPK_MAX_SEG * 2; i += SPK_THREADS) (&s_pmask[0][0])[i] = 0u;
  __syncthreads();
  {
    auto flush = [&](int n_) {
      if (lane < n_) {
        const int e = q[lane], x = e & 0xfff, row = e >> 16, idx = row * wp + x;
        const int root = spk_find(lab, ((e >> 12) & 1) ? idx : lab[idx], S.err);
        if (root >= 0) {                                      // else: joined BIG
          const int V = ~lab[root];
          if ((V & (SPK_TOUCH - 1)) <= window) {
            if (!(V & SPK_TOUCH)) dsp[idx] = (int16_t)FILTERED16;
            else {                                            // undecided: decided after the strips have been united
              const int rrow = root / wp, node = (y0 + rrow) * w + (root - rrow * wp);      // frame index of the root pixel
              if (root == idx) { S.label[fbase + node] = node; S.count[fbase + node] = 0; }
              const int slot = atomicAdd(&s_npend, 1);        // (a per-frame counter in global memory cost more than the rest of the kernel)
              S.pend[sbase + slot] = make_int2((y0 + row) * w + x, node);
              for (int side = 0; side < 2; ++side)
                if (side == 0 ? (row == 0 && nb_top) : (row == nr - 1 && nb_bot)) {
                  S.brow[((size_t)b * 2 * S.n_strips + 2 * strip + side) * w + x] = node;
                  atomicOr(&s_pmask[side][x >> 5], 1u << (x & 31));
                }
            }
          }
        }
      }
    };
    for (int row = wave; row < nr; row += SPK_THREADS / 64) {
      if (use_msk) {
        const unsigned long long mv = row_masks(row);
#pragma unroll
        for (int k = 0; k < SPK_NS; ++k) enqueue(spk_readlane64(mv, 2 * SPK_NS + k), spk_readlane64(mv, 3 * SPK_NS + k), k * 64, row, flush);
      } else {
        const int16_t *d = dsp + row * wp;
        const int *l = lab + row * wp;
        for (int seg = 0; seg < nseg; ++seg) {
          const int x = seg * 64 + lane;
          const bool in = x < w;
          const int dv = in ? d[x] : FILTERED16, lv = in ? l[x] : SPK_BIG + 1, dl = (in && x > 0) ? d[x - 1] : FILTERED16;
          enqueue(__ballot(dv != FILTERED16 && lv != SPK_BIG), __ballot(!ccl_linked(dv, dl, range)), seg * 64, row, flush);
        }
      }
    }
    flush(qn); qn = 0;
  }
  __syncthreads();
  SPK_STAMP(5);
  // ---- convert: the float disparity of the stage (four pixels per lane); the strip's first / last row publish -1 (filtered) / -2 (kept)
  //      where no undecided node stands
  const float sc = 1.f / (1 << DISP_SHIFT);
  for (int g = tid; g < ng; g += SPK_THREADS) {
    const int i0 = g * 4, row = g / gpr, x0 = (g - row * gpr) * 4;
    int dq[4];
    spk_unpack4(*reinterpret_cast<const uint2 *>(dsp + i0), dq);
    float *o = out + (size_t)b * d_bstride + (size_t)(y0 + row) * dstride + x0;
    if (vec && (dstride & 3) == 0 && (d_bstride & 3) == 0) *reinterpret_cast<float4 *>(o) = make_float4((float)dq[0] * sc, (float)dq[1] * sc, (float)dq[2] * sc, (float)dq[3] * sc);
    else {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (x0 + k < w) o[k] = (float)dq[k] * sc;
    }
#pragma unroll
    for (int side = 0; side < 2; ++side)
      if (side == 0 ? (row == 0 && nb_top) : (row == nr - 1 && nb_bot)) {
        int32_t *br = S.brow + ((size_t)b * 2 * S.n_strips + 2 * strip + side) * w + x0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (x0 + k < w && !((s_pmask[side][(x0 + k) >> 5] >> ((x0 + k) & 31)) & 1u)) br[k] = dq[k] == FILTERED16 ? -1 : -2;
      }
  }
  __syncthreads();
  SPK_STAMP(6);
  if (tid == 0) S.npend[b * S.n_strips + strip] = s_npend;
}
// unite the undecided components across strip boundaries.  grid: (n_strips - 1, batch), block 256
__global__ __launch_bounds__(256) void stereo_speckle_merge_kernel(StereoDev S) {
  const int w = S.w, b = blockIdx.y, k = blockIdx.x;            // boundary between strip k and k + 1
  const int yb = (k + 1) * S.strip_rows - 1;                    // last row of strip k
  const size_t fbase = (size_t)b * w * S.h;
  const int32_t *na = S.brow + ((size_t)b * 2 * S.n_strips + 2 * k + 1) * w, *nb = S.brow + ((size_t)b * 2 * S.n_strips + 2 * k + 2) * w;
  const int16_t *da = S.disp16 + fbase + (size_t)yb * w, *db = da + w;
  int32_t *label = S.label + fbase;
  for (int x = threadIdx.x; x < w; x += 256) {
    const int a = na[x], c = nb[x];
    if (a == -1 || c == -1 || (a == -2 && c == -2)) continue;
    if (!ccl_linked(da[x], db[x], S.speckle_range)) continue;
    ccl_union_b(label, a == -2 ? CCL_BIG : a, c == -2 ? CCL_BIG : c, S.err);
  }
}
// pass 0: pixels per united root (saturating); pass 1: filter the pixels of the roots that stay small.  grid: (n_strips, batch), block 256
__global__ __launch_bounds__(256) void stereo_speckle_resolve_kernel(StereoDev S, int pass, float *__restrict__ out, int dstride, size_t d_bstride) {
  const int b = blockIdx.y, np = S.npend[b * S.n_strips + blockIdx.x], w = S.w;
  const size_t fbase = (size_t)b * w * S.h, sbase = fbase + (size_t)blockIdx.x * S.strip_rows * w;
  for (int i = threadIdx.x; i < np; i += 256) {
    const int2 e = S.pend[sbase + i];
    const int root = ccl_find_b(S.label + fbase, e.y, S.err);
    if (root < 0) continue;
    if (pass == 0) { if (S.count[fbase + root] <= S.speckle_window) atomicAdd(&S.count[fbase + root], 1); }
    else if (S.count[fbase + root] <= S.speckle_window) out[(size_t)b * d_bstride + (size_t)(e.x / w) * dstride + (e.x % w)] = (float)FILTERED16 * (1.f / (1 << DISP_SHIFT));
  }
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
  int strip_rows = 0, n_strips = 0;                     // strip speckle filter (0: whole-frame path)
  int32_t *d_brow = nullptr, *d_npend = nullptr;
  int2 *d_pend = nullptr;
  int debug = 0;                                      // SVS_STEREO_DEBUG=1 at create
  int force_frame_ccl = 0;                            // SVS_STEREO_FRAME_CCL=1 at create: the whole-frame union-find path (tests compare the two)
  int force_prefilter4 = 0;                           // SVS_STEREO_PREFILTER4=1 at create: the generic 4-pixel prefilter kernel on aligned input too (tests compare the two)
};

extern "C" int svs_stereo_create(svs_ctx *ctx, int w, int h, int max_batch, const svs_stereo_params *prm, svs_stereo **out) {
  SVS_REQUIRE(ctx, ctx && prm && out && w > 0 && h > 0 && max_batch > 0);
  SVS_DEVICE(ctx);
  if (prm->sad_window != 7 || prm->min_disparity != 0 || prm->num_disparities != NDISP || prm->prefilter_cap < 1 || prm->prefilter_cap > 63 ||
      w < NDISP + 2 * WSZ2 || w > 65535 || h < 2 || prm->speckle_window > 65535) {
    ctx->err = "svs_stereo: only SADWindowSize 7, minDisparity 0, numberOfDisparities 32, preFilterCap 1..63, w >= 38 are supported";
    return SVS_ERR_UNSUPPORTED;
  }
  svs_stereo *s = new svs_stereo();
  s->ctx = ctx; s->w = w; s->h = h; s->max_batch = max_batch; s->pitch = (w + PADL + PADR + 3) & ~3; s->prm = *prm;
  { const char *e = getenv("SVS_STEREO_FRAME_CCL"); s->force_frame_ccl = e && atoi(e) != 0; }
  { const char *e = getenv("SVS_STEREO_PREFILTER4"); s->force_prefilter4 = e && atoi(e) != 0; }      // A/B: the generic 4-pixel kernel on aligned input too
  { const char *e = getenv("SVS_STEREO_DEBUG"); s->debug = e && atoi(e) != 0; }
  const size_t n = (size_t)w * h * max_batch, np = (size_t)s->pitch * h * max_batch + 64;
  SVS_HIP(ctx, hipMalloc(&s->d_lp, np));
  SVS_HIP(ctx, hipMalloc(&s->d_rp, np));
  SVS_HIP(ctx, hipMalloc(&s->d_disp16, n * sizeof(int16_t)));
  SVS_HIP(ctx, hipMalloc(&s->d_cost, n * sizeof(uint16_t)));
  SVS_HIP(ctx, hipMalloc(&s->d_label, n * sizeof(int32_t)));
  SVS_HIP(ctx, hipMalloc(&s->d_count, n * sizeof(int32_t)));
  // strip speckle filter where a strip of >= 16 rows (labels 4 B + disparities 2 B per pixel) fits LDS, and the saturating run counts of
  // SPK_THREADS concurrent adds stay below the mark bit
  {
    const int wp = (w + 3) & ~3;
    int strip_kb = 151;                                  // dynamic LDS of a strip; the kernel's static arrays take 6 KB of the CU's 160 KB
    { const char *e = getenv("SVS_STEREO_STRIP_KB"); if (e && atoi(e) >= 8 && atoi(e) <= 151) strip_kb = atoi(e); }      // experiment: smaller strips, more workgroups per CU
    const int rows = std::min(h, (strip_kb * 1024) / spk_row_bytes(w));
    if (rows >= 16 && w <= 64 * SPK_MAX_SEG && (size_t)rows * w < (1u << 24) && (long long)(SPK_THREADS + 1) * std::max(prm->speckle_window, w) < SPK_TOUCH) {
      s->strip_rows = rows; s->n_strips = div_up(h, rows);
      SVS_HIP(ctx, hipMalloc(&s->d_brow, sizeof(int32_t) * 2 * (size_t)s->n_strips * w * max_batch));
      SVS_HIP(ctx, hipMalloc(&s->d_pend, sizeof(int2) * n));
      SVS_HIP(ctx, hipMalloc(&s->d_npend, sizeof(int32_t) * ((size_t)max_batch * s->n_strips + 32)));
      SVS_HIP(ctx, hipMemset(s->d_npend, 0, sizeof(int32_t) * ((size_t)max_batch * s->n_strips + 32)));
      SVS_HIP(ctx, hipFuncSetAttribute((const void *)stereo_speckle_strip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, rows * spk_row_bytes(w)));
    }
  }
  *out = s;
  return SVS_OK;
}

extern "C" int svs_stereo_destroy(svs_stereo *s) {
  if (!s) return SVS_OK;
  (void)hipStreamSynchronize(s->ctx->stream);
  if (s->d_lp) (void)hipFree(s->d_lp); if (s->d_rp) (void)hipFree(s->d_rp);
  if (s->d_disp16) (void)hipFree(s->d_disp16); if (s->d_cost) (void)hipFree(s->d_cost);
  if (s->d_label) (void)hipFree(s->d_label); if (s->d_count) (void)hipFree(s->d_count);
  if (s->d_brow) (void)hipFree(s->d_brow); if (s->d_pend) (void)hipFree(s->d_pend); if (s->d_npend) (void)hipFree(s->d_npend);
  delete s;
  return SVS_OK;
}

extern "C" int svs_stereo_compute(svs_stereo *s, const uint8_t *d_left, int lstride, size_t l_bstride, const uint8_t *d_right, int rstride,
                                  size_t r_bstride, float *d_disp, int dstride, size_t d_bstride, int n_batch) {
  svs_ctx *ctx = s ? s->ctx : nullptr;
  SVS_REQUIRE(ctx, s && d_left && d_right && d_disp && n_batch >= 1 && n_batch <= s->max_batch && lstride >= s->w && rstride >= s->w && dstride >= s->w);
  SVS_DEVICE(ctx);
  StereoDev S{};
  S.w = s->w; S.h = s->h; S.pitch = s->pitch;
  S.cap = s->prm.prefilter_cap; S.texthr = s->prm.texture_threshold; S.uniq = s->prm.uniqueness_ratio;
  S.speckle_window = s->prm.speckle_window; S.speckle_range = s->prm.speckle_range; S.disp12 = s->prm.disp12_max_diff;
  S.lp = s->d_lp; S.rp = s->d_rp; S.disp16 = s->d_disp16; S.cost = s->d_cost; S.label = s->d_label; S.count = s->d_count;
  S.swz = ctx->xcd_swizzle;
  const int w = s->w, h = s->h, width1 = w - NDISP + 1, n = w * h;
  const bool aligned16 = w % 16 == 0 && s->pitch % 16 == 0 && lstride % 4 == 0 && rstride % 4 == 0 && l_bstride % 4 == 0 && r_bstride % 4 == 0 &&
                         ((uintptr_t)d_left | (uintptr_t)d_right) % 4 == 0 && !s->force_prefilter4;
  if (aligned16)
    hipLaunchKernelGGL(stereo_prefilter16_kernel, dim3(div_up((s->pitch / 16) * div_up(h, 4), 256), 1, 2 * n_batch), dim3(256), 0, ctx->stream, S, d_left, lstride,
                       l_bstride, d_right, rstride, r_bstride);
  else
    hipLaunchKernelGGL(stereo_prefilter_kernel, dim3(div_up(s->pitch, 256), div_up(h, 16), 2 * n_batch), dim3(64, 4), 0, ctx->stream, S, d_left, lstride,
                       l_bstride, d_right, rstride, r_bstride);
  SVS_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(stereo_bm_edge_kernel, dim3(div_up(h, EDGE_ROWS), n_batch), dim3(EDGE_THREADS), 0, ctx->stream, S);
  SVS_LAUNCH_CHECK(ctx);
  if (width1 > 3) {
    if (S.cap <= 41) hipLaunchKernelGGL(stereo_bm_kernel<true>, dim3(div_up(width1 - 3, 64), div_up(h, BM_STRIP), n_batch), dim3(64), 0, ctx->stream, S);
    else hipLaunchKernelGGL(stereo_bm_kernel<false>, dim3(div_up(width1 - 3, 64), div_up(h, BM_STRIP), n_batch), dim3(64), 0, ctx->stream, S);
    SVS_LAUNCH_CHECK(ctx);
  }
  if (s->prm.disp12_max_diff >= 0) {
    { const int R = w <= 2048 ? 4 : 1; hipLaunchKernelGGL(stereo_validate_kernel, dim3(div_up(h, R), n_batch), dim3(64 * R), sizeof(int) * 2 * R * (size_t)w, ctx->stream, S); }
    SVS_LAUNCH_CHECK(ctx);
  }
  const bool ccl = s->prm.speckle_range >= 0 && s->prm.speckle_window > 0;
  const dim3 gp(div_up(n, 256), n_batch);
  if (ccl && s->strip_rows > 0 && !s->force_frame_ccl) {
    S.strip_rows = s->strip_rows; S.n_strips = s->n_strips; S.brow = s->d_brow; S.pend = s->d_pend; S.npend = s->d_npend; S.err = s->d_npend + (((size_t)s->max_batch * s->n_strips + 1) & ~(size_t)1); S.timing = s->debug;
    hipLaunchKernelGGL(stereo_speckle_strip_kernel, dim3(s->n_strips, n_batch), dim3(SPK_THREADS), (size_t)s->strip_rows * spk_row_bytes(w), ctx->stream, S, d_disp, dstride, d_bstride);
    SVS_LAUNCH_CHECK(ctx);
    if (s->n_strips > 1) {
      hipLaunchKernelGGL(stereo_speckle_merge_kernel, dim3(s->n_strips - 1, n_batch), dim3(256), 0, ctx->stream, S);
      for (int pass = 0; pass < 2; ++pass)
        hipLaunchKernelGGL(stereo_speckle_resolve_kernel, dim3(s->n_strips, n_batch), dim3(256), 0, ctx->stream, S, pass, d_disp, dstride, d_bstride);
      SVS_LAUNCH_CHECK(ctx);
    }
    if (s->debug) {
      int ev[24] = {};
      SVS_HIP(ctx, hipMemcpyAsync(ev, S.err, sizeof(ev), hipMemcpyDeviceToHost, ctx->stream));
      SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      const int e = ev[0];
      std::vector<int> np((size_t)n_batch * s->n_strips);
      SVS_HIP(ctx, hipMemcpy(np.data(), s->d_npend, sizeof(int) * np.size(), hipMemcpyDeviceToHost));
      long tot = 0; int mx = 0;
      for (int v : np) { tot += v; mx = std::max(mx, v); }
      { const long long *t = reinterpret_cast<const long long *>(ev + 8);
        fprintf(stderr, "[svs_stereo] strip kernel phases of one workgroup (us): load %.1f runs %.1f stitch %.1f count %.1f classify %.1f convert %.1f\n", (t[1] - t[0]) * 0.01,
                (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01, (t[6] - t[5]) * 0.01); }
      fprintf(stderr, "[svs_stereo] strip speckle filter: error mask %d, undecided pixels %ld in %d frames (max %d per strip)\n", e, tot, n_batch, mx);
    }
    return SVS_OK;
  }
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
