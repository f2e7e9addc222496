// fast.hip -- grid-adaptive FAST-9/16 for gfx950.  Replaces FastGrid::detectAdaptively /
// FastGrid::detect (fast_grid.cpp:60-152), whose inner call is cv::FastFeatureDetector.
//
// MI355X-first design (not the reference's "re-run FAST up to 6 times per cell"):
//   K1 score:   one LDS-tiled pass per frame computes, for every pixel of every cell ROI, the
//               FAST score  s = max t such that the pixel is a FAST-9 corner at threshold t
//               (the segment test is monotone in t) and a 256-bin histogram of
//               scores per cell.  The <=6 adaptive re-detections of the reference then are
//               histogram look-ups:  count(t) = #pixels with s >= t.
//   K2 adapt:   one workgroup per camera stream runs the reference's per-row threshold state
//               machine (prev_thr / prev_prev_thr shared along a row of cells, +-1/+-2 steps,
//               clamps) on the suffix-summed histograms and produces the emit threshold, the
//               persistent threshold and the output offset of every cell.
//   K3 compact: one workgroup per cell emits the corners of the LAST executed detection in the
//               reference's order (cells row-major, row-major inside the cell ROI) and leaves them as
//               a 1-bit-per-pixel CORNER BITMAP for the matcher.
// There is no dense score image (rounds 1-5 wrote one byte per pixel and the compaction and the matcher read it back):
// the score kernel leaves, per 64 x 32 tile, the sparse list of pixels whose score reaches t_lo (the lowest threshold
// any cell can take) and the per-cell histograms; the compaction sets the listed pixels that reach the cell's emit
// threshold in an LDS bitmap of the cell, counts / scans / emits from the bitmap (row-major = the reference's
// order, any number of corners) and stores the bitmap; match.hip tests window positions against it (the reference's
// matcher asks a quadtree of the emitted corners, matcher.cpp:351-357 -- not a score image).
// Integer arithmetic throughout => corner lists are bit-exact to the oracle.
#include "common.h"
#include "fast_view.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

struct LevelDev {
  int w, h;
  int gx, gy, cell_w, cell_h;
  int min_inner, min_outer, max_inner, max_outer, fast_min, fast_max;
  int cell_base;          // index of this level's first cell in the per-slot cell arrays
  // corner bitmap [batch][h][bm_stride bytes]: pixel (x, y) of cell column ci is bit ci * 32 * bm_wpr + (x - ci * cell_w) of row y -- every cell column starts on a
  // dword, so a cell's workgroup owns its dwords (plain stores of the whole cell every frame: no atomics, nothing to clear) and the pad bits between columns stay 0
  uint32_t *bm;
  int bm_wpr, bm_stride;  // dwords per cell row; bytes per image row (>= 8 bytes of zero padding behind the last bit a window can ask for)
  size_t bm_bstride;      // bytes per slot
  int16_t *xy;            // [batch][cap][2]
};
struct FastParams {
  LevelDev lv[SVS_NUM_PYR_LEVELS];
  int n_levels, ncell_total, cap, t_lo;
  unsigned *hist;         // [batch][ncell_total][256]
  int *thr, *emit, *count, *offset;   // [batch][ncell_total]
  int *level_total;       // [batch][n_levels]
  // corner candidates (score >= t_lo) of every tile, written by the score kernel.  cand [batch][n_tiles][CAND_CAP] = score8 << 24 | cell-local y << 12 | x;
  // CAND_CAP = the pixels of a tile, so a list cannot overflow (2 MB of address space per 640 x 480 frame slot, of which a textured frame touches ~25 KB)
  uint32_t *cand; int *cand_n;      // cand_n [batch][n_tiles]
  int *cell_tile0, *cell_ntile;     // [ncell_total]: tiles of a cell are contiguous in the tile list
  int n_tiles;
  int swz;                          // tiles in XCD-contiguous order (common.h: xcd_contiguous)
};
constexpr int CAND_CAP = 2048;      // per 64x32 tile = its pixels
constexpr int BM_LDS_MAX = 144 * 1024;      // LDS bytes of the largest cell's bitmap + row offsets the compaction may use (checked at svs_fast_create)
struct ImgPtrs {
  const uint8_t *img[SVS_NUM_PYR_LEVELS];
  int stride[SVS_NUM_PYR_LEVELS];
  size_t bstride[SVS_NUM_PYR_LEVELS];
};
struct TileDesc { int16_t level, cell, x0, y0; };   // cell = index within the level

constexpr int TW = 64, TH = 32, HALO = 3;   // tile 64x32: two 16-row passes per lane amortise the per-block load/sync latency

// ring offsets, SURVEY.md A.1 (index 0..15)
__device__ __constant__ int8_t c_ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__device__ __constant__ int8_t c_ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// max over the 16 arcs of 9 of the arc minimum, via the 2-4-8(+1) doubling ladder
__device__ __forceinline__ int arc9_maxmin(const int (&d)[16]) {
  int m2[16], m4[16], best = -1024;
#pragma unroll
  for (int k = 0; k < 16; ++k) m2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    int m8 = min(m4[k], m4[(k + 4) & 15]);
    int m9 = min(m8, d[(k + 8) & 15]);
    best = max(best, m9);
  }
  return best;
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t u) { s16x2 r; __builtin_memcpy(&r, &u, 4); return r; }
__device__ __forceinline__ s16x2 arc9_maxmin_pk(const s16x2 (&d)[16]) {      // arc9_maxmin on two independent 16-bit lanes
  s16x2 m2[16], m4[16], best = {-1024, -1024};
#pragma unroll
  for (int k = 0; k < 16; ++k) m2[k] = __builtin_elementwise_min(d[k], d[(k + 1) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) m4[k] = __builtin_elementwise_min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const s16x2 m8 = __builtin_elementwise_min(m4[k], m4[(k + 4) & 15]);
    const s16x2 m9 = __builtin_elementwise_min(m8, d[(k + 8) & 15]);
    best = __builtin_elementwise_max(best, m9);
  }
  return best;
}

// K1.  Each lane pre-tests 4 horizontally adjacent pixels per step from three LDS rows (phase A), the survivors are scored one
// per lane (phase B); the tile itself is staged with (unaligned) dword loads.  Nothing per pixel leaves the kernel: the pixels
// whose score reaches t_lo go to the tile's candidate list and the cell's histogram.
// LDS row = 72 bytes: [4 left halo | 64 tile | 4 right halo].
constexpr int LROW = 19;   // dwords per LDS row (18 used + 1 pad)
__device__ __forceinline__ int byte_of(const uint32_t (&w)[3], int b) { return (w[b >> 2] >> (8 * (b & 3))) & 0xff; }
// bytes b and b+1 of the 12-byte row window, zero-extended into the two 16-bit halves (one v_perm_b32)
__device__ __forceinline__ uint32_t pair_of(const uint32_t (&w)[3], int b) {
  const int j = b >> 2, j1 = (j + 1 < 3) ? j + 1 : j;
  const unsigned i0 = (unsigned)(b & 3);                                   // byte b lives in w[j]   (src1: selectors 0..3)
  const unsigned i1 = ((b + 1) >> 2) == j ? (unsigned)((b + 1) & 3) : 4u + (unsigned)((b + 1) & 3);   // w[j+1] = src0: 4..7
  return __builtin_amdgcn_perm(w[j1], w[j], i0 | (0x0cu << 8) | (i1 << 16) | (0x0cu << 24));
}

__global__ __launch_bounds__(256, 8) void fast_score_kernel(FastParams P, ImgPtrs I, const TileDesc *__restrict__ tiles) {
  __shared__ uint32_t s_img[(TH + 2 * HALO) * LROW];
  __shared__ unsigned s_hist[256];
  __shared__ uint16_t s_surv[TW * TH];    // pixels that pass the compass pre-test
  __shared__ int s_ncorn, s_nsurv;
  // neighbouring tiles share their halo rows (a 72-byte row of a tile lies in the 128-byte lines of two tiles): with the dispatcher's round robin the eight XCDs work on the
  // same frame and every line is fetched into 2-3 L2s; in XCD-contiguous order an XCD works through whole frames
  unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
  if (P.swz) wg = xcd_contiguous(wg, gridDim.x * gridDim.y);
  const int tile_i = wg % gridDim.x, slot = wg / gridDim.x;
  const TileDesc td = tiles[tile_i];
  const LevelDev &L = P.lv[td.level];
  const int tid = threadIdx.x;
  const int ci = td.cell % L.gx, cj = td.cell / L.gx;
  const int u0 = ci * L.cell_w, v0 = cj * L.cell_h;       // cell ROI origin (fast_grid.cpp:45-52)
  const uint8_t *img = I.img[td.level] + (size_t)slot * I.bstride[td.level];
  const int istride = I.stride[td.level];
  s_hist[tid] = 0;
  // stage the tile: rows v0+y0-3 .. +18, bytes u0+x0-4 .. +67.  Anything outside the image is
  // clamped (never used: only ROI-interior pixels are scored and their ring stays inside the ROI)
  const int gx0 = u0 + td.x0 - 4, gy0 = v0 + td.y0 - HALO;
  for (int i = tid; i < (TH + 2 * HALO) * 18; i += 256) {
    const int r = i / 18, c = i - r * 18;
    const int y = min(max(gy0 + r, 0), L.h - 1), x = gx0 + 4 * c;
    const uint8_t *row = img + (size_t)y * istride;
    // branch-free: a dword from the clamped position, shifted so that the in-image bytes land where they belong; the bytes
    // outside the image become 0 and are never used (only ROI-interior pixels are scored and their ring stays inside the
    // ROI).  A per-byte border path would put a wait behind every iteration's load.
    const int xs = min(max(x, 0), L.w - 4), d = x - xs;
    uint32_t raw;
    __builtin_memcpy(&raw, row + xs, 4);
    const uint32_t v = d == 0 ? raw : d >= 4 || d <= -4 ? 0u : d > 0 ? raw >> (8 * d) : raw << (-8 * d);
    s_img[r * LROW + c] = v;
  }
  __syncthreads();
  const int tx = tid & 15;
  const int cx0 = td.x0 + 4 * tx;                            // cell-local x of the lane's 4 pixels
  // ---- phase A (round 3): the compass pre-test for all pixels.  Any arc of 9 contiguous ring pixels contains two CONSECUTIVE compass points
  //      (ring positions 0, 4, 8, 12), so a corner at t needs two consecutive compass pixels that are both brighter than v + t or both darker than
  //      v - t.  Four ring positions instead of sixteen, three LDS rows instead of seven; ~20 % of the pixels of a textured frame survive at
  //      t_lo = 10 and are queued (the full 16-position boolean test cost 4x as much on every pixel).
  if (tid == 0) { s_ncorn = 0; s_nsurv = 0; }
  __syncthreads();
  const int t = P.t_lo;
  constexpr int RDX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int RDY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
#pragma unroll 1
  for (int ty = tid >> 4; ty < TH; ty += 16) {
    const int cy = td.y0 + ty;
    uint32_t win[7][3];      // only rows 0, 3, 6 are read (ring positions 8, {4, 12}, 0)
#pragma unroll
    for (int r = 0; r < 7; r += 3)
#pragma unroll
      for (int c = 0; c < 3; ++c) win[r][c] = s_img[(ty + r) * LROW + tx + c];
    const bool row_ok = cy >= 3 && cy < L.cell_h - 3;
    // SWAR: two horizontally adjacent pixels ride in the two 16-bit halves of a dword.
    //   darker  <=> v - r > t <=> bit 15 of (v + 0x7fff - t) - r      (no carry/borrow between halves)
    //   brighter<=> r - v > t <=> bit 15 of r + (0x7fff - t - v)
    const uint32_t Kd = 0x7fff7fffu - (uint32_t)t * 0x00010001u;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const uint32_t vpair = pair_of(win[3], 4 + 2 * p);
      const uint32_t Vd = vpair + Kd, Vb = Kd - vpair;
      // the verdicts stay where the subtraction leaves them (bit 15 of each half): two consecutive compass points (cyclically) in one polarity is
      // (X0 & X1) | (X1 & X2) | (X2 & X3) | (X3 & X0) on those bits -- four V_AND_OR per polarity, no mask building
      uint32_t X[4], Y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = 4 * i;
        const uint32_t rp = pair_of(win[3 + RDY[q]], 4 + 2 * p + RDX[q]);
        X[i] = Vd - rp; Y[i] = rp + Vb;
      }
      const uint32_t pass = ((X[0] & X[1]) | (X[1] & X[2]) | (X[2] & X[3]) | (X[3] & X[0]) | (Y[0] & Y[1]) | (Y[1] & Y[2]) | (Y[2] & Y[3]) | (Y[3] & Y[0])) & 0x80008000u;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = 2 * p + h, cx = cx0 + k;
        const bool surv = row_ok && cx >= 3 && cx < L.cell_w - 3 && ((pass >> (16 * h)) & 0x8000u) != 0;
        if (surv) { const int slot_i = atomicAdd(&s_nsurv, 1); s_surv[slot_i] = (uint16_t)((ty << 8) | (4 * tx + k)); }
      }
    }
  }
  __syncthreads();
  // ---- phase B: FAST score (max t) of the survivors; those that are corners at t_lo (score >= t_lo) go to the tile's candidate list and the cell's histogram
  const int nsurv = s_nsurv;
  const uint8_t *s_b8 = reinterpret_cast<const uint8_t *>(s_img);
  uint32_t *cand = P.cand + ((size_t)slot * P.n_tiles + tile_i) * CAND_CAP;
  for (int i = tid; i < nsurv; i += 256) {
    const int code = s_surv[i], py = code >> 8, px = code & 0xff;           // tile-local pixel
    const int cb = (py + 3) * (LROW * 4) + px + 4;
    const int v = s_b8[cb];
    // both polarities at once: v - r in the low and r - v in the high 16-bit half, the min / max ladder on V_PK_MIN_I16 / V_PK_MAX_I16
    s16x2 d[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = s_b8[cb + RDY[q] * (LROW * 4) + RDX[q]];
      const int x = v - r;
      d[q] = as_s16x2(((uint32_t)x & 0xffffu) | ((uint32_t)(-x) << 16));
    }
    const s16x2 b2 = arc9_maxmin_pk(d);
    const int sc = max((int)b2.x, (int)b2.y) - 1;                           // corner at t <=> sc >= t
    if (sc >= t) {
      const int s8 = min(sc + 1, 255);
      atomicAdd(&s_hist[s8], 1u);
      cand[atomicAdd(&s_ncorn, 1)] = ((uint32_t)s8 << 24) | ((uint32_t)(td.y0 + py) << 12) | (uint32_t)(td.x0 + px);      // (at most the tile's pixels = CAND_CAP)
    }
  }
  __syncthreads();
  if (tid == 0) P.cand_n[(size_t)slot * P.n_tiles + tile_i] = s_ncorn;
  const unsigned c = s_hist[tid];
  if (c) atomicAdd(&P.hist[((size_t)slot * P.ncell_total + L.cell_base + td.cell) * 256 + tid], c);
}

// K2: threshold state machine of fast_grid.cpp:86-152 on histogram counts.
__global__ __launch_bounds__(256) void fast_adapt_kernel(FastParams P, int trials) {
  extern __shared__ int s_cnt[];    // [ncell_total][256]: #pixels with stored score >= bin
  const int slot = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned *hist = P.hist + (size_t)slot * P.ncell_total * 256;
  for (int c = wave; c < P.ncell_total; c += 4) {
    // lane holds bins 4*lane..4*lane+3; suffix sum inside the lane then across lanes
    uint4 hv = *reinterpret_cast<const uint4 *>(hist + (size_t)c * 256 + 4 * lane);
    *reinterpret_cast<uint4 *>(hist + (size_t)c * 256 + 4 * lane) = make_uint4(0, 0, 0, 0);  // ready for next frame
    int a3 = hv.w, a2 = hv.z + a3, a1 = hv.y + a2, a0 = hv.x + a1;
    int tot = a0, run = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int up = __shfl_down(run, o, 64); if (lane + o < 64) run += up; }
    int above = run - tot;   // sum of bins of higher lanes
    s_cnt[c * 256 + 4 * lane + 0] = a0 + above;
    s_cnt[c * 256 + 4 * lane + 1] = a1 + above;
    s_cnt[c * 256 + 4 * lane + 2] = a2 + above;
    s_cnt[c * 256 + 4 * lane + 3] = a3 + above;
  }
  __syncthreads();
  int *thr = P.thr + (size_t)slot * P.ncell_total;
  int *emit = P.emit + (size_t)slot * P.ncell_total;
  int *count = P.count + (size_t)slot * P.ncell_total;
  int *offset = P.offset + (size_t)slot * P.ncell_total;
  // one thread per (level, row of cells): prev_thr/prev_prev_thr are shared along the row
  int row = tid, lvl = -1, j = 0;
  for (int l = 0; l < P.n_levels; ++l) { if (row < P.lv[l].gy) { lvl = l; j = row; break; } row -= P.lv[l].gy; }
  if (lvl >= 0) {
    const LevelDev &L = P.lv[lvl];
    int prev_thr = -1, prev_prev_thr = -2;
    for (int i = 0; i < L.gx; ++i) {
      const int c = L.cell_base + j * L.gx + i;
      int t = thr[c], used = t, num = 0;
      if (trials <= 0) {             // FastGrid::detect: one pass at the stored threshold
        int tc = min(max(t, 0), 255);
        num = tc + 1 <= 255 ? s_cnt[c * 256 + tc + 1] : 0;
      }
      for (int trial = 0; trial < trials; ++trial) {
        used = t;
        int tc = min(max(t, 0), 255);
        num = tc + 1 <= 255 ? s_cnt[c * 256 + tc + 1] : 0;
        if (prev_prev_thr == t) { t = (t + prev_prev_thr) / 2; break; }
        prev_prev_thr = prev_thr;
        prev_thr = t;
        if (num < L.min_inner) {
          if (t <= L.fast_min) break;
          --t;
          if (num < L.min_outer) { if (t <= L.fast_min) break; --t; continue; }
        } else if (num > L.max_inner) {
          if (t >= L.fast_max) break;
          ++t;
          if (num > L.max_outer) { if (t >= L.fast_max) break; ++t; continue; }
        }
        break;
      }
      thr[c] = t; emit[c] = used; count[c] = num;
    }
  }
  __syncthreads();
  if (tid < P.n_levels) {            // exclusive scan of the counts of one level
    const LevelDev &L = P.lv[tid];
    int run = 0;
    for (int c = L.cell_base; c < L.cell_base + L.gx * L.gy; ++c) { offset[c] = run; run += count[c]; }
    P.level_total[(size_t)slot * P.n_levels + tid] = run;
  }
}

// K3: ordered compaction of one cell from the candidate lists.  The listed pixels that reach the cell's emit threshold are set in an LDS bitmap of the
// cell (one wave per tile list; LDS atomic OR, any order), a thread per row counts its bits, one wave scans the row counts, a thread per bitmap
// dword emits its corners -- row-major = the reference's order (fast_grid.cpp:60-83: the cell's keypoints in cv::FAST's scan order) -- and the
// bitmap leaves as the cell's part of the level's corner bitmap (match.hip).  No cap on the corners of a cell, nothing sorted, one launch.
// grid: (ncell_total, batch); dynamic LDS: (rows * wpr + rows) * 4 bytes of the largest cell.
template <int NT>      // 1024 when few cells are in flight (latency mode), 256 when the batch fills the device
__global__ __launch_bounds__(NT) void fast_compact_bitmap_kernel(FastParams P) {
  extern __shared__ uint32_t s_bm[];      // [rows][wpr] bitmap, then [rows] row offsets
  const int slot = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int lvl = 0;
  while (lvl + 1 < P.n_levels && c >= P.lv[lvl + 1].cell_base) ++lvl;
  const LevelDev &L = P.lv[lvl];
  const int cl = c - L.cell_base, ci = cl % L.gx, cj = cl / L.gx;
  const int u0 = ci * L.cell_w, v0 = cj * L.cell_h;
  const int rows = L.cell_h, wpr = L.bm_wpr;
  int *s_row = reinterpret_cast<int *>(s_bm + rows * wpr);
  const int thr1 = min(max(P.emit[(size_t)slot * P.ncell_total + c], 0), 255) + 1;
  const int base = P.offset[(size_t)slot * P.ncell_total + c];
  for (int i = tid; i < rows * wpr; i += NT) s_bm[i] = 0u;
  __syncthreads();
  if (thr1 <= 255) {
    const int t0 = P.cell_tile0[c], nt = P.cell_ntile[c];
    for (int t = wave; t < nt; t += NT / 64) {
      const int n = min(P.cand_n[(size_t)slot * P.n_tiles + t0 + t], CAND_CAP);
      const uint32_t *cand = P.cand + ((size_t)slot * P.n_tiles + t0 + t) * CAND_CAP;
      for (int i = lane; i < n; i += 64) {
        const uint32_t r = cand[i];
        if ((int)(r >> 24) >= thr1) {
          const int x = r & 0xfffu, y = (r >> 12) & 0xfffu;      // cell-local
          atomicOr(&s_bm[y * wpr + (x >> 5)], 1u << (x & 31));
        }
      }
    }
  }
  __syncthreads();
  for (int r = tid; r < rows; r += NT) {
    int n = 0;
    for (int j = 0; j < wpr; ++j) n += __popc(s_bm[r * wpr + j]);
    s_row[r] = n;
  }
  __syncthreads();
  if (wave == 0) {                                       // exclusive scan of the row counts, 64 rows per step
    int carry = 0;
    for (int r0 = 0; r0 < rows; r0 += 64) {
      const int v = r0 + lane < rows ? s_row[r0 + lane] : 0;
      int run = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int dn = __shfl_up(run, o, 64); if (lane >= o) run += dn; }
      if (r0 + lane < rows) s_row[r0 + lane] = carry + run - v;
      carry += __shfl(run, 63, 64);
    }
  }
  __syncthreads();
  int16_t *xy = L.xy + (size_t)slot * P.cap * 2;
  uint32_t *gbm = L.bm + (size_t)slot * (L.bm_bstride / 4) + (size_t)v0 * (L.bm_stride / 4) + ci * wpr;
  for (int i = tid; i < rows * wpr; i += NT) {
    const int r = i / wpr, j = i - r * wpr;
    uint32_t w = s_bm[i];
    gbm[(size_t)r * (L.bm_stride / 4) + j] = w;
    if (w) {
      int pos = base + s_row[r];
      for (int k = 0; k < j; ++k) pos += __popc(s_bm[r * wpr + k]);
      while (w) {
        const int b = __ffs((int)w) - 1;
        w &= w - 1;
        if (pos < P.cap) {      // (x, y) as one dword store
          const uint32_t pk = (uint32_t)(uint16_t)(u0 + 32 * j + b) | ((uint32_t)(uint16_t)(v0 + r) << 16);
          __builtin_memcpy(xy + 2 * pos, &pk, 4);
        }
        ++pos;
      }
    }
  }
}

}  // namespace

struct svs_fast {
  svs_ctx *ctx;
  FastParams P;
  int batch;
  TileDesc *d_tiles; int n_tiles;
  size_t cmp_lds = 0;         // dynamic LDS of the compaction: bitmap + row offsets of the largest cell
};

extern "C" int svs_fast_create(svs_ctx *ctx, int n_levels, const int32_t *w, const int32_t *h,
                               const svs_fastgrid *grids, int batch, int cap, svs_fast **out) {
  SVS_REQUIRE(ctx, ctx && out && w && h && grids && n_levels >= 1 && n_levels <= SVS_NUM_PYR_LEVELS && batch >= 1 && cap >= 1);
  SVS_DEVICE(ctx);
  {      // everything that can be refused is refused before the first allocation
    int nc = 0;
    for (int l = 0; l < n_levels; ++l) {
      const svs_fastgrid &g = grids[l];
      SVS_REQUIRE(ctx, g.gx >= 1 && g.gy >= 1 && g.gx * g.gy <= SVS_MAX_CELLS && g.cell_w >= 1 && g.cell_h >= 1 && g.cell_w * g.gx <= w[l] && g.cell_h * g.gy <= h[l]);
      SVS_REQUIRE(ctx, g.cell_w <= 4096 && g.cell_h <= 4096);      // 12-bit cell-local coordinates in the candidate records
      // a cell's bitmap lives in LDS during the compaction: cells up to ~1.1 M pixels (e.g. 1184 x 1024)
      SVS_REQUIRE(ctx, ((size_t)g.cell_h * ((g.cell_w + 31) / 32) + g.cell_h) * 4 <= (size_t)BM_LDS_MAX);
      nc += g.gx * g.gy;
    }
    SVS_REQUIRE(ctx, (size_t)nc * 256 * 4 <= 64 * 1024);          // the adaptation kernel's suffix-summed histograms of one stream in LDS
  }
  svs_fast *f = new svs_fast();
  f->ctx = ctx; f->batch = batch;
  FastParams &P = f->P;
  P = FastParams{};
  P.n_levels = n_levels; P.cap = cap;
  std::vector<TileDesc> tiles;
  std::vector<int> cell_tile0, cell_ntile;
  int cell_base = 0, t_lo = 255;
  for (int l = 0; l < n_levels; ++l) {
    const svs_fastgrid &g = grids[l];
    SVS_REQUIRE(ctx, g.gx >= 1 && g.gy >= 1 && g.gx * g.gy <= SVS_MAX_CELLS && g.cell_w * g.gx <= w[l] && g.cell_h * g.gy <= h[l]);
    SVS_REQUIRE(ctx, g.cell_w <= 4096 && g.cell_h <= 4096);      // 12-bit cell-local coordinates in the candidate records
    LevelDev &L = P.lv[l];
    L.w = w[l]; L.h = h[l]; L.gx = g.gx; L.gy = g.gy; L.cell_w = g.cell_w; L.cell_h = g.cell_h;
    L.min_inner = g.min_inner; L.min_outer = g.min_outer; L.max_inner = g.max_inner; L.max_outer = g.max_outer;
    L.fast_min = g.fast_min; L.fast_max = g.fast_max; L.cell_base = cell_base;
    // corner bitmap: cell columns on dword boundaries; a row ends with >= 8 zero bytes behind the last bit a window can ask for (match.hip reads 8 bytes from the
    // byte of any in-image pixel), rows below the cell grid and the pad bits are zeroed here and never written
    L.bm_wpr = (g.cell_w + 31) / 32;
    const int gap = 32 * L.bm_wpr - g.cell_w;
    L.bm_stride = ((std::max(w[l], g.gx * g.cell_w) + gap * (g.gx - 1) + 7) / 8 + 8 + 15) / 16 * 16;
    SVS_REQUIRE(ctx, L.bm_stride >= 4 * L.bm_wpr * g.gx);
    L.bm_bstride = (size_t)L.bm_stride * h[l] + 16;
    SVS_HIP(ctx, hipMalloc(&L.bm, L.bm_bstride * batch));
    SVS_HIP(ctx, hipMemsetAsync(L.bm, 0, L.bm_bstride * batch, ctx->stream));
    f->cmp_lds = std::max(f->cmp_lds, ((size_t)g.cell_h * L.bm_wpr + g.cell_h) * 4);
    SVS_HIP(ctx, hipMalloc(&L.xy, sizeof(int16_t) * 2 * (size_t)cap * batch));
    for (int c = 0; c < g.gx * g.gy; ++c) {
      t_lo = std::min(t_lo, std::min(g.fast_min, g.thr[c]));
      cell_tile0.push_back((int)tiles.size());
      for (int y0 = 0; y0 < g.cell_h; y0 += TH)
        for (int x0 = 0; x0 < g.cell_w; x0 += TW) tiles.push_back(TileDesc{(int16_t)l, (int16_t)c, (int16_t)x0, (int16_t)y0});
      cell_ntile.push_back((int)tiles.size() - cell_tile0.back());
    }
    cell_base += g.gx * g.gy;
  }
  P.ncell_total = cell_base;
  P.t_lo = std::max(t_lo, 0);
  SVS_REQUIRE(ctx, (size_t)P.ncell_total * 256 * 4 <= 64 * 1024);
  SVS_REQUIRE(ctx, f->cmp_lds <= (size_t)BM_LDS_MAX);      // a cell's bitmap lives in LDS during the compaction: cells up to ~1.1 M pixels (e.g. 1184 x 1024)
  if (f->cmp_lds > 64 * 1024) {
    SVS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fast_compact_bitmap_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f->cmp_lds));
    SVS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fast_compact_bitmap_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f->cmp_lds));
  }
  size_t nc = (size_t)P.ncell_total * batch;
  SVS_HIP(ctx, hipMalloc(&P.hist, nc * 256 * sizeof(unsigned)));
  SVS_HIP(ctx, hipMemsetAsync(P.hist, 0, nc * 256 * sizeof(unsigned), ctx->stream));
  SVS_HIP(ctx, hipMalloc(&P.thr, nc * sizeof(int)));
  SVS_HIP(ctx, hipMalloc(&P.emit, nc * sizeof(int)));
  SVS_HIP(ctx, hipMalloc(&P.count, nc * sizeof(int)));
  SVS_HIP(ctx, hipMalloc(&P.offset, nc * sizeof(int)));
  SVS_HIP(ctx, hipMalloc(&P.level_total, (size_t)batch * n_levels * sizeof(int)));
  SVS_HIP(ctx, hipMemsetAsync(P.emit, 0, nc * sizeof(int), ctx->stream));
  SVS_HIP(ctx, hipMemsetAsync(P.count, 0, nc * sizeof(int), ctx->stream));
  SVS_HIP(ctx, hipMemsetAsync(P.offset, 0, nc * sizeof(int), ctx->stream));
  SVS_HIP(ctx, hipMemsetAsync(P.level_total, 0, (size_t)batch * n_levels * sizeof(int), ctx->stream));
  std::vector<int> thr0(nc);
  for (int b = 0; b < batch; ++b)
    for (int l = 0; l < n_levels; ++l)
      for (int c = 0; c < grids[l].gx * grids[l].gy; ++c) thr0[(size_t)b * P.ncell_total + P.lv[l].cell_base + c] = grids[l].thr[c];
  SVS_HIP(ctx, hipMemcpyAsync(P.thr, thr0.data(), nc * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  f->n_tiles = (int)tiles.size();
  P.n_tiles = f->n_tiles;
  SVS_HIP(ctx, hipMalloc(&P.cand, sizeof(uint32_t) * (size_t)batch * tiles.size() * CAND_CAP));
  SVS_HIP(ctx, hipMalloc(&P.cand_n, sizeof(int) * (size_t)batch * tiles.size()));
  SVS_HIP(ctx, hipMalloc(&P.cell_tile0, sizeof(int) * cell_tile0.size()));
  SVS_HIP(ctx, hipMalloc(&P.cell_ntile, sizeof(int) * cell_ntile.size()));
  SVS_HIP(ctx, hipMemcpyAsync(P.cell_tile0, cell_tile0.data(), sizeof(int) * cell_tile0.size(), hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipMemcpyAsync(P.cell_ntile, cell_ntile.data(), sizeof(int) * cell_ntile.size(), hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipMalloc(&f->d_tiles, sizeof(TileDesc) * tiles.size()));
  SVS_HIP(ctx, hipMemcpyAsync(f->d_tiles, tiles.data(), sizeof(TileDesc) * tiles.size(), hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = f;
  return SVS_OK;
}

extern "C" int svs_fast_destroy(svs_fast *f) {
  if (!f) return SVS_OK;
  (void)hipStreamSynchronize(f->ctx->stream);
  for (int l = 0; l < f->P.n_levels; ++l) { hipFree(f->P.lv[l].bm); hipFree(f->P.lv[l].xy); }
  hipFree(f->P.hist); hipFree(f->P.thr); hipFree(f->P.emit); hipFree(f->P.count); hipFree(f->P.offset);
  hipFree(f->P.level_total); hipFree(f->d_tiles);
  hipFree(f->P.cand); hipFree(f->P.cand_n); hipFree(f->P.cell_tile0); hipFree(f->P.cell_ntile);
  delete f;
  return SVS_OK;
}

extern "C" int svs_fast_detect(svs_fast *f, const uint8_t *const *d_img, const int32_t *stride, const size_t *bstride,
                               int n_batch, int trials) {
  SVS_REQUIRE(f ? f->ctx : nullptr, f && d_img && stride && bstride && n_batch >= 1 && n_batch <= f->batch && trials >= 0);
  SVS_DEVICE(f->ctx);
  svs_ctx *ctx = f->ctx;
  ImgPtrs I{};
  for (int l = 0; l < f->P.n_levels; ++l) { I.img[l] = d_img[l]; I.stride[l] = stride[l]; I.bstride[l] = bstride[l]; }
  f->P.swz = ctx->xcd_swizzle;
  hipLaunchKernelGGL(fast_score_kernel, dim3(f->n_tiles, n_batch), dim3(256), 0, ctx->stream, f->P, I, f->d_tiles);
  SVS_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(fast_adapt_kernel, dim3(n_batch), dim3(256), (size_t)f->P.ncell_total * 256 * sizeof(int), ctx->stream, f->P, trials);
  SVS_LAUNCH_CHECK(ctx);
  if ((long)f->P.ncell_total * n_batch <= 1024) hipLaunchKernelGGL(fast_compact_bitmap_kernel<1024>, dim3(f->P.ncell_total, n_batch), dim3(1024), f->cmp_lds, ctx->stream, f->P);
  else hipLaunchKernelGGL(fast_compact_bitmap_kernel<256>, dim3(f->P.ncell_total, n_batch), dim3(256), f->cmp_lds, ctx->stream, f->P);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

extern "C" int svs_fast_download(svs_fast *f, int slot, int level, int16_t *h_xy, int cap, int32_t *h_n,
                                 int32_t *h_cell_count, int32_t *h_emit_thr, int32_t *h_thr_state) {
  SVS_REQUIRE(f ? f->ctx : nullptr, f && slot >= 0 && slot < f->batch && level >= 0 && level < f->P.n_levels);
  SVS_DEVICE(f->ctx);
  svs_ctx *ctx = f->ctx;
  const FastParams &P = f->P;
  const LevelDev &L = P.lv[level];
  int total = 0;
  SVS_HIP(ctx, hipMemcpyAsync(&total, P.level_total + (size_t)slot * P.n_levels + level, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h_n) *h_n = total;
  int nc = L.gx * L.gy;
  size_t co = (size_t)slot * P.ncell_total + L.cell_base;
  if (h_cell_count) SVS_HIP(ctx, hipMemcpyAsync(h_cell_count, P.count + co, nc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  if (h_emit_thr) SVS_HIP(ctx, hipMemcpyAsync(h_emit_thr, P.emit + co, nc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  if (h_thr_state) SVS_HIP(ctx, hipMemcpyAsync(h_thr_state, P.thr + co, nc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  int ncopy = std::min(std::min(total, cap), P.cap);
  if (h_xy && ncopy > 0)
    SVS_HIP(ctx, hipMemcpyAsync(h_xy, L.xy + (size_t)slot * P.cap * 2, sizeof(int16_t) * 2 * (size_t)ncopy, hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h_xy && (total > cap || total > P.cap)) return SVS_ERR_CAPACITY;
  return SVS_OK;
}

extern "C" int svs_fast_set_thresholds(svs_fast *f, int slot, int level, const int32_t *h_thr) {
  SVS_REQUIRE(f ? f->ctx : nullptr, f && h_thr && slot >= 0 && slot < f->batch && level >= 0 && level < f->P.n_levels);
  SVS_DEVICE(f->ctx);
  svs_ctx *ctx = f->ctx;
  const LevelDev &L = f->P.lv[level];
  int nc = L.gx * L.gy;
  for (int c = 0; c < nc; ++c) SVS_REQUIRE(ctx, h_thr[c] >= f->P.t_lo);   // scores below t_lo are not kept
  SVS_HIP(ctx, hipMemcpyAsync(f->P.thr + (size_t)slot * f->P.ncell_total + L.cell_base, h_thr, nc * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVS_OK;
}

extern "C" int svs_fast_device_view(svs_fast *f, int level, const uint32_t **d_corner_bits, int32_t *row_stride_bytes, size_t *batch_stride_bytes,
                                    int32_t *cell_column_bits, const int32_t **d_emit_thr, size_t *emit_bstride) {
  SVS_REQUIRE(f ? f->ctx : nullptr, f && level >= 0 && level < f->P.n_levels);
  const LevelDev &L = f->P.lv[level];
  if (d_corner_bits) *d_corner_bits = L.bm;
  if (row_stride_bytes) *row_stride_bytes = L.bm_stride;
  if (batch_stride_bytes) *batch_stride_bytes = L.bm_bstride;
  if (cell_column_bits) *cell_column_bits = 32 * L.bm_wpr;
  if (d_emit_thr) *d_emit_thr = f->P.emit + L.cell_base;
  if (emit_bstride) *emit_bstride = (size_t)f->P.ncell_total;
  return SVS_OK;
}

// internal accessor for match.hip
FastView svs_fast_view_internal(const svs_fast *f) {
  FastView v{};
  v.n_levels = f->P.n_levels; v.emit = f->P.emit; v.ncell_total = f->P.ncell_total;
  for (int l = 0; l < f->P.n_levels; ++l) {
    const LevelDev &L = f->P.lv[l];
    v.bm[l] = reinterpret_cast<const uint8_t *>(L.bm); v.bm_stride[l] = L.bm_stride; v.bm_bstride[l] = L.bm_bstride; v.bm_gap[l] = 32 * L.bm_wpr - L.cell_w;
    v.cell_base[l] = L.cell_base; v.gx[l] = L.gx; v.gy[l] = L.gy; v.cell_w[l] = L.cell_w; v.cell_h[l] = L.cell_h;
    v.w[l] = L.w; v.h[l] = L.h;
  }
  return v;
}
