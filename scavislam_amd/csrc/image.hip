// image.hip -- context + preprocessing kernels (u8 pyramid step, f32 convert + Sobel).
// Replaces FrameGrabber::preprocessing (frame_grabber.cpp:285-336): cv::buildPyramid on u8 and
// the CPU path's convertTo(CV_32F,1/255.) + Sobel(ksize=1).  Integer path is bit-exact to the
// OpenCV 2.4.2 semantics restated in SURVEY.md A.2; HBM-bound stencils, LDS-tiled.
#include "common.h"
#include <algorithm>
#include <mutex>
#include <vector>
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
namespace {
// Per device, process-wide.  `last` = the event behind the latest launch that went THROUGH the gate, `last_cus` the compute units that launch may hold; `lanes` = the latest
// priority-lane launch of every other context (its event + its compute units).  All of it is touched under `m`, which a context holds from svs_spin_enter to
// svs_spin_leave (host side of one launch: microseconds -- nobody ever blocks on device work while holding it).
struct SpinLane { svs_ctx *c; hipEvent_t ev; int cus; };
struct SpinGate { std::mutex m; hipEvent_t last = nullptr; svs_ctx *owner = nullptr; int last_cus = 0; std::vector<SpinLane> lanes; };
SpinGate g_spin_gate[64];
SpinGate &gate_of(const svs_ctx *c) { return g_spin_gate[(unsigned)c->device % 64u]; }
}  // namespace
// The invariant the gate keeps: the workgroups of all kernels in flight whose workgroups wait for each other INSIDE the launch fit the device TOGETHER, counted at one
// whole compute unit per workgroup (the latency-mode trackers: 512 lanes at 180+ registers, nothing fits beside one on a CU; the tile-resident Cholesky: > 80 KB of LDS
// each).  Then every one of them becomes resident whatever the dispatcher does, and none can starve another.
//  * n_workgroups: the size of the launch that is about to be made (0: unknown -- the whole device).
//  * A launch of at most SVS_SPIN_SMALL workgroups may take the PRIORITY LANE: it neither waits for the gate nor closes it -- but only while the invariant holds WITH it:
//    its workgroups + those of the gated launch still in flight (event not yet fired) + those of the other contexts' lane launches still in flight <= #CUs.  (Rounds 4-5
//    assumed that; a 16-workgroup grid solve beside a full-device tracker did not fit the assumption.)  Otherwise it goes through the gate like everything else.
//  * A gated launch waits (stream-side) for the previous gated launch of another context, and for the other contexts' lane launches if it does not fit beside them.
// Why the lane exists: a real-time frame of the front end (one camera stream: 8 workgroups) queued behind a whole 5.8 ms grid solve of the back-end thread -- two
// threads, one GPU: p99 of the frame 6.1 ms; with the lane 0.66 ms (the tile solve leaves 60 of 256 CUs free for exactly this).
constexpr int SVS_SPIN_SMALL = 16;
int svs_spin_enter(svs_ctx *c, int n_workgroups) {
  SpinGate &g = gate_of(c);
  const int demand = n_workgroups > 0 ? std::min(n_workgroups, c->n_cu) : c->n_cu;
  g.m.lock();                    // held until svs_spin_leave: the order of the launches is the order of the chain
  // lane launches that have finished no longer count
  for (size_t i = 0; i < g.lanes.size();) {
    if (g.lanes[i].c != c && hipEventQuery(g.lanes[i].ev) == hipErrorNotReady) ++i;
    else if (g.lanes[i].c == c) ++i;                                   // my own: ordered behind by my stream (kept for the others to see; replaced at leave)
    else { g.lanes[i] = g.lanes.back(); g.lanes.pop_back(); }
  }
  int lanes_busy = 0;
  for (const SpinLane &l : g.lanes) if (l.c != c) lanes_busy += l.cus;
  const bool gated_busy = g.last && g.owner != c && hipEventQuery(g.last) == hipErrorNotReady;
  (void)hipGetLastError();       // (hipErrorNotReady is an answer, not an error to be found by the next SVS_LAUNCH_CHECK)
  c->spin_demand = demand;
  c->spin_lane = n_workgroups > 0 && n_workgroups <= SVS_SPIN_SMALL && demand + lanes_busy + (gated_busy ? g.last_cus : 0) <= c->n_cu;
  if (c->spin_lane) { ++c->spin_n_lane; return SVS_OK; }
  ++c->spin_n_gated;
  hipError_t e = hipSuccess;
  if (g.last && g.owner != c) e = hipStreamWaitEvent(c->stream, g.last, 0);
  if (demand + lanes_busy > c->n_cu)
    for (const SpinLane &l : g.lanes) if (l.c != c && e == hipSuccess) e = hipStreamWaitEvent(c->stream, l.ev, 0);
  if (e != hipSuccess) { g.m.unlock(); c->err = std::string("svs_spin_enter: hipStreamWaitEvent -> ") + hipGetErrorString(e); return SVS_ERR_HIP; }
  return SVS_OK;
}
int svs_spin_leave(svs_ctx *c) {
  SpinGate &g = gate_of(c);
  hipError_t e = hipSuccess;
  hipEvent_t &ev = c->spin_lane ? c->lane_ev : c->spin_ev;
  if (!ev) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(ev, c->stream);
  if (e == hipSuccess) {
    if (c->spin_lane) {
      bool found = false;
      for (SpinLane &l : g.lanes) if (l.c == c) { l.ev = ev; l.cus = c->spin_demand; found = true; }
      if (!found) g.lanes.push_back(SpinLane{c, ev, c->spin_demand});
    } else { g.last = ev; g.owner = c; g.last_cus = c->spin_demand; }
  }
  c->spin_lane = false;
  g.m.unlock();
  if (e != hipSuccess) { c->err = std::string("svs_spin_leave: ") + hipGetErrorString(e); return SVS_ERR_HIP; }
  return SVS_OK;
}
// a context that goes away takes its entries with it (its events are destroyed with it)
static void spin_forget(svs_ctx *c) {
  SpinGate &g = gate_of(c);
  std::lock_guard<std::mutex> lk(g.m);
  for (size_t i = 0; i < g.lanes.size();) { if (g.lanes[i].c == c) { g.lanes[i] = g.lanes.back(); g.lanes.pop_back(); } else ++i; }
  if (g.owner == c) { g.last = nullptr; g.owner = nullptr; g.last_cus = 0; }
}
extern "C" int svs_ctx_create(int device, void *hip_stream, svs_ctx **out) {
  if (!out) return SVS_ERR_INVALID;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SVS_ERR_NO_DEVICE;
  svs_ctx *c = new svs_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return SVS_ERR_NO_DEVICE; }
  { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) c->n_cu = cu; }
  if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SVS_ERR_HIP; }
    c->own_stream = true;
  }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { delete c; return SVS_ERR_HIP; }
  auto env_int = [](const char *name, int lo, int hi) { const char *e = getenv(name); if (!e) return 0; const int v = atoi(e); return v < lo ? lo : (v > hi ? hi : v); };
  c->trk_nwg = env_int("SVS_TRK_NWG", 1, 64);
  if (getenv("SVS_TRK_BALANCE")) c->trk_balance = atoi(getenv("SVS_TRK_BALANCE"));
  c->trk_regs = getenv("SVS_TRK_ONE_PER_CU") ? 1 : (getenv("SVS_TRK_TWO_PER_CU") ? 2 : 0);
  c->full_nwg = env_int("SVS_FULL_NWG", 1, 1024);
  *out = c;
  return SVS_OK;
}
extern "C" int svs_ctx_destroy(svs_ctx *c) {
  if (!c) return SVS_OK;
  (void)hipStreamSynchronize(c->stream);
  spin_forget(c);                                                   // (the stream is drained: nobody needs to wait for this context any more)
  if (c->spin_ev) (void)hipEventDestroy(c->spin_ev);
  if (c->lane_ev) (void)hipEventDestroy(c->lane_ev);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->match_scratch) (void)hipFree(c->match_scratch);
  if (c->seq_buf) (void)hipFree(c->seq_buf);
  if (c->seq_stats) (void)hipFree(c->seq_stats);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return SVS_OK;
}
int svs_ctx_scratch(svs_ctx *c, size_t bytes, void **out) {
  SVS_REQUIRE(c, c && out);
  if (c->scratch_bytes < bytes) {
    SVS_DEVICE(c);
    if (c->scratch) { SVS_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
    const size_t want = bytes + bytes / 2 + 4096;
    SVS_HIP(c, hipMalloc(&c->scratch, want));
    c->scratch_bytes = want;
  }
  *out = c->scratch;
  return SVS_OK;
}
int svs_ctx_match_scratch(svs_ctx *c, size_t bytes, void **out) {
  SVS_REQUIRE(c, c && out);
  if (c->match_scratch_bytes < bytes) {
    SVS_DEVICE(c);
    if (c->match_scratch) { SVS_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->match_scratch); c->match_scratch = nullptr; c->match_scratch_bytes = 0; }
    const size_t want = bytes + bytes / 2 + 4096;
    SVS_HIP(c, hipMalloc(&c->match_scratch, want));
    c->match_scratch_bytes = want;
  }
  *out = c->match_scratch;
  return SVS_OK;
}
extern "C" int svs_ctx_set_option(svs_ctx *c, const char *name, int value) {
  SVS_REQUIRE(c, c && name && value >= 0);
  const std::string n(name);
  if (n == "trk_nwg") c->trk_nwg = value > 64 ? 64 : value;
  else if (n == "trk_regs") c->trk_regs = value > 2 ? 0 : value;
  else if (n == "trk_balance") c->trk_balance = value < 0 || value > 2 ? 1 : value;
  else if (n == "full_nwg") c->full_nwg = value > 1024 ? 1024 : value;
  else if (n == "mo_legacy") c->mo_legacy = value != 0;
  else if (n == "match_legacy") c->match_legacy = (int)value;
  else if (n == "fe_overlap") c->fe_overlap = value != 0;
  else if (n == "fe_pipeline") c->fe_pipeline = value != 0;
  else if (n == "match_order") c->match_order = value != 0;
  else if (n == "fe_fuse_tail") c->fe_fuse_tail = value != 0;
  else if (n == "mo_spec") c->mo_spec = value != 0;
  else if (n == "xcd_swizzle") c->xcd_swizzle = value != 0;
  else if (n == "trk_flat") c->trk_flat = value != 0;
  else if (n == "trk_split") c->trk_split = value > 15 ? 15 : value;
  else if (n == "trk_cont_slots") c->trk_cont_slots = value > 4096 ? 4096 : value;
  else if (n == "trk_seq_chi2") c->trk_seq_chi2 = value != 0;
  else if (n == "trk_lazy_chi2") c->trk_lazy_chi2 = value > 2 ? 1 : value;      // (2: kernel A/B only -- the passes store their terms, the accept test stays on the f64 sums)
  else SVS_REQUIRE(c, !"unknown option");
  return SVS_OK;
}
extern "C" int svs_ctx_get_stat(svs_ctx *c, const char *name, long long *out) {
  SVS_REQUIRE(c, c && name && out);
  const std::string n(name);
  // the spin gate's book-keeping of this context (host-side counters): launches that took the priority lane / that went through the gate
  if (n == "spin_lane_launches") { *out = c->spin_n_lane; return SVS_OK; }
  if (n == "spin_gated_launches") { *out = c->spin_n_gated; return SVS_OK; }
  SVS_REQUIRE(c, n == "trk_exact_sums" || n == "trk_exact_fallbacks");
  unsigned v[2] = {0, 0};
  if (c->seq_stats) {
    SVS_DEVICE(c);
    SVS_HIP(c, hipMemcpyAsync(v, c->seq_stats, sizeof v, hipMemcpyDeviceToHost, c->stream));
    SVS_HIP(c, hipStreamSynchronize(c->stream));
  }
  *out = n == "trk_exact_sums" ? v[0] : v[1];
  return SVS_OK;
}
extern "C" int svs_api_version(void) { return SVS_API_VERSION; }
extern "C" void svs_pose_opt_params_default(svs_pose_opt_params *p) {
  if (!p) return;
  __builtin_memset(p, 0, sizeof *p);
  p->robust_kernel = 1; p->num_iter = 15; p->kernel_param = 2.0; p->initial_mu = -1.0; p->tau = 1e-5; p->min_obs = 0;
}
extern "C" int svs_ctx_sync(svs_ctx *c) { SVS_REQUIRE(c, c); SVS_HIP(c, hipStreamSynchronize(c->stream)); return SVS_OK; }
extern "C" void *svs_ctx_stream(svs_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" const char *svs_last_error(svs_ctx *c) { return c ? c->err.c_str() : "null ctx"; }
extern "C" int svs_malloc(svs_ctx *c, size_t bytes, void **p) {
  SVS_REQUIRE(c, c && p);
  SVS_HIP(c, hipSetDevice(c->device));
  SVS_HIP(c, hipMalloc(p, bytes ? bytes : 1));
  return SVS_OK;
}
extern "C" int svs_free(svs_ctx *c, void *p) { SVS_REQUIRE(c, c); if (p) SVS_HIP(c, hipFree(p)); return SVS_OK; }
extern "C" int svs_memcpy_h2d(svs_ctx *c, void *d, const void *h, size_t n) {
  SVS_REQUIRE(c, c);
  SVS_DEVICE(c);
  SVS_HIP(c, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->stream));
  SVS_HIP(c, hipStreamSynchronize(c->stream));  // h may be pageable and reused by the caller
  return SVS_OK;
}
extern "C" int svs_memcpy_d2h(svs_ctx *c, void *h, const void *d, size_t n) {
  SVS_REQUIRE(c, c);
  SVS_DEVICE(c);
  SVS_HIP(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream));
  SVS_HIP(c, hipStreamSynchronize(c->stream));
  return SVS_OK;
}
extern "C" int svs_timer_start(svs_ctx *c) { SVS_REQUIRE(c, c);
  SVS_DEVICE(c); SVS_HIP(c, hipEventRecord(c->ev0, c->stream)); return SVS_OK; }
extern "C" int svs_timer_stop_ms(svs_ctx *c, float *ms) {
  SVS_REQUIRE(c, c && ms);
  SVS_DEVICE(c);
  SVS_HIP(c, hipEventRecord(c->ev1, c->stream));
  SVS_HIP(c, hipEventSynchronize(c->ev1));
  SVS_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return SVS_OK;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) { i = i < 0 ? -i : 2 * n - 2 - i; }
  return i;
}

// One pyrDown step, no LDS: a lane produces a 4 x 2 patch of output pixels from a 7-row x 11-byte source footprint
// held in registers (three unaligned dword loads per row), horizontal [1 4 6 4 1] per row with V_DOT4_U32_U8, the two
// vertical combinations on packed u16 pairs; `(s+128)>>8`; one dword store per output row.
__device__ __forceinline__ uint32_t ld_u32u(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
// COPY: the lane also stores the 8 x 4 source pixels that only it covers (columns 2 ox .. 2 ox + 7, rows 2 oy .. 2 oy + 3: exactly the byte-aligned dwords
// d01 / d12 of its four middle rows) into `cpy` -- the front end takes a caller's device frame into its own level-0 buffer while it builds level 1,
// instead of with a copy kernel of its own (one read of the frame instead of two).
template <bool COPY>
__global__ __launch_bounds__(256) void pyr_down_u8_kernel(const uint8_t *__restrict__ src, int w, int h, int ss,
                                                         size_t sb, uint8_t *__restrict__ dst, int dw, int dh,
                                                         int ds, size_t db, uint8_t *__restrict__ cpy, int cs, size_t cb, int swz) {
  // work items (4 x 2 output patches) are dealt to the lanes in row-major order over the whole image: every lane of every wave but the last has a patch (a 64 x 4 block of
  // lanes per 256 x 8 outputs left 3 of 8 lanes idle on a 320-column level).  Vertical neighbours share three of their source rows: blocks in XCD-contiguous order (an
  // XCD works through whole images), so those rows are fetched into one L2
  unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
  if (swz) wg = xcd_contiguous(wg, gridDim.x * gridDim.y);
  const int ncol = (dw + 3) >> 2, nrow = (dh + 1) >> 1;
  const int bz = wg / gridDim.x, item = (int)(wg % gridDim.x) * 256 + (int)threadIdx.x;
  if (item >= ncol * nrow) return;
  const int oyh = item / ncol;
  const int ox = 4 * (item - oyh * ncol), oy = 2 * oyh;
  src += (size_t)bz * sb;
  dst += (size_t)bz * db;
  if (COPY) cpy += (size_t)bz * cb;
  const int sx0 = 2 * ox - 2, sy0 = 2 * oy - 2;
  // Column borders without divergence: the leftmost lane (sx0 = -2) and the lane whose third dword crosses the right
  // edge load from a clamped in-row address and rebuild their bytes with V_PERM (BORDER_REFLECT_101); rows are reflected
  // per wave (uniform).  Only footprints whose first two dwords cross the right edge (widths not a multiple of 8) take
  // the per-byte path.
  const bool left = sx0 < 0, cross2 = sx0 + 12 > w;
  const bool fast = w >= 16 && sx0 + 8 <= w;
  uint32_t sel0 = 0x03020100u, sel2 = 0x03020100u;
  int off2 = sx0 + 8;
  if (left) sel0 = 0x01000102u;                                 // cols -2,-1,0,1 -> 2,1,0,1 of the dword loaded at col 0
  if (cross2) {
    off2 = w - 4;
    sel2 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = sx0 + 8 + k, cr = c < w ? c : 2 * w - 2 - c;
      sel2 |= (uint32_t)max(min(cr - (w - 4), 3), 0) << (8 * k);
    }
  }
  typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
  union Pk2 { uint32_t u; us2_t h; };
  Pk2 hp[7][2];      // hsum[r][0..1], hsum[r][2..3] as packed u16 (max 16 * 255 = 4080)
  const uint32_t W4 = 1u | (4u << 8) | (6u << 16) | (4u << 24);
  // the 21 dwords of the footprint are requested up front where the whole wave is on the dword path (every interior wave): one round trip per wave, not one per row
  // -- with the row loads inside the per-row `if (fast)` the compiler waited for each row before it asked for the next, and the kernel ran at a third of the copy rate
  uint32_t D0[7], D1[7], D2[7];
  if (h >= 8 && __all(fast)) {
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      int y = sy0 + r;                                          // in [-2, h + 3]: one reflection is enough
      y = y < 0 ? -y : y;
      y = y >= h ? 2 * h - 2 - y : y;
      const uint8_t *p = src + (size_t)y * ss;
      D0[r] = ld_u32u(p + (left ? 0 : sx0)); D1[r] = ld_u32u(p + sx0 + 4); D2[r] = ld_u32u(p + off2);
    }
  } else {
#pragma unroll 1
    for (int r = 0; r < 7; ++r) {
      const uint8_t *p = src + (size_t)reflect101(sy0 + r, h) * ss;
      uint32_t d0, d1, d2;
      if (fast) {
        d0 = __builtin_amdgcn_perm(0u, ld_u32u(p + (left ? 0 : sx0)), sel0);
        d1 = ld_u32u(p + sx0 + 4);
        d2 = __builtin_amdgcn_perm(0u, ld_u32u(p + off2), sel2);
      } else {
        uint32_t bt[12];
#pragma unroll
        for (int k = 0; k < 11; ++k) bt[k] = p[reflect101(sx0 + k, w)];
        bt[11] = 0;
        d0 = bt[0] | (bt[1] << 8) | (bt[2] << 16) | (bt[3] << 24);
        d1 = bt[4] | (bt[5] << 8) | (bt[6] << 16) | (bt[7] << 24);
        d2 = bt[8] | (bt[9] << 8) | (bt[10] << 16);
      }
      // (dynamic index: this path is the image border's and tiny images')
      if (r == 0) { D0[0] = d0; D1[0] = d1; D2[0] = d2; } else if (r == 1) { D0[1] = d0; D1[1] = d1; D2[1] = d2; } else if (r == 2) { D0[2] = d0; D1[2] = d1; D2[2] = d2; }
      else if (r == 3) { D0[3] = d0; D1[3] = d1; D2[3] = d2; } else if (r == 4) { D0[4] = d0; D1[4] = d1; D2[4] = d2; } else if (r == 5) { D0[5] = d0; D1[5] = d1; D2[5] = d2; }
      else { D0[6] = d0; D1[6] = d1; D2[6] = d2; }
    }
    sel0 = 0x03020100u; sel2 = 0x03020100u;                     // already in place
  }
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const uint32_t d0 = __builtin_amdgcn_perm(0u, D0[r], sel0), d1 = D1[r], d2 = __builtin_amdgcn_perm(0u, D2[r], sel2);
    // horizontal pass: out_j = [1 4 6 4] . bytes[2j..2j+3] + bytes[2j+4]  -> V_DOT4_U32_U8 on byte-aligned dwords
    const uint32_t d01 = __builtin_amdgcn_alignbyte(d1, d0, 2);      // bytes 2..5
    const uint32_t d12 = __builtin_amdgcn_alignbyte(d2, d1, 2);      // bytes 6..9
    const uint32_t h0 = __builtin_amdgcn_udot4(d0, W4, d1 & 0xffu, false);              // + byte 4
    const uint32_t h1 = __builtin_amdgcn_udot4(d01, W4, d12 & 0xffu, false);            // + byte 6
    const uint32_t h2 = __builtin_amdgcn_udot4(d1, W4, d2 & 0xffu, false);              // + byte 8
    const uint32_t h3 = __builtin_amdgcn_udot4(d12, W4, (d2 >> 16) & 0xffu, false);     // + byte 10
    hp[r][0].u = h0 | (h1 << 16);
    hp[r][1].u = h2 | (h3 << 16);
    if (COPY && r >= 2 && r <= 5) {
      const int cy = 2 * oy + r - 2, cx = 2 * ox;
      if (cy < h) {
        uint8_t *q = cpy + (size_t)cy * cs + cx;
        if (cx + 7 < w && (cs & 3) == 0 && (cb & 3) == 0 && ((uintptr_t)cpy & 3) == 0) {
          reinterpret_cast<uint32_t *>(q)[0] = d01;
          reinterpret_cast<uint32_t *>(q)[1] = d12;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (cx + j < w) q[j] = (uint8_t)((j < 4 ? d01 : d12) >> (8 * (j & 3)));
        }
      }
    }
  }
  // vertical pass on packed pairs: (h0 + h4) + 4 (h1 + h3) + 6 h2 <= 16 * 4080 = 65280 fits u16; (s + 128) >> 8
  const us2_t c4 = {4, 4}, c6 = {6, 6}, c128 = {128, 128};
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    if (oy + o >= dh) break;
    uint32_t v[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const us2_t acc = (hp[2 * o][q].h + hp[2 * o + 4][q].h) + c4 * (hp[2 * o + 1][q].h + hp[2 * o + 3][q].h) + c6 * hp[2 * o + 2][q].h;
      // the 17-bit sum of the last +128 cannot be formed in u16: (acc >> 8) + ((acc & 255) >= 128)
      Pk2 t;
      t.h = acc;
      v[2 * q] = ((t.u & 0xffffu) + 128u) >> 8;
      v[2 * q + 1] = ((t.u >> 16) + 128u) >> 8;
    }
    (void)c128;
    const uint32_t packed = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    uint8_t *q = dst + (size_t)(oy + o) * ds + ox;
    if (ox + 3 < dw && ((ds | (int)(db & 3)) & 3) == 0 && ((uintptr_t)dst & 3) == 0) *reinterpret_cast<uint32_t *>(q) = packed;
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (ox + j < dw) q[j] = (uint8_t)v[j];
    }
  }
}

extern "C" int svs_pyr_down_u8(svs_ctx *ctx, const uint8_t *d_src, int w, int h, int sstride, size_t s_bstride,
                               uint8_t *d_dst, int dstride, size_t d_bstride, int batch) {
  SVS_REQUIRE(ctx, ctx && d_src && d_dst && w >= 3 && h >= 3 && batch >= 1);
  SVS_DEVICE(ctx);
  int dw = (w + 1) / 2, dh = (h + 1) / 2;
  dim3 grid(div_up(div_up(dw, 4) * div_up(dh, 2), 256), batch), block(256);
  hipLaunchKernelGGL(pyr_down_u8_kernel<false>, grid, block, 0, ctx->stream, d_src, w, h, sstride, s_bstride, d_dst, dw,
                     dh, dstride, d_bstride, (uint8_t *)nullptr, 0, (size_t)0, ctx->xcd_swizzle);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
// internal (frontend.hip): one pyrDown step that also copies the source image into d_copy (same size as the source)
int svs_pyr_down_u8_copy(svs_ctx *ctx, const uint8_t *d_src, int w, int h, int sstride, size_t s_bstride, uint8_t *d_dst, int dstride, size_t d_bstride,
                         uint8_t *d_copy, int cstride, size_t c_bstride, int batch) {
  SVS_REQUIRE(ctx, ctx && d_src && d_dst && d_copy && w >= 3 && h >= 3 && batch >= 1);
  SVS_DEVICE(ctx);
  int dw = (w + 1) / 2, dh = (h + 1) / 2;
  dim3 grid(div_up(div_up(dw, 4) * div_up(dh, 2), 256), batch), block(256);
  hipLaunchKernelGGL(pyr_down_u8_kernel<true>, grid, block, 0, ctx->stream, d_src, w, h, sstride, s_bstride, d_dst, dw, dh, dstride, d_bstride, d_copy,
                     cstride, c_bstride, ctx->xcd_swizzle);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}

// convertTo(CV_32F, 1/255.) then Sobel(ksize=1): dx = I(x+1)-I(x-1), dy = I(y+1)-I(y-1),
// BORDER_REFLECT_101 (=> 0 on the border).  One thread per 4 consecutive pixels: 5 u8 reads
// per row neighbourhood, 3 x float4 coalesced stores.
__global__ __launch_bounds__(256) void convert_sobel_kernel(const uint8_t *__restrict__ src, int w, int h, int ss,
                                                            size_t sb, float *__restrict__ img, float *__restrict__ dx,
                                                            float *__restrict__ dy, int fs, size_t fb) {
  const float sc = (float)(1. / 255.);
  int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  int y = blockIdx.y;
  if (x4 >= w) return;
  src += (size_t)blockIdx.z * sb;
  size_t fo = (size_t)blockIdx.z * fb + (size_t)y * fs;
  const uint8_t *row = src + (size_t)y * ss;
  const uint8_t *rowm = src + (size_t)reflect101(y - 1, h) * ss;
  const uint8_t *rowp = src + (size_t)reflect101(y + 1, h) * ss;
  float c[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) c[k] = (float)row[reflect101(x4 - 1 + k, w)] * sc;
  if (x4 + 3 < w && (fs & 3) == 0 && (fb & 3) == 0 && (((uintptr_t)img | (uintptr_t)dx | (uintptr_t)dy) & 15) == 0) {
    // full group of 4: three 16-byte stores per lane (1 KiB per wave-instruction)
    float4 vi = make_float4(c[1], c[2], c[3], c[4]);
    float4 vx = make_float4(c[2] - c[0], c[3] - c[1], c[4] - c[2], c[5] - c[3]);
    float4 vy;
    vy.x = (float)rowp[x4] * sc - (float)rowm[x4] * sc;
    vy.y = (float)rowp[x4 + 1] * sc - (float)rowm[x4 + 1] * sc;
    vy.z = (float)rowp[x4 + 2] * sc - (float)rowm[x4 + 2] * sc;
    vy.w = (float)rowp[x4 + 3] * sc - (float)rowm[x4 + 3] * sc;
    *reinterpret_cast<float4 *>(img + fo + x4) = vi;
    *reinterpret_cast<float4 *>(dx + fo + x4) = vx;
    *reinterpret_cast<float4 *>(dy + fo + x4) = vy;
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int x = x4 + k;
    if (x < w) {
      img[fo + x] = c[k + 1];
      dx[fo + x] = c[k + 2] - c[k];
      dy[fo + x] = (float)rowp[x] * sc - (float)rowm[x] * sc;
    }
  }
}

extern "C" int svs_convert_sobel_f32(svs_ctx *ctx, const uint8_t *d_src, int w, int h, int sstride, size_t s_bstride,
                                     float *d_img, float *d_dx, float *d_dy, int fstride, size_t f_bstride,
                                     int batch) {
  SVS_REQUIRE(ctx, ctx && d_src && d_img && d_dx && d_dy && w >= 2 && h >= 2 && batch >= 1);
  SVS_DEVICE(ctx);
  dim3 block(64), grid(div_up(div_up(w, 4), 64), h, batch);
  hipLaunchKernelGGL(convert_sobel_kernel, grid, block, 0, ctx->stream, d_src, w, h, sstride, s_bstride, d_img,
                     d_dx, d_dy, fstride, f_bstride);
  SVS_LAUNCH_CHECK(ctx);
  return SVS_OK;
}
