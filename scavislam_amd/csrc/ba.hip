// ba.hip -- double-window bundle adjustment (Schur-complement LM) for gfx950.
// Replaces SlamGraph::optimize (slam_graph.cpp:312-355): the g2o pipeline BlockSolver_6_3 +
// OptimizationAlgorithmLevenberg + RobustKernelHuber + sparse Cholesky, with the edge
// arithmetic of g2o_types/anchored_points.cpp:148-235 and transformations.h:62-95.
//
// MI355X-first design (f64 throughout; no MFMA: blocks are 3x3 / 6x3 / 6x6):
//  * edges are sorted by (landmark, observer) and packed into chunks of <= 64 whole landmarks'
//    edges; ONE LANE PER EDGE, one wavefront per chunk.  Everything a landmark needs from its
//    edges (H_ll, b_l, the anchor's W and H_pp block) is a SEGMENTED wave64 shuffle reduction
//    over the landmark's lanes -- no LDS, no per-landmark scratch in HBM;
//  * the 3x3 block inverse and all 6x6 Schur outer products W_i D^-1 W_j^T are formed in
//    registers (pairs of a landmark = pairs of lanes of its segment, partner data by shuffle)
//    and go straight into the packed upper-block reduced camera system with hardware f64
//    atomics (global_atomic_add_f64); H_pl is never materialised;
//  * back-substitution re-linearises instead of reloading stored W blocks (64 B/edge read
//    instead of 144+ B/edge) and fuses the trial-state chi2 into the same launch;
//  * landmark shards (multi-GPU) only exchange the packed reduced system (one all-reduce of
//    36*P(P+1)/2 + 12P + 1 doubles) and two scalars per LM trial; the 6P x 6P Cholesky is
//    replicated (deterministic, no broadcast).
#include "common.h"
#include <climits>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

#include "ba_schur.inc"
#include "ba_solve.inc"
#include "ba_solve_tiles.inc"
// expands the packed upper blocks to a full symmetric matrix + bred (parity tests)
__global__ void ba_expand_kernel(BaDev B, double *__restrict__ Hfull, double *__restrict__ bred) {
  const int P = B.P, n = 6 * P;
  const long total = (long)n * n;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / n), col = (int)(e % n);
    int bi = row / 6, bj = col / 6, r = row % 6, c = col % 6;
    double v;
    if (bi < bj) v = B.H[blk_index(bi, bj, P) * 36 + 6 * r + c];
    else if (bi > bj) v = B.H[blk_index(bj, bi, P) * 36 + 6 * c + r];
    else v = r <= c ? B.H[blk_index(bi, bi, P) * 36 + 6 * r + c] : B.H[blk_index(bi, bi, P) * 36 + 6 * c + r];
    if (row == col) v += B.lambda;
    Hfull[e] = v;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bred[i] = B.bp[i] - B.bs[i];
}

// Accept / reject of one LM trial on the device (OptimizationAlgorithmLevenberg::solve), used by the
// speculative enqueue of svs_ba_optimize: on acceptance the next trial's lambda is published in ctl[0]; anything else
// (rejection, rho == 0, non-finite chi2) raises the abort flag, the already enqueued kernels of later trials return
// immediately and the host takes over with its ordinary loop.  Record per iteration: chi2 at the current state, trial
// chi2, scale, rho, accepted, lambda after the update, solver failure, done marker.
__global__ void ba_lm_kernel(BaDev B, int it) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || B.ctl[1] != 0.0) return;
  ba_lm_decide(B, it);
}

}  // namespace

// Small host-side worker pool for the per-call marshalling (svs_ba_set_problem): the phases are passes over the 64-byte
// edge records and are memory-bound on one core (~0.25 ms per pass at 100k edges).  Workers spin briefly between the
// back-to-back phases of one call and sleep on a condition variable between calls.
class HostPool {
 public:
  explicit HostPool(int n) : n_(std::max(1, n)) {
    for (int t = 1; t < n_; ++t) th_.emplace_back([this, t] { loop(t); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_.store(true); gen_fast_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  int size() const { return n_; }
  // fn(t, n) on every worker t in [0, n); returns when all are done
  void run(const std::function<void(int, int)> &fn) {
    if (n_ == 1) { fn(0, 1); return; }
    job_ = &fn;
    pending_.store(n_ - 1, std::memory_order_relaxed);
    { std::lock_guard<std::mutex> lk(m_); gen_fast_.fetch_add(1, std::memory_order_release); }      // under the lock: no lost wake-up
    cv_.notify_all();
    fn(0, n_);
    for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; ++spin)       // phases are sub-millisecond: spin briefly, then let go of the core
      if (spin > 4096) std::this_thread::yield();
  }
  std::mutex use_;                      // one caller at a time (the pool is shared by every optimizer of the process)

 private:
  void loop(int t) {
    unsigned seen = 0;
    for (;;) {
      bool got = false;
      for (int spin = 0; spin < 20000 && !got; ++spin) got = gen_fast_.load(std::memory_order_acquire) != seen;     // ~100 us of polling
      if (!got) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_.load() || gen_fast_.load(std::memory_order_acquire) != seen; });
      }
      if (stop_.load()) return;
      seen = gen_fast_.load(std::memory_order_acquire);
      (*job_)(t, n_);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  const std::function<void(int, int)> *job_ = nullptr;
  std::atomic<unsigned> gen_fast_{0};
  std::atomic<int> pending_{0};
  std::atomic<bool> stop_{false};
};

struct BaOptions {                      // experiment / test switches, latched at svs_ba_create (never read from the environment per call)
  int no_order = 0;      // "no_order": keep the caller's pose order in the solve (A/B partner of the fill-reducing order)
  int no_graph = 0;          // "no_graph": the speculative trials of an optimize are enqueued kernel by kernel (rounds 1-5) instead of replayed from a HIP graph
  int no_lds_panel = 0;      // "no_lds_panel": the fused solve's back-substitution panel through global memory (rounds 3-5), A/B only
  int no_speculation = 0, one_front = 0, no_fused_solve = 0, no_lds_solve = 0, no_fused_cons = 0, no_grid_solve = 0, no_tile_solve = 0, debug = 0;
  int nw = 0, nw4 = 0, p1 = -1, group = 0, host_threads = 0, grid_g = 0;
  int host_marshal = 0;                 // svs_ba_set_problem: 0 = by size (device route from 30k edges), 1 = always on the host (rounds 1-2), 2 = always on the device
};
int svs_comm_allreduce_hook(void *d_buf, size_t count, void *user);      // comm.hip
// waves per workgroup of the Schur kernel: the smallest of 4..8 that gets the grid down to one workgroup per CU (if any does)
static inline int pick_nw(int n_chunks, int n_cu, const BaOptions &opt) {
  int nw = 4;
  if (!opt.nw4) for (int c = 5; c <= 8 && (n_chunks + nw - 1) / nw > n_cu; ++c) if ((n_chunks + c - 1) / c <= n_cu) nw = c;
  if (opt.nw >= 4) nw = opt.nw;      // 4..8 (clamped where it is set)
  return nw;
}

struct svs_ba {
  svs_ctx *ctx = nullptr;
  BaOptions opt;
  svs_comm *comm = nullptr;             // library-owned collective of sharded runs (svs_ba_set_comm)
  bool problem_valid = false;           // set by a COMPLETED svs_ba_set_problem / svs_ba_window_update; every other entry point requires it
  // ---- persistent window (svs_ba_window_*): every observation handed over since the last reset stays on the device ----
  svs_ba_edge *w_store = nullptr; size_t w_n = 0, w_cap = 0;           // observation store, ids in .point / .pose, arrival order
  int *w_pose_tab = nullptr, *w_point_tab = nullptr; size_t w_pose_tab_n = 0, w_point_tab_n = 0;      // id -> window index (-1: not in the window)
  void *w_work = nullptr; size_t w_work_bytes = 0;                      // per-call work arrays (grow-only)
  std::vector<svs_ba_constraint> w_cons_idx;
  void *w_in = nullptr; size_t w_in_bytes = 0;                          // device mirror of the block a window_update call brings (ids, state, new observations, constraints)
  unsigned char *w_hback = nullptr; size_t w_hback_bytes = 0;           // pinned read-back (counters, landmark lengths, pattern)
  int P = 0, L = 0, E = 0, C = 0, n_chunks = 0, n_wide = 0, add_pose_terms = 1;
  std::vector<int> w_chunk_nlm;                       // landmarks per wave chunk (set_problem work vector)
  int nw_sched = 0;                                   // waves per Schur workgroup the chunk list was laid out for (0: chosen at launch)
  svs_cam cam{};
  svs_ba_params prm{};
  double *d_poses[2] = {nullptr, nullptr}, *d_psi[2] = {nullptr, nullptr};
  int cur = 0;
  svs_ba_edge *d_edges = nullptr;
  int *d_chunk_start = nullptr, *d_chunk_len = nullptr;
  svs_ba_constraint *d_cons = nullptr;
  double *d_red = nullptr;     // [nblk*36][bp 6P][bs 6P][chi2]
  size_t red_count = 0;
  double *d_x = nullptr, *d_scal = nullptr, *d_linv = nullptr;
  int *d_rowmax = nullptr, *d_colmin = nullptr;
  double *d_upanel = nullptr;           // [P][R][36] panel rows of the LDS-window solve
  int env_R = 0;                        // max envelope row length + 1 (0 = unknown)
  // fill-reducing order of the pose blocks for the solve (what LinearSolverCSparse's block ordering does for the reference, slam_graph.cpp:1063-1074):
  // solver row k <-> pose perm[k].  The Schur kernels keep writing the system in the caller's pose order; when the order pays, the solve runs on a permuted
  // copy (ba_permute_system_kernel) and its x / trial poses are scattered back (ba_unpermute_kernel).  env_R_natural = the envelope without it.
  bool perm_active = false; int env_R_natural = 0;
  int *d_perm = nullptr; double *d_perm_sys = nullptr; size_t cap_perm_sys = 0;      // [P]; H' + bp' + bs' + poses' + trial poses' + x'
  std::vector<int> h_perm;
  bool use_lds_solve = false, use_fused_solve = false; size_t lds_solve_smem = 0;
  bool timing = false;                         // hipEvent brackets around the three dominant kernels of every trial (svs_ba_set_timing / svs_ba_kernel_times)
  int fuse_lds_panel = 0;                      // the fused solve keeps its back-substitution panel in LDS (ensure_profile)
  int fuse_P1 = 0;    // two-front fused solve: rows of the reversed front (0 = single front), launch counter for its flags
  int *d_rowmax2 = nullptr; size_t cap_rowmax2 = 0; double *d_xfer = nullptr; unsigned *d_flags = nullptr;
  unsigned *d_gridbar = nullptr; int grid_G = 0;      // multi-workgroup solve: arrival counter + failure flag, number of workgroups (0 = not used)
  double *d_tilews = nullptr; size_t cap_tilews = 0; int tiles_G = 0, tiles_S = 0, tiles_tpw = 0, tiles_gq = 0, tiles_ncm = 0;      // tile-resident variant of it (ba_solve_tiles.inc): workspace, grid, tile rows, own tiles per workgroup (tiles_G = 0: not used)
  // The all-accepted optimize of a resident window has a FIXED launch topology -- control upload, num_iters x (clear, Schur, [permute,] solve, [unpermute,] back-substitute,
  // decide), control read-back: ~25 stream operations, ~5 us of host time each, i.e. the ceiling of windows in flight (6 k windows / s in bench.py where the kernels
  // would allow 10 k).  It is captured ONCE per (problem layout, current state buffer) and replayed with one hipGraphLaunch (slam_graph.cpp:312-355 = one call).
  struct Graph { hipGraphExec_t exec = nullptr; uint64_t sig = 0, seen = 0; };      // seen: the signature of the last call that ran kernel by kernel
  Graph graph[2];                              // by ba->cur at the start of the call
  long long n_graph_launches = 0, n_graph_captures = 0;
  double *d_ctl = nullptr, *h_ctl = nullptr;   // LM control block of the speculative path (device + pinned host mirror)
  int ctl_iters = 0;
  std::vector<hipEvent_t> spec_ev;      // 6 events per speculative trial
  double *h_scal = nullptr;             // pinned host mirror of d_scal (the per-trial read-back must not go through a pageable staging copy)
  double *d_pattern = nullptr;          // [P*P] structural indicator (all-reduced in sharded runs)
  std::vector<double> h_pattern;
  // persistent host work arrays / pinned staging of set_problem, device capacities (grow-only)
  std::vector<int> w_anchor, w_nobs, w_pos, w_off, w_aoff, w_alist, w_order, w_cs, w_cl;
  std::vector<int> w_cnt;                      // [workers][L] per-worker landmark counts -> start offsets
  std::vector<uint64_t> w_keys, w_ent;         // per edge: (point, pose) / per slot: (pose, source index)
  unsigned char *h_stage = nullptr; size_t h_stage_cap = 0, h_stage_used = 0;      // pinned staging of the small per-call uploads
  unsigned char *h_state = nullptr; size_t h_state_cap = 0;                         // pinned landing area of svs_ba_get_state
  std::vector<int32_t> w_ids_p, w_ids_l, w_ids_a;      // svs_ba_set_problem's device route: identity ids, anchors by point
  HostPool *pool = nullptr;                    // marshalling workers: ONE pool per process, shared by all optimizers (created on first use)
  std::vector<double> w_pat_local;
  svs_ba_edge *h_edges = nullptr; size_t h_edges_cap = 0;
  size_t cap_poses[2] = {0, 0}, cap_psi[2] = {0, 0}, cap_edges = 0, cap_cs = 0, cap_cl = 0, cap_cons = 0, cap_red = 0, cap_x = 0, cap_scal = 0,
         cap_linv = 0, cap_rowmax = 0, cap_colmin = 0, cap_pattern = 0, cap_upanel = 0;
  bool profile_ready = false;
  hipEvent_t ev[6] = {};
  float t_reduce = 0, t_solve = 0, t_backsub = 0;
  int n_reduce = 0;
  void free_all() {
    for (int i = 0; i < 2; ++i) { if (d_poses[i]) (void)hipFree(d_poses[i]); if (d_psi[i]) (void)hipFree(d_psi[i]); d_poses[i] = d_psi[i] = nullptr; }
    if (d_edges) (void)hipFree(d_edges); if (d_chunk_start) (void)hipFree(d_chunk_start); if (d_chunk_len) (void)hipFree(d_chunk_len);
    if (d_cons) (void)hipFree(d_cons); if (d_red) (void)hipFree(d_red); if (d_x) (void)hipFree(d_x); if (d_scal) (void)hipFree(d_scal);
    if (d_linv) (void)hipFree(d_linv);
    if (d_rowmax) (void)hipFree(d_rowmax); if (d_colmin) (void)hipFree(d_colmin); if (d_pattern) (void)hipFree(d_pattern);
    if (d_upanel) (void)hipFree(d_upanel); d_upanel = nullptr; env_R = 0; use_lds_solve = false;
    d_rowmax = d_colmin = nullptr; d_pattern = nullptr; profile_ready = false;
    d_edges = nullptr; d_chunk_start = d_chunk_len = nullptr; d_cons = nullptr; d_red = d_x = d_scal = d_linv = nullptr;
    cap_poses[0] = cap_poses[1] = cap_psi[0] = cap_psi[1] = cap_edges = cap_cs = cap_cl = cap_cons = cap_red = cap_x = cap_scal = cap_linv = cap_rowmax =
        cap_colmin = cap_pattern = cap_upanel = 0;
    if (h_edges) { (void)hipHostFree(h_edges); h_edges = nullptr; h_edges_cap = 0; }
  }
};

static BaDev make_dev(const svs_ba *ba, double lambda, int cur = -1, double *ctl = nullptr) {
  BaDev B{};
  if (cur < 0) cur = ba->cur;
  B.ctl = ctl;
  B.P = ba->P; B.L = ba->L; B.E = ba->E; B.C = ba->add_pose_terms ? ba->C : 0; B.n_chunks = ba->n_chunks;
  B.poses = ba->d_poses[cur]; B.psi = ba->d_psi[cur];
  B.poses_trial = ba->d_poses[1 - cur]; B.psi_trial = ba->d_psi[1 - cur];
  B.edges = ba->d_edges; B.chunk_start = ba->d_chunk_start; B.chunk_len = ba->d_chunk_len; B.cons = ba->d_cons;
  const size_t nblk = (size_t)ba->P * (ba->P + 1) / 2;
  B.H = ba->d_red; B.bp = ba->d_red + nblk * 36; B.bs = B.bp + 6 * (size_t)ba->P; B.chi2_cur = B.bs + 6 * (size_t)ba->P;
  B.x = ba->d_x; B.scal = ba->d_scal;
  B.cam = ba->cam; B.delta = ba->prm.huber_delta; B.lambda = lambda; B.robust = ba->prm.use_robust; B.self_mode = ba->prm.self_edge_mode;
  B.fuse_cons = (B.C > 0 && B.n_chunks > 0 && !ba->opt.no_fused_cons) ? 1 : 0;
  B.n_wide = ba->n_wide; B.wide_start = ba->d_chunk_start + ba->n_chunks; B.wide_len = ba->d_chunk_len + ba->n_chunks;
  B.wide_split = 16;
  return B;
}

// Small uploads go through one pinned staging area: an asynchronous copy from pageable memory makes the runtime wait for
// the stream (it has to reuse its own bounce buffer), which would serialise the host behind the big edge DMA.  The area is
// reset by svs_ba_set_problem after its initial stream synchronisation.
static int stage_reserve(svs_ba *ba, size_t bytes) {
  svs_ctx *ctx = ba->ctx;
  if (bytes <= ba->h_stage_cap) return SVS_OK;
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ba->h_stage) (void)hipHostFree(ba->h_stage);
  ba->h_stage = nullptr; ba->h_stage_cap = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  SVS_HIP(ctx, hipHostMalloc((void **)&ba->h_stage, want, hipHostMallocDefault));
  ba->h_stage_cap = want;
  return SVS_OK;
}
static int stage_upload(svs_ba *ba, void *d_dst, const void *h_src, size_t bytes) {
  svs_ctx *ctx = ba->ctx;
  if (bytes == 0) return SVS_OK;
  const size_t off = (ba->h_stage_used + 63) & ~(size_t)63;
  if (off + bytes > ba->h_stage_cap) {      // should not happen (reserved up front): fall back to the pageable copy
    SVS_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SVS_OK;
  }
  ba->h_stage_used = off + bytes;
  // big arrays go in pieces: the DMA of a piece runs while the next one is copied into the pinned area
  const size_t piece = bytes > (1u << 20) ? (size_t)1 << 19 : bytes;
  for (size_t o = 0; o < bytes; o += piece) {
    const size_t nb = std::min(piece, bytes - o);
    std::memcpy(ba->h_stage + off + o, static_cast<const char *>(h_src) + o, nb);
    SVS_HIP(ctx, hipMemcpyAsync(static_cast<char *>(d_dst) + o, ba->h_stage + off + o, nb, hipMemcpyHostToDevice, ctx->stream));
  }
  return SVS_OK;
}

extern "C" int svs_ba_create(svs_ctx *ctx, svs_ba **out) {
  SVS_REQUIRE(ctx, ctx && out);
  svs_ba *ba = new svs_ba();
  ba->ctx = ctx;
  SVS_DEVICE(ctx);
  {
    auto flag = [](const char *n) { return getenv(n) ? 1 : 0; };
    auto num = [](const char *n, int lo, int hi, int dflt) { const char *e = getenv(n); if (!e) return dflt; const int v = atoi(e); return v < lo ? lo : (v > hi ? hi : v); };
    BaOptions &o = ba->opt;
    o.no_speculation = flag("SVS_BA_NO_SPECULATION"); o.one_front = flag("SVS_BA_ONE_FRONT"); o.no_fused_solve = flag("SVS_BA_NO_FUSED_SOLVE");
    o.no_lds_solve = flag("SVS_BA_NO_LDS_SOLVE"); o.no_fused_cons = flag("SVS_BA_NO_FUSED_CONS"); o.nw4 = flag("SVS_BA_NW4");
    o.debug = num("SVS_BA_DEBUG", 0, 2, 0); o.nw = num("SVS_BA_NW", 4, 8, 0); o.p1 = num("SVS_BA_P1", 0, SOLVE_MAX_P, -1);
    o.group = num("SVS_BA_GROUP", 1, WIN, 0); o.host_threads = num("SVS_HOST_THREADS", 1, 64, 0);
  }
  for (auto &e : ba->ev) SVS_HIP(ctx, hipEventCreate(&e));
  *out = ba;
  return SVS_OK;
}
extern "C" int svs_ba_destroy(svs_ba *ba) {
  if (!ba) return SVS_OK;
  (void)hipStreamSynchronize(ba->ctx->stream);
  if (ba->h_scal) { (void)hipHostFree(ba->h_scal); ba->h_scal = nullptr; }
  if (ba->h_ctl) { (void)hipHostFree(ba->h_ctl); ba->h_ctl = nullptr; }
  if (ba->h_stage) { (void)hipHostFree(ba->h_stage); ba->h_stage = nullptr; ba->h_stage_cap = 0; }
  if (ba->h_state) { (void)hipHostFree(ba->h_state); ba->h_state = nullptr; ba->h_state_cap = 0; }
  if (ba->d_ctl) { (void)hipFree(ba->d_ctl); ba->d_ctl = nullptr; }
  if (ba->d_rowmax2) (void)hipFree(ba->d_rowmax2);
  if (ba->d_perm) (void)hipFree(ba->d_perm);
  if (ba->d_perm_sys) (void)hipFree(ba->d_perm_sys);
  if (ba->d_xfer) (void)hipFree(ba->d_xfer);
  if (ba->d_flags) (void)hipFree(ba->d_flags);
  if (ba->d_gridbar) (void)hipFree(ba->d_gridbar);
  if (ba->d_tilews) (void)hipFree(ba->d_tilews);
  if (ba->w_store) (void)hipFree(ba->w_store);
  if (ba->w_pose_tab) (void)hipFree(ba->w_pose_tab);
  if (ba->w_point_tab) (void)hipFree(ba->w_point_tab);
  if (ba->w_work) (void)hipFree(ba->w_work);
  if (ba->w_in) (void)hipFree(ba->w_in);
  if (ba->w_hback) (void)hipHostFree(ba->w_hback);
  for (auto &e : ba->spec_ev) if (e) (void)hipEventDestroy(e);
  ba->spec_ev.clear();
  for (auto &g : ba->graph) if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
  ba->free_all();
  ba->pool = nullptr;                   // shared, not owned
  for (auto &e : ba->ev) if (e) (void)hipEventDestroy(e);
  delete ba;
  return SVS_OK;
}

// the marshalling workers: ONE pool per process, shared by all optimizers, created on first use
static HostPool *host_pool(svs_ba *ba) {
  if (!ba->pool) {
    static std::mutex pool_mutex;
    static HostPool *shared_pool = nullptr;
    std::lock_guard<std::mutex> lk(pool_mutex);
    if (!shared_pool) {
      const unsigned hc = std::max(1u, std::thread::hardware_concurrency());
      int nt = (int)std::min(8u, std::max(1u, hc / 2));
      if (ba->opt.host_threads > 0) nt = std::min(ba->opt.host_threads, (int)hc);      // never more workers than cores
      shared_pool = new HostPool(nt);                                                  // lives for the process
    }
    ba->pool = shared_pool;
  }
  return ba->pool;
}

static int window_update_impl(svs_ba *ba, int P, const int32_t *h_pose_ids, const double *h_poses, int L, const int32_t *h_point_ids,
                              const double *h_psi, const int32_t *h_anchor_pose_ids, int n_new, const svs_ba_edge *h_new_obs, int C,
                              const svs_ba_constraint *h_cons, const svs_cam *cam, const svs_ba_params *prm, bool obs_checked = false);
extern "C" int svs_ba_set_problem(svs_ba *ba, int P, const double *h_poses, int L, const double *h_psi, int E,
                                  const svs_ba_edge *h_edges, int C, const svs_ba_constraint *h_cons, const svs_cam *cam,
                                  const svs_ba_params *prm, int add_pose_terms) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && h_poses && (L == 0 || h_psi) && (E == 0 || h_edges) && (C == 0 || h_cons) && cam && prm);
  SVS_REQUIRE(ctx, P >= 1 && L >= 0 && E >= 0 && C >= 0);
  SVS_DEVICE(ctx);
  // Round 3: a complete window handed over by index takes the DEVICE route of the persistent window -- the edges are uploaded as they come and the
  // sort by (anchor, landmark, observer), the wave chunks, the slot order and the co-visibility pattern are built by kernels (ba_window.inc) -- unless
  // this rank holds a landmark shard (comm / add_pose_terms == 0), a persistent window is in use on this handle, or a point has no observation.
  // (Round 4, set + optimize + get per call, device vs host route: 15 KF / 3k (15 k edges) 0.39 vs 0.43 ms; 30 KF / 8k 0.50 vs 0.74; 50 KF / 20k 0.73 vs 1.39.
  // Below ~8k edges the dozen stream operations of the device route are the larger part.  "host_marshal" = 2 forces the device route for any size: tests.)
  if (ba->opt.host_marshal != 1 && (E >= 8192 || ba->opt.host_marshal == 2) && !ba->comm && add_pose_terms && ba->w_n == 0 && E > 0 && L > 0 && P <= SOLVE_MAX_P) {
    ba->problem_valid = false;                   // a call that fails must leave the handle unusable, not half old / half new
    std::vector<int32_t> &pid = ba->w_ids_p, &lid = ba->w_ids_l, &aid = ba->w_ids_a;
    pid.resize((size_t)P); lid.resize((size_t)L); aid.assign((size_t)L, -1);
    for (int i = 0; i < P; ++i) pid[i] = i;
    for (int i = 0; i < L; ++i) lid[i] = i;
    // two passes over the 12 index bytes of every record, on the worker pool: (1) range check, anchor of the point noted (plain racing stores: every
    // writer of a consistent input stores the same value); (2) every record's anchor equals the noted one -- an inconsistent input fails pass 2 whatever
    // the interleaving of pass 1 was
    std::atomic<int> bad{0};
    {
      HostPool &pool = *host_pool(ba);
      std::lock_guard<std::mutex> pool_lock(pool.use_);
      int32_t *a = aid.data();
      pool.run([&](int t, int n) {
        const int e0 = (int)((long long)E * t / n), e1 = (int)((long long)E * (t + 1) / n);
        for (int e = e0; e < e1; ++e) {
          const svs_ba_edge &ed = h_edges[e];
          if (!(ed.point >= 0 && ed.point < L && ed.pose >= 0 && ed.pose < P && ed.anchor >= 0 && ed.anchor < P)) { bad.store(1); return; }
          __atomic_store_n(a + ed.point, ed.anchor, __ATOMIC_RELAXED);
        }
      });
      if (!bad.load())
        pool.run([&](int t, int n) {
          const int e0 = (int)((long long)E * t / n), e1 = (int)((long long)E * (t + 1) / n);
          for (int e = e0; e < e1; ++e)
            if (__atomic_load_n(a + h_edges[e].point, __ATOMIC_RELAXED) != h_edges[e].anchor) { bad.store(2); return; }
        });
    }
    SVS_REQUIRE(ctx, bad.load() == 0);           // an index out of range, or two anchors for one point
    bool all_seen = true;
    for (int i = 0; i < L && all_seen; ++i) all_seen = aid[i] >= 0;
    if (all_seen) {
      const int rc = window_update_impl(ba, P, pid.data(), h_poses, L, lid.data(), h_psi, aid.data(), E, h_edges, C, h_cons, cam, prm, true);
      ba->w_n = 0;                               // the observation store served as scratch: this handle has no persistent window
      return rc;
    }
  }
  // a call that fails half-way must not leave new sizes next to old buffers behind: the handle is unusable until a call completes
  ba->problem_valid = false;
  if (P > SOLVE_MAX_P) { ctx->err = "svs_ba: P > 256 poses not supported by the single-workgroup solve yet"; return SVS_ERR_UNSUPPORTED; }
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ba->h_stage_used = 0;
  { int rc = stage_reserve(ba, sizeof(double) * (12 * (size_t)P + 3 * (size_t)L) + sizeof(svs_ba_constraint) * (size_t)C + sizeof(int) * ((size_t)E / 8 + 6 * (size_t)P) + 8192);
    if (rc) return rc; }
  const bool dbg_t = ba->opt.debug != 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t_0 = now(), t_1 = t_0, t_2 = t_0, t_3 = t_0, t_4 = t_0, t_5 = t_0;
  ba->P = P; ba->L = L; ba->E = E; ba->C = C; ba->cam = *cam; ba->prm = *prm; ba->add_pose_terms = add_pose_terms; ba->cur = 0;
  ba->profile_ready = false; ba->env_R = 0; ba->use_lds_solve = ba->use_fused_solve = false;
  // Edge order (copyDataToG2o iterates hash sets, so the reference has no meaningful edge order to
  // preserve): landmarks are grouped by anchor so that the poses a workgroup touches stay inside its
  // LDS accumulation window, but inside a group of G consecutive anchors the landmarks are dealt
  // round-robin over the anchors -- neighbouring lanes then hit different pose blocks and the LDS
  // atomics of one instruction rarely collide on an address.  Whole landmarks are packed into
  // <=64-edge wave chunks, observers ascending inside a landmark.
  // All of it is linear time (bucket passes, no comparison sort of the edge list): this marshalling is what a caller
  // pays per optimize() once the kernels take a fraction of a millisecond.
  std::vector<int> &anchor_of = ba->w_anchor, &n_obs = ba->w_nobs, &lm_pos = ba->w_pos, &lm_off = ba->w_off;
  anchor_of.assign(L, -1); n_obs.assign(L, 0);
  int span = 1;
  // worker pool: one pass over the edge records is memory-bound on one core
  HostPool &pool = *host_pool(ba);
  std::lock_guard<std::mutex> pool_lock(pool.use_);
  const bool par = pool.size() > 1 && E >= 16384 && (size_t)L * pool.size() <= ((size_t)1 << 24);
  auto for_range = [&](int total, const std::function<void(int, int, int)> &body) {      // body(t, begin, end)
    if (!par) { body(0, 0, total); return; }
    pool.run([&](int t, int n) { const long long a = (long long)total * t / n, b = (long long)total * (t + 1) / n; body(t, (int)a, (int)b); });
  };
  std::vector<uint64_t> &keys = ba->w_keys, &ent = ba->w_ent;
  keys.resize((size_t)std::max(E, 1)); ent.resize((size_t)std::max(E, 1));
  // Pass 1 over the edge records, no shared writes: every worker keeps, per landmark, the count of its own edges and the
  // anchor it saw ((anchor + 1) << 8 | count in one word), and leaves a compact (point, pose) key per edge.  The merge
  // turns the per-worker counts into per-worker start offsets inside the landmark (a counting sort without atomics whose
  // result -- input order inside a landmark -- does not depend on the number of workers).
  const int T = par ? pool.size() : 1;
  std::vector<int> &cnt = ba->w_cnt;
  cnt.assign((size_t)T * std::max(L, 1), 0);
  {
    std::atomic<int> err{0};
    std::vector<int> span_t(T, 1);
    for_range(E, [&](int t, int i0, int i1) {
      int sp = 1;
      int *c = cnt.data() + (size_t)t * L;
      for (int i = i0; i < i1; ++i) {
        const svs_ba_edge &e = h_edges[i];
        if (!(e.point >= 0 && e.point < L && e.pose >= 0 && e.pose < P && e.anchor >= 0 && e.anchor < P)) { err.store(1); return; }
        const int w = c[e.point];
        if (w != 0 && (w >> 12) != e.anchor + 1) { err.store(2); return; }        // one anchor per point (slam_graph.hpp:121-133)
        if ((w & 0xfff) >= WIDE_THREADS) { err.store(3); return; }
        c[e.point] = ((e.anchor + 1) << 12) | ((w & 0xfff) + 1);
        sp = std::max(sp, std::abs(e.pose - e.anchor) + 1);
        keys[i] = ((uint64_t)(uint32_t)e.point << 32) | (uint32_t)e.pose;           // all the later passes need of the record
      }
      span_t[t] = sp;
    });
    if (err.load() == 0)
      for_range(L, [&](int, int l0, int l1) {
        for (int l = l0; l < l1; ++l) {
          int total = 0, anc = -1;
          for (int t = 0; t < T; ++t) {
            int &w = cnt[(size_t)t * L + l];
            if (w != 0) {
              const int a = (w >> 12) - 1;
              if (anc >= 0 && a != anc) err.store(2);
              anc = a;
            }
            const int n = w & 0xfff;
            w = total;                                                              // start offset of worker t inside landmark l
            total += n;
          }
          if (total > WIDE_THREADS) err.store(3);
          anchor_of[l] = anc; n_obs[l] = total;
        }
      });
    if (err.load() == 3) { ctx->err = "svs_ba: a landmark with more than 256 observations (more than a window can hold)"; return SVS_ERR_UNSUPPORTED; }
    if (err.load() == 1) { ctx->err = "svs_ba_set_problem: edge index out of range"; return SVS_ERR_INVALID; }
    if (err.load() == 2) { ctx->err = "svs_ba_set_problem: a point is observed with two different anchors"; return SVS_ERR_INVALID; }
    for (int v : span_t) span = std::max(span, v);
  }
  t_1 = now();
  // anchors interleaved per group.  One: a workgroup's landmarks then share their anchor (or two neighbouring ones where a batch crosses
  // the group boundary), its window is as narrow as the data allows (span .. span + 1 poses) and holds the most copies; the lanes that
  // collide on one address -- the landmarks of a wave seen from the same keyframe -- are dealt to different copies by landmark ordinal.
  // (Round 1 interleaved up to 8 anchors to spread those lanes over blocks instead: that needs a 2 x wider window = a quarter of the copies.)
  int G = 1;
  if (ba->opt.group > 0) G = ba->opt.group;                                 // experiments only (clamped to 1..WIN at svs_ba_create / set_option)
  // landmark order: per anchor the landmarks in index order; per group of G anchors deal them round-robin
  std::vector<int> &by_anchor_off = ba->w_aoff, &by_anchor = ba->w_alist, &lm_order = ba->w_order;
  by_anchor_off.assign(P + 1, 0);
  for (int l = 0; l < L; ++l) if (anchor_of[l] >= 0) ++by_anchor_off[anchor_of[l] + 1];
  for (int a = 0; a < P; ++a) by_anchor_off[a + 1] += by_anchor_off[a];
  by_anchor.resize(by_anchor_off[P]);
  {
    std::vector<int> fill(by_anchor_off.begin(), by_anchor_off.end() - 1);
    for (int l = 0; l < L; ++l) if (anchor_of[l] >= 0) by_anchor[fill[anchor_of[l]]++] = l;
  }
  lm_order.clear(); lm_order.reserve(by_anchor.size());
  for (int g0 = 0; g0 < P; g0 += G) {
    const int g1 = std::min(P, g0 + G);
    int longest = 0;
    for (int a = g0; a < g1; ++a) longest = std::max(longest, by_anchor_off[a + 1] - by_anchor_off[a]);
    for (int r = 0; r < longest; ++r)
      for (int a = g0; a < g1; ++a)
        if (r < by_anchor_off[a + 1] - by_anchor_off[a]) lm_order.push_back(by_anchor[by_anchor_off[a] + r]);
  }
  // landmarks with more than 64 observations go to the end of the order: they do not fit a wave chunk and get a workgroup each
  const size_t n_lm_all = lm_order.size();
  std::stable_partition(lm_order.begin(), lm_order.end(), [&](int l) { return n_obs[l] <= 64; });
  size_t n_lm_reg = n_lm_all;
  while (n_lm_reg > 0 && n_obs[lm_order[n_lm_reg - 1]] > 64) --n_lm_reg;
  // Workgroup composition.  A workgroup of the Schur kernel = nw consecutive wave chunks (whole landmarks, <= 64 edges each).
  //  * The pair phase of a wave runs as many rounds as its LARGEST landmark needs, and the waves of a workgroup share one LDS
  //    pipeline: inside each workgroup's batch the landmarks are sorted by observation count, so most waves hold landmarks of
  //    one size (1..2 rounds) and only the wave(s) with the large ones run 3 (every wave ran 3..4 before: max of a random mix).
  //    Every workgroup keeps the same mix, so the CUs stay evenly loaded.
  //  * A workgroup accumulates into an LDS window of WIN poses from the smallest pose it touches; one that held landmarks of TWO
  //    anchor groups would reach past it (2G + span - 1 poses) and send those blocks to global atomics -- 6 such workgroups of 231
  //    set the kernel's span at 50 KF (39 vs 32 us).  So a group of anchors starts a new workgroup (empty chunks pad the last one).
  std::vector<int> &chunk_nlm = ba->w_chunk_nlm;
  chunk_nlm.clear();
  {
    long e_reg = 0;
    for (size_t k = 0; k < n_lm_reg; ++k) e_reg += n_obs[lm_order[k]];
    const int nw = pick_nw((int)((e_reg + 59) / 60), ctx->n_cu, ba->opt);
    ba->nw_sched = nw;
    const bool align_groups = 2 * G + span - 1 > WIN;
    std::vector<int> out, cand, carry;
    out.reserve(n_lm_reg);
    size_t k = 0;
    while (k < n_lm_reg || !carry.empty()) {
      const int grp = !carry.empty() ? anchor_of[carry[0]] / G : anchor_of[lm_order[k]] / G;
      cand.clear();
      int sum = 0;
      for (int l : carry) { cand.push_back(l); sum += n_obs[l]; }
      carry.clear();
      while (k < n_lm_reg && sum + n_obs[lm_order[k]] <= nw * 64 && (!align_groups || anchor_of[lm_order[k]] / G == grp)) { sum += n_obs[lm_order[k]]; cand.push_back(lm_order[k]); ++k; }
      std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return n_obs[a] > n_obs[b]; });
      int n_ch = 0, len = 0, cnt = 0;
      size_t i = 0;
      for (; i < cand.size(); ++i) {
        const int n = n_obs[cand[i]];
        if (cnt > 0 && len + n > 64) { chunk_nlm.push_back(cnt); ++n_ch; len = 0; cnt = 0; if (n_ch == nw) break; }
        len += n; ++cnt; out.push_back(cand[i]);
      }
      if (cnt > 0) { chunk_nlm.push_back(cnt); ++n_ch; }
      for (; i < cand.size(); ++i) carry.push_back(cand[i]);          // did not fit the nw chunks after sorting: first in line for the next workgroup
      const bool group_ends = carry.empty() && (k >= n_lm_reg || anchor_of[lm_order[k]] / G != grp);
      if (align_groups && group_ends && k < n_lm_reg) while (n_ch < nw) { chunk_nlm.push_back(0); ++n_ch; }
    }
    std::copy(out.begin(), out.end(), lm_order.begin());
  }
  // edge slots: landmarks in that order, each with its observers ascending (insertion sort inside the landmark's slots)
  lm_pos.assign(L, -1); lm_off.assign(lm_order.size() + 1, 0);
  for (size_t k = 0; k < lm_order.size(); ++k) { lm_pos[lm_order[k]] = (int)k; lm_off[k + 1] = lm_off[k] + n_obs[lm_order[k]]; }
  if ((size_t)E > ba->h_edges_cap) {
    if (ba->h_edges) (void)hipHostFree(ba->h_edges);
    ba->h_edges_cap = (size_t)E + (size_t)E / 4 + 64;
    SVS_HIP(ctx, hipHostMalloc((void **)&ba->h_edges, sizeof(svs_ba_edge) * ba->h_edges_cap, hipHostMallocDefault));
  }
  // device buffers persist across calls and only grow (the window changes by about one keyframe per call)
  auto ensure = [&](void **ptr, size_t *cap, size_t bytes) -> hipError_t {
    if (bytes <= *cap && *ptr) return hipSuccess;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr; *cap = 0;
    const size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(ptr, want);
    if (e == hipSuccess) *cap = want;
    return e;
  };
  t_2 = now();
  svs_ba_edge *sorted = ba->h_edges;      // pinned: the upload is a plain DMA, no pageable staging copy
  for_range(E, [&](int t, int i0, int i1) {            // pass 2, compact arrays only: every edge takes its slot (same worker ranges as pass 1)
    int *c = cnt.data() + (size_t)t * L;
    for (int i = i0; i < i1; ++i) {
      const int pt = (int)(keys[i] >> 32);
      ent[lm_off[lm_pos[pt]] + c[pt]++] = ((keys[i] & 0xffffffffull) << 32) | (uint32_t)i;      // (pose, source index)
    }
  });
  // ... then each landmark is put in observer order (<= 64 entries, insertion sort on the (pose, index) words; the result
  // does not depend on the slot order above because (point, keyframe) pairs are unique), and the structural pattern of
  // the reduced camera system is marked: poses sharing a landmark, and constraints
  const size_t n_lm = lm_order.size();
  ba->h_pattern.assign((size_t)P * P, 0.0);
  std::vector<double> &pat_local = ba->w_pat_local;
  pat_local.assign((size_t)P * P * (par ? pool.size() : 0), 0.0);
  {
    std::atomic<int> dup{0};
    for_range((int)n_lm, [&](int t, int k0, int k1) {
      double *pat = par ? pat_local.data() + (size_t)t * P * P : ba->h_pattern.data();
      for (int k = k0; k < k1; ++k) {
        const int a = lm_off[k], b = lm_off[k + 1];
        for (int i = a + 1; i < b; ++i) {
          const uint64_t e = ent[i];
          int j = i;
          while (j > a && ent[j - 1] > e) { ent[j] = ent[j - 1]; --j; }
          ent[j] = e;
        }
        for (int i = a + 1; i < b; ++i) if ((ent[i] >> 32) == (ent[i - 1] >> 32)) dup.store(1);      // one observation per (point, keyframe)
        const int anc = anchor_of[lm_order[k]], p_first = (int)(ent[a] >> 32), p_last = (int)(ent[b - 1] >> 32);
        const int lo = std::min(p_first, anc), hi = std::max(p_last, anc);
        // envelope only needs, per pose, the farthest co-visible pose: mark (p, hi) for every pose of the landmark
        for (int e = a; e < b; ++e) pat[(size_t)(ent[e] >> 32) * P + hi] = 1.0;
        pat[(size_t)anc * P + hi] = 1.0;
        pat[(size_t)lo * P + hi] = 1.0;
        // ... and, for the reversed front of the two-front solve, per pose the FIRST co-visible pose: mark (lo, p)
        for (int e = a; e < b; ++e) pat[(size_t)lo * P + (ent[e] >> 32)] = 1.0;
        pat[(size_t)lo * P + anc] = 1.0;
      }
    });
    if (dup.load()) { ctx->err = "svs_ba_set_problem: two observations of one point in one keyframe"; return SVS_ERR_INVALID; }
    if (par)
      for (int t = 0; t < pool.size(); ++t)
        for (size_t i = 0; i < (size_t)P * P; ++i) if (pat_local[(size_t)t * P * P + i] != 0.0) ba->h_pattern[i] = 1.0;
  }
  t_3 = now();
  // gather the records into slot order (sequential writes into the pinned buffer, each record read once) in a few
  // sub-ranges, each uploaded as soon as it is complete: the DMA of one overlaps the gather of the next
  SVS_HIP(ctx, ensure((void **)&ba->d_edges, &ba->cap_edges, sizeof(svs_ba_edge) * (size_t)std::max(E, 1)));
  {
    const int n_sub = par ? 4 : 1;
    for (int sb = 0; sb < n_sub; ++sb) {
      const int s0 = (int)((long long)E * sb / n_sub), s1 = (int)((long long)E * (sb + 1) / n_sub);
      for_range(s1 - s0, [&](int, int i0, int i1) {
        for (int i = s0 + i0; i < s0 + i1; ++i) sorted[i] = h_edges[(uint32_t)ent[i]];
      });
      if (s1 > s0) SVS_HIP(ctx, hipMemcpyAsync(ba->d_edges + s0, sorted + s0, sizeof(svs_ba_edge) * (size_t)(s1 - s0), hipMemcpyHostToDevice, ctx->stream));
    }
  }
  std::vector<int> &cs = ba->w_cs, &cl = ba->w_cl;
  cs.clear(); cl.clear();
  {
    size_t k = 0;
    for (int cnt : chunk_nlm) { cs.push_back(lm_off[k]); cl.push_back(lm_off[k + cnt] - lm_off[k]); k += cnt; }
  }
  ba->n_chunks = (int)cs.size();
  ba->n_wide = (int)(n_lm - n_lm_reg);
  for (size_t k = n_lm_reg; k < n_lm; ++k) { cs.push_back(lm_off[k]); cl.push_back(lm_off[k + 1] - lm_off[k]); }      // wide landmarks: entries behind the chunks
  if (add_pose_terms)
    for (int c = 0; c < C; ++c) {
      SVS_REQUIRE(ctx, h_cons[c].pose1 >= 0 && h_cons[c].pose1 < P && h_cons[c].pose2 >= 0 && h_cons[c].pose2 < P);
      ba->h_pattern[(size_t)std::min(h_cons[c].pose1, h_cons[c].pose2) * P + std::max(h_cons[c].pose1, h_cons[c].pose2)] = 1.0;
    }
  t_4 = now();
  const size_t nblk = (size_t)P * (P + 1) / 2;
  ba->red_count = nblk * 36 + 12 * (size_t)P + SC_SLOTS;
  for (int k = 0; k < 2; ++k) {
    SVS_HIP(ctx, ensure((void **)&ba->d_poses[k], &ba->cap_poses[k], sizeof(double) * 12 * (size_t)P));
    SVS_HIP(ctx, ensure((void **)&ba->d_psi[k], &ba->cap_psi[k], sizeof(double) * 3 * (size_t)std::max(L, 1)));
  }
  { int rc = stage_upload(ba, ba->d_poses[0], h_poses, sizeof(double) * 12 * (size_t)P); if (rc) return rc; }
  SVS_HIP(ctx, hipMemcpyAsync(ba->d_poses[1], ba->d_poses[0], sizeof(double) * 12 * (size_t)P, hipMemcpyDeviceToDevice, ctx->stream));
  if (L) {
    int rc = stage_upload(ba, ba->d_psi[0], h_psi, sizeof(double) * 3 * (size_t)L); if (rc) return rc;
    SVS_HIP(ctx, hipMemcpyAsync(ba->d_psi[1], ba->d_psi[0], sizeof(double) * 3 * (size_t)L, hipMemcpyDeviceToDevice, ctx->stream));
  }
  SVS_HIP(ctx, ensure((void **)&ba->d_chunk_start, &ba->cap_cs, sizeof(int) * (size_t)std::max(ba->n_chunks + ba->n_wide, 1)));
  SVS_HIP(ctx, ensure((void **)&ba->d_chunk_len, &ba->cap_cl, sizeof(int) * (size_t)std::max(ba->n_chunks + ba->n_wide, 1)));
  SVS_HIP(ctx, ensure((void **)&ba->d_cons, &ba->cap_cons, sizeof(svs_ba_constraint) * (size_t)std::max(C, 1)));
  SVS_HIP(ctx, ensure((void **)&ba->d_red, &ba->cap_red, sizeof(double) * ba->red_count));
  SVS_HIP(ctx, ensure((void **)&ba->d_x, &ba->cap_x, sizeof(double) * 6 * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_scal, &ba->cap_scal, sizeof(double) * SC_N));
  SVS_HIP(ctx, ensure((void **)&ba->d_linv, &ba->cap_linv, sizeof(double) * 36 * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_rowmax, &ba->cap_rowmax, sizeof(int) * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_colmin, &ba->cap_colmin, sizeof(int) * (size_t)P));
  SVS_HIP(ctx, ensure((void **)&ba->d_pattern, &ba->cap_pattern, sizeof(double) * (size_t)P * P));
  if (!cs.empty()) {
    int rc = stage_upload(ba, ba->d_chunk_start, cs.data(), sizeof(int) * cs.size()); if (rc) return rc;
    rc = stage_upload(ba, ba->d_chunk_len, cl.data(), sizeof(int) * cl.size()); if (rc) return rc;
  }
  if (C) { int rc = stage_upload(ba, ba->d_cons, h_cons, sizeof(svs_ba_constraint) * (size_t)C); if (rc) return rc; }
  SVS_HIP(ctx, hipMemsetAsync(ba->d_x, 0, sizeof(double) * 6 * (size_t)P, ctx->stream));
  t_5 = now();
  // No synchronisation here: everything above is ordered on the ctx stream in front of whatever the caller enqueues next;
  // the caller's (pageable) arrays have been staged by the runtime when hipMemcpyAsync returns, the pinned edge buffer and
  // the work vectors are only touched again by the next set_problem, which starts with a stream synchronisation.
  if (dbg_t) SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (dbg_t) {
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "[svs_ba] set_problem: validate+count %.0f us, landmark order %.0f us, slots+sort+pattern %.0f us, gather+upload+chunks %.0f us, enqueue copies %.0f us, wait %.0f us\n",
            us(t_0, t_1), us(t_1, t_2), us(t_2, t_3), us(t_3, t_4), us(t_4, t_5), us(t_5, now()));
  }
  ba->problem_valid = true;
  return SVS_OK;
}

#include "ba_window.inc"
extern "C" int svs_ba_reset_state(svs_ba *ba, const double *h_poses, const double *h_psi) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && h_poses && ba->d_poses[0] && ba->problem_valid);
  SVS_DEVICE(ctx);
  ba->cur = 0;
  for (int k = 0; k < 2; ++k) {
    SVS_HIP(ctx, hipMemcpyAsync(ba->d_poses[k], h_poses, sizeof(double) * 12 * (size_t)ba->P, hipMemcpyHostToDevice, ctx->stream));
    if (ba->L && h_psi) SVS_HIP(ctx, hipMemcpyAsync(ba->d_psi[k], h_psi, sizeof(double) * 3 * (size_t)ba->L, hipMemcpyHostToDevice, ctx->stream));
  }
  SVS_HIP(ctx, hipMemsetAsync(ba->d_x, 0, sizeof(double) * 6 * (size_t)ba->P, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVS_OK;
}

// Block envelope of the reduced system (with Cholesky fill).  In sharded runs the structural
// pattern is summed over ranks first (each rank only knows its own landmarks' co-visibility).
static int ensure_profile(svs_ba *ba, svs_allreduce_fn allreduce, void *user) {
  if (ba->profile_ready) return SVS_OK;
  svs_ctx *ctx = ba->ctx;
  const int P = ba->P;
  std::vector<double> pat = ba->h_pattern;
  if (allreduce) {
    SVS_HIP(ctx, hipMemcpyAsync(ba->d_pattern, pat.data(), sizeof(double) * pat.size(), hipMemcpyHostToDevice, ctx->stream));
    if (allreduce(ba->d_pattern, pat.size(), user)) { ctx->err = "allreduce callback failed"; return SVS_ERR_INVALID; }
    SVS_HIP(ctx, hipMemcpyAsync(pat.data(), ba->d_pattern, sizeof(double) * pat.size(), hipMemcpyDeviceToHost, ctx->stream));
    SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  std::vector<int> rowmax(P), colmin(P);
  auto filled_envelope = [&](const std::vector<double> &pt, std::vector<int> &rm) {      // returns max row length (block rows) and fills rm
    for (int i = 0; i < P; ++i) {
      rm[i] = i;
      for (int j = P - 1; j > i; --j) if (pt[(size_t)i * P + j] != 0.0 || pt[(size_t)j * P + i] != 0.0) { rm[i] = j; break; }
    }
    for (int k = 0; k < P; ++k)                       // fill: row k spreads its reach to rows k+1..rowmax[k]
      for (int i = k + 1; i <= rm[k]; ++i) rm[i] = std::max(rm[i], rm[k]);
    int Rm = 1;
    for (int k = 0; k < P; ++k) Rm = std::max(Rm, rm[k] - k + 1);
    return Rm;
  };
  const int R_nat = filled_envelope(pat, rowmax);
  ba->env_R_natural = R_nat;
  ba->perm_active = false;
  // Reverse Cuthill-McKee on the pose block graph (P <= 256 nodes: host, microseconds): a loop closure on a chain of keyframes couples its two ends, which in
  // the caller's order makes the filled envelope span the whole chain; breadth-first levels from a peripheral node interleave the two arms of the loop and
  // bring the band back to about twice the co-visibility reach.  Taken only when it pays (a landmark seen from most of the window is a clique no order removes).
  if (!ba->opt.no_order && P >= 16 && R_nat >= 24) {
    // the pattern marshalling left (pat) only marks every pose's farthest partners: the full co-visibility pattern comes from the edges on the device
    {
      BaDev Bd = make_dev(ba, 0.0);
      SVS_HIP(ctx, hipMemsetAsync(ba->d_pattern, 0, sizeof(double) * (size_t)P * P, ctx->stream));
      if (Bd.n_chunks > 0) { hipLaunchKernelGGL(ba_full_pattern_kernel, dim3(Bd.n_chunks), dim3(64), 0, ctx->stream, Bd, ba->d_pattern); SVS_LAUNCH_CHECK(ctx); }
      if (Bd.n_wide > 0) { hipLaunchKernelGGL(ba_full_pattern_wide_kernel, dim3(Bd.n_wide), dim3(256), 0, ctx->stream, Bd, ba->d_pattern); SVS_LAUNCH_CHECK(ctx); }
      if (allreduce && allreduce(ba->d_pattern, (size_t)P * P, user)) { ctx->err = "allreduce callback failed"; return SVS_ERR_INVALID; }
      std::vector<double> full((size_t)P * P);
      SVS_HIP(ctx, hipMemcpyAsync(full.data(), ba->d_pattern, sizeof(double) * full.size(), hipMemcpyDeviceToHost, ctx->stream));
      SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      for (size_t i = 0; i < full.size(); ++i) if (full[i] != 0.0) pat[i] = 1.0;      // + the pose-pose constraints pat already holds
    }
    std::vector<std::vector<int>> adj(P);
    for (int i = 0; i < P; ++i) for (int j = 0; j < P; ++j) if (i != j && (pat[(size_t)i * P + j] != 0.0 || pat[(size_t)j * P + i] != 0.0)) adj[i].push_back(j);
    auto bfs_levels = [&](int root, std::vector<int> &order, std::vector<char> &seen) {      // appends the component of root in CM order; returns the last node visited
      size_t head = order.size();
      order.push_back(root); seen[root] = 1;
      while (head < order.size()) {
        const int u = order[head++];
        std::vector<int> nb;
        for (int v : adj[u]) if (!seen[v]) { seen[v] = 1; nb.push_back(v); }
        std::sort(nb.begin(), nb.end(), [&](int a, int b) { return adj[a].size() != adj[b].size() ? adj[a].size() < adj[b].size() : a < b; });
        for (int v : nb) order.push_back(v);
      }
      return order.back();
    };
    std::vector<int> order;
    std::vector<char> seen(P, 0);
    for (int start = 0; start < P; ++start) {
      if (seen[start]) continue;
      // pseudo-peripheral root of this component: run the search from the far end of a search from the node of least degree, twice
      std::vector<char> tmp(seen);
      std::vector<int> comp;
      bfs_levels(start, comp, tmp);
      int root = comp[0];
      for (int v : comp) if (adj[v].size() < adj[root].size()) root = v;
      for (int rep = 0; rep < 2; ++rep) { std::vector<char> t2(seen); std::vector<int> o2; root = bfs_levels(root, o2, t2); }
      bfs_levels(root, order, seen);
    }
    std::reverse(order.begin(), order.end());
    std::vector<double> pp((size_t)P * P, 0.0);
    for (int i = 0; i < P; ++i) for (int j = 0; j < P; ++j) pp[(size_t)i * P + j] = (pat[(size_t)order[i] * P + order[j]] != 0.0 || pat[(size_t)order[j] * P + order[i]] != 0.0) ? 1.0 : 0.0;
    std::vector<int> rm_p(P);
    const int R_perm = filled_envelope(pp, rm_p);
    if (R_perm * 4 <= R_nat * 3) {      // at least a quarter narrower
      ba->perm_active = true;
      ba->h_perm = order;
      rowmax = rm_p;
      pat = pp;      // (the two-front profile below reads the pattern of the order the solve runs in)
      const size_t nblk_p = (size_t)P * (P + 1) / 2, n_sys = nblk_p * 36 + 12 * (size_t)P + 24 * (size_t)P + 6 * (size_t)P;
      if (!ba->d_perm || ba->cap_perm_sys < sizeof(double) * n_sys) {
        if (ba->d_perm) (void)hipFree(ba->d_perm);
        if (ba->d_perm_sys) (void)hipFree(ba->d_perm_sys);
        ba->d_perm = nullptr; ba->d_perm_sys = nullptr;
        SVS_HIP(ctx, hipMalloc(&ba->d_perm, sizeof(int) * SOLVE_MAX_P));
        SVS_HIP(ctx, hipMalloc(&ba->d_perm_sys, sizeof(double) * n_sys));
        ba->cap_perm_sys = sizeof(double) * n_sys;
      }
      { int rc = stage_upload(ba, ba->d_perm, order.data(), sizeof(int) * P); if (rc) return rc; }
    }
  }
  for (int k = 0; k < P; ++k) { colmin[k] = k; for (int i = 0; i < k; ++i) if (rowmax[i] >= k) { colmin[k] = i; break; } }
  { int rc = stage_upload(ba, ba->d_rowmax, rowmax.data(), sizeof(int) * P); if (rc) return rc; }
  { int rc = stage_upload(ba, ba->d_colmin, colmin.data(), sizeof(int) * P); if (rc) return rc; }
  int R = 1;
  for (int k = 0; k < P; ++k) R = std::max(R, rowmax[k] - k + 1);
  ba->env_R = R;                                    // (of the order the solve runs in)
  // LDS budget of the window solve: rhs + R*R window + 2 x (Z, Y) panels + U_kk + 1/diag + z + scratch + envelope + block LUT
  // (the fused kernel keeps U_kk of all rows in LDS: P * 42 doubles; the window kernel only the current row's, the rest in global memory)
  const size_t need_common = sizeof(double) * ((size_t)6 * P + (size_t)R * R * 36 + 4 * (size_t)R * 36 + 16 + 128 + (size_t)R * 6) +
                             sizeof(int) * (size_t)P + (size_t)R * (R - 1) + 16;
  const size_t need_lds = need_common + sizeof(double) * 48, need_fused = need_common + sizeof(double) * (size_t)P * 42;
  ba->use_lds_solve = need_lds <= 150 * 1024 && R * 36 <= 64 * PIPE_LD && !ba->opt.no_lds_solve;
  ba->use_fused_solve = ba->use_lds_solve && need_fused <= 150 * 1024 && R <= FUSE_SLOTS && !ba->opt.no_fused_solve;
  const size_t need = ba->use_fused_solve ? need_fused : need_lds;
  ba->lds_solve_smem = need;
  // two-front elimination (fused kernel only): front 1 takes the last P1 block rows in reversed order.  Balance: front 0
  // needs front 1's deltas when it reaches row P_top-(R-1), i.e. after P - P1 - (R-1) stages; front 1 needs P1 stages + the hand-over.
  ba->fuse_P1 = 0;
  if (ba->use_fused_solve && !ba->opt.one_front) {
    int P1 = (P - (R - 1) - 2) / 2;
    if (ba->opt.p1 >= 0) P1 = ba->opt.p1;                                // experiments only; goes through the same validity checks
    if (P1 >= 4 && P - P1 - (R - 1) >= 2) ba->fuse_P1 = P1;
  }
  std::vector<int> rm2(2 * (size_t)P, 0);
  if (ba->fuse_P1 > 0) {
    const int P1 = ba->fuse_P1, P_top = P - P1, Pv = P1 + R - 1;
    for (int k = 0; k < P_top; ++k) rm2[k] = std::min(rowmax[k], P_top - 1);
    // reversed profile: local row k' <-> natural P-1-k'; its reach = the topmost natural row coupled to it, with the same fill closure
    std::vector<int> rr(P);
    for (int kk = 0; kk < P; ++kk) {
      const int i = P - 1 - kk;
      int top = i;
      for (int j = 0; j < i; ++j) if (pat[(size_t)j * P + i] != 0.0) { top = j; break; }
      rr[kk] = P - 1 - top;
    }
    for (int kk = 0; kk < P; ++kk) for (int i = kk + 1; i <= rr[kk]; ++i) rr[i] = std::max(rr[i], rr[kk]);
    bool ok = true;
    for (int kk = 0; kk < P; ++kk) if (rr[kk] - kk + 1 > R) ok = false;      // (a symmetric band: same width both ways)
    for (int kk = 0; kk < Pv; ++kk) rm2[P + kk] = std::min(rr[kk], Pv - 1);
    if (!ok) ba->fuse_P1 = 0;
  }
  if (ba->fuse_P1 == 0) for (int k = 0; k < P; ++k) rm2[k] = rowmax[k];
  // the fused kernel's back-substitution panel in LDS where the rows of the larger front fit beside the rest (gfx950: 160 KB per workgroup)
  ba->fuse_lds_panel = 0;
  if (ba->use_fused_solve && !ba->opt.no_lds_panel) {
    const int rows = ba->fuse_P1 > 0 ? std::max(P - ba->fuse_P1, ba->fuse_P1 + R - 1) : P;
    const size_t extra = sizeof(double) * ((size_t)rows * FUSE_SLOTS * 36 + 400) + 32;      // + the zero region behind the rows + alignment
    if (ba->lds_solve_smem + extra <= 158 * 1024) { ba->fuse_lds_panel = 1; ba->lds_solve_smem += extra; }
  }
  if (ba->use_fused_solve) {
    if (!ba->d_rowmax2 || ba->cap_rowmax2 < sizeof(int) * rm2.size()) {
      if (ba->d_rowmax2) (void)hipFree(ba->d_rowmax2);
      ba->cap_rowmax2 = sizeof(int) * rm2.size() + 256;
      SVS_HIP(ctx, hipMalloc(&ba->d_rowmax2, ba->cap_rowmax2));
    }
    { int rc = stage_upload(ba, ba->d_rowmax2, rm2.data(), sizeof(int) * rm2.size()); if (rc) return rc; }
    if (!ba->d_xfer) {
      SVS_HIP(ctx, hipMalloc(&ba->d_xfer, sizeof(double) * (FUSE_SLOTS * FUSE_SLOTS * 36 + 2 * FUSE_SLOTS * 6)));
      SVS_HIP(ctx, hipMalloc(&ba->d_flags, sizeof(unsigned) * 4));
      SVS_HIP(ctx, hipMemsetAsync(ba->d_flags, 0, sizeof(unsigned) * 4, ctx->stream));
    }
  }
  if (ba->use_lds_solve) {
    const size_t up_count = 2 * (36 * (size_t)P * std::max(R, FUSE_SLOTS) + 128);      // per front: panel rows + write sink + zero block of the fused kernel
    if (sizeof(double) * up_count > ba->cap_upanel || !ba->d_upanel) {
      if (ba->d_upanel) { (void)hipFree(ba->d_upanel); ba->d_upanel = nullptr; ba->cap_upanel = 0; }
      const size_t want = sizeof(double) * (up_count + up_count / 4);
      SVS_HIP(ctx, hipMalloc(&ba->d_upanel, want));
      ba->cap_upanel = want;
    }
    // the fused kernel's zero block sits right behind the P x 10 panel rows of each front (position depends on P): clear sink + zero block
    SVS_HIP(ctx, hipMemsetAsync(ba->d_upanel + (up_count / 2 - 128), 0, sizeof(double) * 128, ctx->stream));
    SVS_HIP(ctx, hipMemsetAsync(ba->d_upanel + (up_count - 128), 0, sizeof(double) * 128, ctx->stream));
    if (ba->lds_solve_smem > 64 * 1024) {
      SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ba->lds_solve_smem));
      SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_fused_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ba->lds_solve_smem));
      SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_fused_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ba->lds_solve_smem));
      SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_fused_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ba->lds_solve_smem));
      SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_fused_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ba->lds_solve_smem));
    }
  }
  // wide envelopes (neither LDS variant fits): the trailing updates are spread over several workgroups once they are worth it
  ba->grid_G = 0;
  // (measured at P = 230: one workgroup 0.98 ms at R = 19, 29.9 ms at R = 224; several 1.59 / 3.67 ms -- the per-step arrival wait costs ~6 us)
  if (!ba->use_lds_solve && !ba->opt.no_grid_solve && R >= 40) {
    const long tile_rows = (long)R * (R + 1) / 2 * 6;
    ba->grid_G = (int)std::max(8l, std::min((long)ctx->n_cu, (tile_rows + 4 * SOLVE_THREADS - 1) / (4 * SOLVE_THREADS)));
    if (ba->opt.grid_g > 0) ba->grid_G = std::min(ba->opt.grid_g, ctx->n_cu);      // experiments only
    if (!ba->d_gridbar) {
      SVS_HIP(ctx, hipMalloc(&ba->d_gridbar, sizeof(unsigned) * (32 * GRID_NBAR + 4)));
    }
    const size_t smem_g = sizeof(double) * ((size_t)P * 36 + 36 + 6 * (size_t)P);
    if (smem_g > 64 * 1024) SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g));
    // the tile-resident blocked Cholesky where its tiles fit the workgroups' LDS (ba_solve_tiles.inc); the kernel above otherwise (and as the A/B switch "no_tile_solve")
    ba->tiles_G = 0;
    if (!ba->opt.no_tile_solve) {
      const int S = div_up(P, TS_M);
      // square workgroup grid: tile (I, J) belongs to workgroup (I mod gq, J mod gq), one workgroup per CU.  A fifth of the CUs stays free for the front end's
      // latency-mode frame (eight 512-lane workgroups at 180 registers: nothing fits beside them on a CU), which takes the spin gate's priority lane while its
      // workgroups + these fit the device together (image.hip) -- two threads on one GPU: p99 of a frame 0.67 ms with 196 of 256 CUs taken, 1.2-1.5 ms with 225
      int gq = 1;
      while ((gq + 1) * (gq + 1) <= ctx->n_cu - 48) ++gq;
      if (ba->opt.grid_g > 0) { gq = 1; while ((gq + 1) * (gq + 1) <= std::min(ba->opt.grid_g, ctx->n_cu)) ++gq; }      // experiments only
      const int G = gq * gq, per = div_up(S, gq), tpw = per * (per + 1) / 2;      // own tiles at most: the upper triangle of a per x per block
      const int ncm = 2 * per;                            // the rows and the columns of a per x per block of tiles
      // the tiles must fit the LDS a workgroup of THIS device can have (gfx950: 160 KB): the runtime is asked, and a part that refuses (64 KB of LDS) keeps the grid
      // kernel above instead of failing the whole call
      if (tpw <= TS_MAXT && ncm <= 2 * TS_MAXT && ts_lds_bytes(tpw, ncm) <= 158 * 1024 &&
          hipFuncSetAttribute((const void *)ba_solve_tiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ts_lds_bytes(tpw, ncm)) == hipSuccess) {
        const size_t want = sizeof(double) * ts_ws_doubles(S);
        if (!ba->d_tilews || ba->cap_tilews < want) {
          if (ba->d_tilews) (void)hipFree(ba->d_tilews);
          ba->d_tilews = nullptr; ba->cap_tilews = 0;
          SVS_HIP(ctx, hipMalloc(&ba->d_tilews, want + want / 4));
          ba->cap_tilews = want + want / 4;
        }
        ba->tiles_G = G; ba->tiles_S = S; ba->tiles_tpw = tpw; ba->tiles_gq = gq; ba->tiles_ncm = ncm;
      }
    }
  }
  ba->profile_ready = true;
  return SVS_OK;
}

// buildSystem + Schur reduction at the current state (MODE 0 kernels)
static int launch_reduce(svs_ba *ba, double lambda, int cur = -1, double *ctl = nullptr, hipEvent_t *ev = nullptr) {
  svs_ctx *ctx = ba->ctx;
  BaDev B = make_dev(ba, lambda, cur, ctl);
  if (!ev) ev = ba->ev;
  SVS_HIP(ctx, hipMemsetAsync(ba->d_red, 0, sizeof(double) * ba->red_count, ctx->stream));
  if (B.C > 0 && !B.fuse_cons) { hipLaunchKernelGGL(ba_constraint_kernel<0>, dim3(B.C), dim3(64), 0, ctx->stream, B); SVS_LAUNCH_CHECK(ctx); }
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[0], ctx->stream));     // brackets the landmark (Schur) kernel alone
  const bool timeline = ba->opt.debug >= 2 && B.n_chunks > 0;
  if (timeline) SVS_HIP(ctx, hipMalloc(&B.dbg, sizeof(long long) * DBG_W * (size_t)B.n_chunks));
  int dbg_nw = 1;
  if (B.n_chunks > 0) {
    const int nw = (ba->nw_sched > 0 && ba->opt.nw < 4 && !ba->opt.nw4) ? ba->nw_sched : pick_nw(B.n_chunks, ctx->n_cu, ba->opt);
    const int xc = B.fuse_cons ? B.C : 0;      // pose-pose constraints in extra workgroups of the same launch
    const int n_wg = div_up(B.n_chunks, nw);
    dbg_nw = nw;
    switch (nw) {
      case 5: hipLaunchKernelGGL((ba_landmark_kernel<0, 5, 2>), dim3(n_wg + xc), dim3(320), 0, ctx->stream, B); break;
      case 6: hipLaunchKernelGGL((ba_landmark_kernel<0, 6, 2>), dim3(n_wg + xc), dim3(384), 0, ctx->stream, B); break;
      case 7: hipLaunchKernelGGL((ba_landmark_kernel<0, 7, 2>), dim3(n_wg + xc), dim3(448), 0, ctx->stream, B); break;
      case 8: hipLaunchKernelGGL((ba_landmark_kernel<0, 8, 2>), dim3(n_wg + xc), dim3(512), 0, ctx->stream, B); break;
      default:
        if (n_wg <= ctx->n_cu) hipLaunchKernelGGL((ba_landmark_kernel<0, 4, 2>), dim3(n_wg + xc), dim3(256), 0, ctx->stream, B);      // one workgroup per CU: LDS to spare
        else hipLaunchKernelGGL((ba_landmark_kernel<0, 4, 1>), dim3(n_wg + xc), dim3(256), 0, ctx->stream, B);                         // two per CU must fit
        break;
    }
    SVS_LAUNCH_CHECK(ctx);
  }
  if (B.n_wide > 0) { hipLaunchKernelGGL(ba_wide_landmark_kernel<0>, dim3(B.n_wide * B.wide_split), dim3(WIDE_THREADS), 0, ctx->stream, B); SVS_LAUNCH_CHECK(ctx); }
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[1], ctx->stream));
  if (timeline) {   // per-wave timeline of the Schur kernel (debug only; synchronises)
    std::vector<long long> h(DBG_W * (size_t)B.n_chunks);
    SVS_HIP(ctx, hipMemcpyAsync(h.data(), B.dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(B.dbg);
    long long t0 = h[0], t1 = h[DBG_N - 1];
    for (int c = 0; c < B.n_chunks; ++c) { t0 = std::min(t0, h[(size_t)DBG_W * c]); t1 = std::max(t1, h[(size_t)DBG_W * c + DBG_N - 1]); }
    double ph[DBG_N] = {}, phmax[DBG_N] = {}, s_start = 0, mx_start = 0, mx_dur = 0;
    int slow = 0;
    for (int c = 0; c < B.n_chunks; ++c) {
      const long long *d = &h[(size_t)DBG_W * c];
      const double st = (d[0] - t0) * 0.01, dur = (d[DBG_N - 1] - d[0]) * 0.01;
      s_start += st; mx_start = std::max(mx_start, st);
      if (dur > mx_dur) { mx_dur = dur; slow = c; }
      for (int k = 1; k < DBG_N; ++k) { ph[k] += (d[k] - d[k - 1]) * 0.01; phmax[k] = std::max(phmax[k], (d[k] - d[k - 1]) * 0.01); }
    }
    const double n = B.n_chunks;
    static const char *names[DBG_N] = {"", "load", "linearize + landmark sums", "segment reduce", "D^-1, W", "anchor block", "observer blocks", "pairs", "barrier", "flush"};
    fprintf(stderr, "[svs_ba] schur kernel timeline: span %.1f us, %d waves, start avg %.1f max %.1f us, wave duration max %.1f us; phases (us avg / max / slowest wave):",
            (t1 - t0) * 0.01, B.n_chunks, s_start / n, mx_start, mx_dur);
    for (int k = 1; k < DBG_N; ++k) fprintf(stderr, " %s %.2f / %.2f / %.2f |", names[k], ph[k] / n, phmax[k], (h[(size_t)DBG_W * slow + k] - h[(size_t)DBG_W * slow + k - 1]) * 0.01);
    {   // structure of the waves: landmarks per wave, largest same-address multiplicity of one observer-block add / anchor-block add, pair rounds
      static const char *xn[DBG_X] = {"landmarks", "observer multiplicity", "anchor multiplicity", "pair rounds", "", "", "", ""};
      {
        double a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
        for (int c = 0; c < B.n_chunks; ++c) { const long long *d = &h[(size_t)DBG_W * c]; a1 += (d[DBG_N + 4] - d[0]) * 0.01; a2 += (d[DBG_N + 5] - d[DBG_N + 4]) * 0.01; a3 += (d[DBG_N + 6] - d[DBG_N + 5]) * 0.01; a4 += (d[DBG_N + 7] - d[DBG_N + 6]) * 0.01; a5 += (d[1] - d[DBG_N + 7]) * 0.01; }
        fprintf(stderr, "\n[svs_ba]   load phase (avg us, debug waits serialise it): edge record %.2f, state gathers %.2f, window extent + barrier %.2f, zeroing + barrier %.2f, linearize %.2f", a1 / n, a2 / n, a3 / n, a4 / n, a5 / n);
      }
      fprintf(stderr, "\n[svs_ba]   wave structure (avg / max / slowest wave):");
      for (int k = 0; k < 4; ++k) {
        double sum = 0; long long mx = 0;
        for (int c = 0; c < B.n_chunks; ++c) { const long long v = h[(size_t)DBG_W * c + DBG_N + k]; sum += (double)v; mx = std::max(mx, v); }
        fprintf(stderr, " %s %.1f / %lld / %lld |", xn[k], sum / n, mx, h[(size_t)DBG_W * slow + DBG_N + k]);
      }
      // workgroup view: duration of the slowest wave of each workgroup
      const int nwg = (B.n_chunks + dbg_nw - 1) / dbg_nw;
      std::vector<double> wg((size_t)nwg, 0.0);
      for (int c = 0; c < B.n_chunks; ++c) wg[c / dbg_nw] = std::max(wg[c / dbg_nw], (h[(size_t)DBG_W * c + DBG_N - 2] - h[(size_t)DBG_W * c]) * 0.01);
      std::vector<double> srt(wg); std::sort(srt.begin(), srt.end());
      fprintf(stderr, "\n[svs_ba]   workgroups (%d of %d waves): time to the barrier of the slowest wave: min %.1f median %.1f p90 %.1f max %.1f us", nwg, dbg_nw,
              srt.front(), srt[srt.size() / 2], srt[srt.size() * 9 / 10], srt.back());
    }
    fprintf(stderr, "\n");
  }
  return SVS_OK;
}

extern "C" int svs_ba_reduced_system(svs_ba *ba, double lambda, double *h_Hred, double *h_bred, double *h_chi2) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && ba->d_red && ba->problem_valid);
  SVS_DEVICE(ctx);
  int rc = launch_reduce(ba, lambda);
  if (rc) return rc;
  BaDev B = make_dev(ba, lambda);
  const int n = 6 * ba->P;
  double *d_full = nullptr, *d_b = nullptr;
  SVS_HIP(ctx, hipMalloc(&d_full, sizeof(double) * (size_t)n * n));
  SVS_HIP(ctx, hipMalloc(&d_b, sizeof(double) * n));
  hipLaunchKernelGGL(ba_expand_kernel, dim3(256), dim3(256), 0, ctx->stream, B, d_full, d_b);
  SVS_LAUNCH_CHECK(ctx);
  if (h_Hred) SVS_HIP(ctx, hipMemcpyAsync(h_Hred, d_full, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
  if (h_bred) SVS_HIP(ctx, hipMemcpyAsync(h_bred, d_b, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  double chi_slots[SC_SLOTS] = {};
  if (h_chi2) SVS_HIP(ctx, hipMemcpyAsync(chi_slots, B.chi2_cur, sizeof(chi_slots), hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h_chi2) { double t = 0; for (int i = 0; i < SC_SLOTS; ++i) t += chi_slots[i]; *h_chi2 = t; }
  (void)hipFree(d_full); (void)hipFree(d_b);
  return SVS_OK;
}

// enqueue one LM trial on the ctx stream: system at the current state, solve, trial state, trial chi2 + scale
static int enqueue_trial(svs_ba *ba, double lambda, int cur, double *ctl, hipEvent_t *ev, size_t smem_fallback, svs_allreduce_fn allreduce, void *user) {
  svs_ctx *ctx = ba->ctx;
  // (re)build at the current state with this lambda; the state only changes on accept, so a
  // rebuilt system equals g2o's "restore diagonal + add new lambda"
  int rc = launch_reduce(ba, lambda, cur, ctl, ev);
  if (rc) return rc;
  if (allreduce) { rc = allreduce(ba->d_red, ba->red_count, user); if (rc) { ctx->err = "allreduce callback failed"; return SVS_ERR_INVALID; } }
  BaDev B = make_dev(ba, lambda, cur, ctl);
  if (B.n_chunks == 0) SVS_HIP(ctx, hipMemsetAsync(ba->d_scal, 0, sizeof(double) * SC_N, ctx->stream));      // else zeroed by the Schur kernel
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[2], ctx->stream));
  // the order the solve runs in: a permuted copy of the reduced system where the fill-reducing order pays (ensure_profile)
  BaDev Bn = B;                         // the system in the caller's pose order (kept for the scatter back)
  double *x_solve = ba->d_x;
  if (ba->perm_active) {
    const size_t nblk = (size_t)ba->P * (ba->P + 1) / 2;
    double *sys = ba->d_perm_sys;
    B.H = sys; B.bp = sys + nblk * 36; B.bs = B.bp + 6 * (size_t)ba->P;
    double *poses_p = B.bs + 6 * (size_t)ba->P, *trial_p = poses_p + 12 * (size_t)ba->P;
    x_solve = trial_p + 12 * (size_t)ba->P;
    hipLaunchKernelGGL(ba_permute_system_kernel, dim3((unsigned)std::min<size_t>(nblk, 4096)), dim3(256), 0, ctx->stream, Bn, ba->d_perm, B.H, B.bp, B.bs, poses_p);
    SVS_LAUNCH_CHECK(ctx);
    B.poses = poses_p; B.poses_trial = trial_p;
  }
  if (ba->use_fused_solve)
  {
    FuseFronts F{ba->P - ba->fuse_P1, ba->fuse_P1, ba->d_xfer, ba->d_flags, ba->fuse_lds_panel};      // (the launch epoch of the fronts' hand-over flags lives in d_flags[2])
    const dim3 fgrid(ba->fuse_P1 > 0 ? 2 : 1);
    const bool dbg_clocks = ba->opt.debug != 0;      // the per-stage clocks of SVS_BA_DEBUG: their own instantiations (they sit on the pivot wave's critical path)
    if (ba->fuse_lds_panel) {
      if (dbg_clocks) hipLaunchKernelGGL((ba_solve_fused_kernel<true, true>), fgrid, dim3(FUSE_THREADS), ba->lds_solve_smem, ctx->stream, B, x_solve, ba->d_upanel, ba->d_rowmax2, ba->env_R, F);
      else hipLaunchKernelGGL((ba_solve_fused_kernel<true, false>), fgrid, dim3(FUSE_THREADS), ba->lds_solve_smem, ctx->stream, B, x_solve, ba->d_upanel, ba->d_rowmax2, ba->env_R, F);
    } else {
      if (dbg_clocks) hipLaunchKernelGGL((ba_solve_fused_kernel<false, true>), fgrid, dim3(FUSE_THREADS), ba->lds_solve_smem, ctx->stream, B, x_solve, ba->d_upanel, ba->d_rowmax2, ba->env_R, F);
      else hipLaunchKernelGGL((ba_solve_fused_kernel<false, false>), fgrid, dim3(FUSE_THREADS), ba->lds_solve_smem, ctx->stream, B, x_solve, ba->d_upanel, ba->d_rowmax2, ba->env_R, F);
    }
  }
  else if (ba->use_lds_solve)
    hipLaunchKernelGGL(ba_solve_lds_kernel, dim3(1), dim3(PIPE_THREADS), ba->lds_solve_smem, ctx->stream, B, x_solve, ba->d_upanel, ba->d_rowmax, ba->env_R,
                       ba->d_upanel + (36 * (size_t)ba->P * std::max(ba->env_R, FUSE_SLOTS) + 128));      // U_kk of all rows: the second front's half of the panel buffer (unused by this kernel)
  else if (ba->grid_G > 0) {
    const size_t smem_g = sizeof(double) * ((size_t)ba->P * 36 + 36 + 6 * (size_t)ba->P);
    SVS_HIP(ctx, hipMemsetAsync(ba->d_gridbar, 0, sizeof(unsigned) * (32 * GRID_NBAR + 4), ctx->stream));      // arrival counter + failure flag of this launch (a speculative
    SvsSpinScope gate(ctx, ba->tiles_G > 0 ? ba->tiles_G : ba->grid_G);      // grid-wide arrivals; the size of the launch that is really made (grid_G is 8..16 where tiles_G is 196)
    if (gate.rc) return gate.rc;
    if (ba->tiles_G > 0) {
      TilesArgs TA{};
      const int S = ba->tiles_S;
      TA.S = S; TA.tpw = ba->tiles_tpw; TA.gq = ba->tiles_gq; TA.ncm = ba->tiles_ncm;
      TA.rowbuf = ba->d_tilews; TA.bbuf = TA.rowbuf + 2 * (size_t)S * TS_T; TA.ufac = TA.bbuf + 2 * TS_N; TA.ybuf = TA.ufac + (size_t)S * (S + 1) / 2 * TS_T;
      TA.bar = ba->d_gridbar;
      hipLaunchKernelGGL(ba_solve_tiles_kernel, dim3(ba->tiles_G), dim3(TS_THREADS), ts_lds_bytes(ba->tiles_tpw, ba->tiles_ncm), ctx->stream, B, x_solve, TA);
    } else
    hipLaunchKernelGGL(ba_solve_grid_kernel, dim3(ba->grid_G), dim3(SOLVE_THREADS), smem_g, ctx->stream, B, x_solve, ba->d_linv, ba->d_rowmax, ba->d_gridbar, 0u);      // launch may be skipped, so the counter starts at 0 every time)
    SVS_LAUNCH_CHECK(ctx);
    if (int grc = gate.leave()) return grc;
  }
  else
    hipLaunchKernelGGL(ba_solve_kernel, dim3(1), dim3(SOLVE_THREADS), smem_fallback, ctx->stream, B, x_solve, ba->d_linv, ba->d_rowmax, ba->d_colmin);
  if (ba->perm_active) {
    SVS_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(ba_unpermute_kernel, dim3(div_up(ba->P, 64)), dim3(64), 0, ctx->stream, Bn, ba->d_perm, x_solve, B.poses_trial, ba->d_x);
    B = Bn;                             // everything behind the solve (constraints, back-substitution, trial chi2) works in the caller's pose order
  }
  SVS_LAUNCH_CHECK(ctx);
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[3], ctx->stream));
  if (B.C > 0 && !B.fuse_cons) { hipLaunchKernelGGL(ba_constraint_kernel<1>, dim3(B.C), dim3(64), 0, ctx->stream, B); SVS_LAUNCH_CHECK(ctx); }
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[5], ctx->stream));
  if (B.n_chunks > 0) {
    // the same waves per workgroup as the Schur pass: one workgroup per CU where that is possible (403 four-wave workgroups left some CUs
    // with two and others with one: 19.6 us; 231 seven-wave ones: 17.0 us at 50 KF / 20k)
    const int nw = (ba->nw_sched > 0 && ba->opt.nw < 4 && !ba->opt.nw4) ? ba->nw_sched : pick_nw(B.n_chunks, ctx->n_cu, ba->opt);
    const int xc = B.fuse_cons ? B.C : 0;
    switch (nw) {
      case 5: hipLaunchKernelGGL((ba_landmark_kernel<1, 5>), dim3(div_up(B.n_chunks, 5) + xc), dim3(320), 0, ctx->stream, B); break;
      case 6: hipLaunchKernelGGL((ba_landmark_kernel<1, 6>), dim3(div_up(B.n_chunks, 6) + xc), dim3(384), 0, ctx->stream, B); break;
      case 7: hipLaunchKernelGGL((ba_landmark_kernel<1, 7>), dim3(div_up(B.n_chunks, 7) + xc), dim3(448), 0, ctx->stream, B); break;
      case 8: hipLaunchKernelGGL((ba_landmark_kernel<1, 8>), dim3(div_up(B.n_chunks, 8) + xc), dim3(512), 0, ctx->stream, B); break;
      default: hipLaunchKernelGGL((ba_landmark_kernel<1, 4>), dim3(div_up(B.n_chunks, 4) + xc), dim3(256), 0, ctx->stream, B); break;
    }
    SVS_LAUNCH_CHECK(ctx);
  }
  if (B.n_wide > 0) { hipLaunchKernelGGL(ba_wide_landmark_kernel<1>, dim3(B.n_wide), dim3(WIDE_THREADS), 0, ctx->stream, B); SVS_LAUNCH_CHECK(ctx); }
  if (ba->timing) SVS_HIP(ctx, hipEventRecord(ev[4], ctx->stream));
  if (allreduce) { rc = allreduce(ba->d_scal + SC_CHI, 2 * SC_SLOTS, user); if (rc) { ctx->err = "allreduce callback failed"; return SVS_ERR_INVALID; } }
  return SVS_OK;
}
static int add_trial_times(svs_ba *ba, hipEvent_t *ev) {
  svs_ctx *ctx = ba->ctx;
  float ms;
  if (!ba->timing) return SVS_OK;
  SVS_HIP(ctx, hipEventElapsedTime(&ms, ev[0], ev[1])); ba->t_reduce += ms; ba->n_reduce++;
  SVS_HIP(ctx, hipEventElapsedTime(&ms, ev[2], ev[3])); ba->t_solve += ms;
  SVS_HIP(ctx, hipEventElapsedTime(&ms, ev[5], ev[4])); ba->t_backsub += ms;
  return SVS_OK;
}

// OptimizationAlgorithmLevenberg::solve x num_iters, device math.  Control flow: the common path -- the
// first trial of every iteration is accepted -- is enqueued speculatively for all iterations at once with the
// accept/reject decision and the lambda update on the device (ba_lm_kernel), one host synchronisation in total; the first
// rejection (or terminate condition) turns the remaining enqueued kernels into no-ops and the host loop below resumes
// from exactly that point.
// An optimize() in two halves, so that several windows can be in flight at once (svs_ba_optimize_batch): optimize_begin enqueues the
// speculative trials of one window on ITS context's stream and returns without waiting; optimize_finish waits, replays the device's
// decisions and runs the host-driven remainder (rejected trials) if there is one.
struct OptRun {
  svs_allreduce_fn allreduce = nullptr; void *user = nullptr;
  double lambda = 0, ni = 2;
  svs_ba_stats st{};
  size_t smem = 0;
  bool speculate = false;
};
static int optimize_begin(svs_ba *ba, svs_allreduce_fn allreduce, void *user, OptRun &R) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && ba->d_red && ba->problem_valid);
  SVS_DEVICE(ctx);
  if (!allreduce && ba->comm) { allreduce = svs_comm_allreduce_hook; user = ba->comm; }      // library-owned collective (svs_ba_set_comm)
  R.allreduce = allreduce; R.user = user;
  const svs_ba_params &prm = ba->prm;
  double lambda = prm.lambda_init;
  R.lambda = lambda; R.ni = 2; R.st = svs_ba_stats{};
  ba->t_reduce = ba->t_solve = ba->t_backsub = 0; ba->n_reduce = 0;
  { int rc = ensure_profile(ba, allreduce, user); if (rc) return rc; }
  const size_t smem = sizeof(double) * ((size_t)6 * ba->P + (size_t)ba->P * 36 + 36);
  R.smem = smem;
  if (smem > 64 * 1024) SVS_HIP(ctx, hipFuncSetAttribute((const void *)ba_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const bool speculate = prm.num_iters >= 1 && prm.max_trials > 1 && !ba->opt.debug && !ba->opt.no_speculation;
  R.speculate = speculate;
  if (speculate) {
    const int n_it = prm.num_iters;
    const size_t n_ctl = 8 + 8 * (size_t)n_it;
    if (ba->ctl_iters < n_it) {
      if (ba->d_ctl) (void)hipFree(ba->d_ctl);
      if (ba->h_ctl) (void)hipHostFree(ba->h_ctl);
      ba->d_ctl = ba->h_ctl = nullptr;
      SVS_HIP(ctx, hipMalloc(&ba->d_ctl, sizeof(double) * n_ctl));
      SVS_HIP(ctx, hipHostMalloc((void **)&ba->h_ctl, sizeof(double) * n_ctl, hipHostMallocDefault));
      ba->ctl_iters = n_it;
    }
    while (ba->spec_ev.size() < 6 * (size_t)n_it) { hipEvent_t e; SVS_HIP(ctx, hipEventCreate(&e)); ba->spec_ev.push_back(e); }
    for (size_t i = 0; i < n_ctl; ++i) ba->h_ctl[i] = 0.0;
    ba->h_ctl[0] = lambda;
    auto enqueue_all = [&]() -> int {
      SVS_HIP(ctx, hipMemcpyAsync(ba->d_ctl, ba->h_ctl, sizeof(double) * n_ctl, hipMemcpyHostToDevice, ctx->stream));
      int cur = ba->cur;
      for (int j = 0; j < n_it; ++j) {
        int rc = enqueue_trial(ba, lambda, cur, ba->d_ctl, &ba->spec_ev[6 * j], smem, allreduce, user);
        if (rc) return rc;
        BaDev B = make_dev(ba, lambda, cur, ba->d_ctl);
        hipLaunchKernelGGL(ba_lm_kernel, dim3(1), dim3(64), 0, ctx->stream, B, j);      // (taking this decision in the last workgroup of the
        SVS_LAUNCH_CHECK(ctx);                                                           //  previous kernel costs more than the launch: 400 same-address tickets)
        cur = 1 - cur;                   // as if accepted
      }
      SVS_HIP(ctx, hipMemcpyAsync(ba->h_ctl, ba->d_ctl, sizeof(double) * n_ctl, hipMemcpyDeviceToHost, ctx->stream));
      return SVS_OK;
    };
    // Replay from a graph where the launch sequence is a function of the resident problem alone: no collective callback between the kernels, no event brackets, and
    // not the multi-workgroup solves (they go through the process-wide spin gate: a mutex and events of other contexts, nothing a recording can hold).
    const bool graphable = !allreduce && !ba->timing && !ba->opt.no_graph && !(ba->grid_G > 0 && !ba->use_lds_solve);
    if (!graphable) return enqueue_all();
    // everything a recorded launch bakes in: the device view of the problem (pointers, counts, lambda) + the solver's configuration + the control blocks
    uint64_t sig = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t nbytes) { const unsigned char *b = static_cast<const unsigned char *>(p); for (size_t i = 0; i < nbytes; ++i) { sig ^= b[i]; sig *= 1099511628211ull; } };
    { BaDev B0 = make_dev(ba, lambda, ba->cur, ba->d_ctl); mix(&B0, sizeof B0); }
    const uint64_t cfg[] = {(uint64_t)ba->P, (uint64_t)ba->nw_sched, (uint64_t)ba->opt.nw, (uint64_t)ba->opt.nw4, (uint64_t)ba->use_fused_solve, (uint64_t)ba->use_lds_solve,
                            (uint64_t)ba->fuse_P1, (uint64_t)ba->fuse_lds_panel, (uint64_t)ba->lds_solve_smem, (uint64_t)ba->env_R, (uint64_t)ba->perm_active, (uint64_t)n_it,
                            (uint64_t)smem, (uint64_t)ba->red_count, (uint64_t)(uintptr_t)ba->d_x, (uint64_t)(uintptr_t)ba->d_upanel, (uint64_t)(uintptr_t)ba->d_rowmax2,
                            (uint64_t)(uintptr_t)ba->d_rowmax, (uint64_t)(uintptr_t)ba->d_linv, (uint64_t)(uintptr_t)ba->d_colmin, (uint64_t)(uintptr_t)ba->d_perm,
                            (uint64_t)(uintptr_t)ba->d_perm_sys, (uint64_t)(uintptr_t)ba->d_xfer, (uint64_t)(uintptr_t)ba->d_flags, (uint64_t)(uintptr_t)ba->d_ctl,
                            (uint64_t)(uintptr_t)ba->h_ctl, (uint64_t)(uintptr_t)ba->d_red, (uint64_t)(uintptr_t)ba->d_scal, (uint64_t)(uintptr_t)ctx->stream};
    mix(cfg, sizeof cfg);
    svs_ba::Graph &G = ba->graph[ba->cur & 1];
    // a layout is recorded when it comes back: a sliding window whose observation count changes with every call (svs_ba_window_update) would otherwise pay a recording
    // (capture + instantiation, ~0.2 ms) per call and never replay it
    if ((!G.exec || G.sig != sig) && G.seen != sig) { G.seen = sig; return enqueue_all(); }
    if (!G.exec || G.sig != sig) {
      if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
      hipGraph_t graph = nullptr;
      SVS_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
      const int rc = enqueue_all();
      const hipError_t e_end = hipStreamEndCapture(ctx->stream, &graph);
      if (rc || e_end != hipSuccess || !graph) {      // a recording that failed leaves nothing behind: this call (and the next) takes the kernel-by-kernel path
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        ba->opt.no_graph = 1;
        return rc ? rc : enqueue_all();
      }
      const hipError_t e_inst = hipGraphInstantiate(&G.exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e_inst != hipSuccess) { G.exec = nullptr; (void)hipGetLastError(); ba->opt.no_graph = 1; return enqueue_all(); }
      G.sig = sig;
      ++ba->n_graph_captures;
    }
    SVS_HIP(ctx, hipGraphLaunch(G.exec, ctx->stream));
    ++ba->n_graph_launches;
  }
  return SVS_OK;
}
static int optimize_finish(svs_ba *ba, OptRun &R, svs_ba_stats *stats) {
  svs_ctx *ctx = ba->ctx;
  SVS_DEVICE(ctx);
  const svs_ba_params &prm = ba->prm;
  svs_allreduce_fn allreduce = R.allreduce; void *user = R.user;
  double lambda = R.lambda, ni = R.ni;
  svs_ba_stats st = R.st;
  const size_t smem = R.smem;
  bool ok = true;
  int it = 0;
  bool resume = false;                 // the first trial of iteration `it` was already run (and rejected) by the speculative phase
  double r_rho = 0, r_chi = 0;
  if (R.speculate) {
    const int n_it = prm.num_iters;
    SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < n_it; ++j) {
      const double *rec = ba->h_ctl + 8 + 8 * j;
      if (rec[7] == 0.0) break;        // not executed: an earlier trial raised the abort flag
      if (rec[6] == 2.0) { ctx->err = "svs_ba_optimize: the workgroups of the multi-workgroup solve could not synchronise (device shared with other work): nothing was applied, retry"; if (stats) *stats = st; return SVS_ERR_BUSY; }
      { int rc = add_trial_times(ba, &ba->spec_ev[6 * j]); if (rc) return rc; }
      ++st.trials;
      if (j == 0) st.chi2_init = rec[0];
      if (rec[4] != 0.0) {             // accepted on the device: this iteration is complete
        lambda = rec[5]; ni = 2; ++st.accepted;
        ba->cur = 1 - ba->cur;
        ++st.iterations;
        st.chi2_final = rec[1];
        it = j + 1;
      } else {                         // rejected (or rho == 0 / not finite): same bookkeeping as the host path, then resume there
        lambda *= ni; ni *= 2;
        resume = true; r_rho = rec[3]; r_chi = rec[0];
        it = j;
        break;
      }
    }
  }
  for (; it < prm.num_iters && ok; ++it) {
    if (it == 0 && !resume) { lambda = prm.lambda_init; ni = 2; }
    double rho = 0, currentChi = 0;
    int qmax = 0;
    bool skip = resume;
    if (resume) { rho = r_rho; currentChi = r_chi; qmax = 1; resume = false; }
    do {
      if (skip) { skip = false; continue; }      // `continue` in a do-while jumps to the condition: the speculative phase ran this trial
      int rc = enqueue_trial(ba, lambda, -1, nullptr, ba->ev, smem, allreduce, user);
      if (rc) return rc;
      if (!ba->h_scal) SVS_HIP(ctx, hipHostMalloc((void **)&ba->h_scal, sizeof(double) * SC_N, hipHostMallocDefault));
      double *h = ba->h_scal;
      SVS_HIP(ctx, hipMemcpyAsync(h, ba->d_scal, sizeof(double) * SC_N, hipMemcpyDeviceToHost, ctx->stream));
      SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      { int rc2 = add_trial_times(ba, ba->ev); if (rc2) return rc2; }
      if (ba->opt.debug)
        fprintf(stderr, "[svs_ba] solve phases: init %.1f us, forward %.1f us (pivot wave: load+row update %.1f, eliminate+emit %.1f, - %.1f, barrier wait %.1f), backward %.1f us\n",
                h[5], h[6], h[8], h[9], h[10], h[11], h[7]);
      if (h[3] == 2.0) { ctx->err = "svs_ba_optimize: the workgroups of the multi-workgroup solve could not synchronise (device shared with other work): this trial was not applied, retry"; st.lambda_final = lambda; if (stats) *stats = st; return SVS_ERR_BUSY; }
      const bool fail = h[3] != 0.0;
      if (qmax == 0) { currentChi = h[4]; if (it == 0) st.chi2_init = currentChi; }
      double chi_t = 0, scale_l = 0;
      for (int i = 0; i < SC_SLOTS; ++i) { chi_t += h[SC_CHI + i]; scale_l += h[SC_SCL + i]; }
      double tempChi = fail ? 1.7976931348623157e308 : chi_t;
      rho = currentChi - tempChi;
      double scale = scale_l + h[2] + 1e-3;
      rho /= scale;
      ++st.trials;
      if (rho > 0 && std::isfinite(tempChi)) {
        const double q = 2 * rho - 1;
        double alpha = 1. - q * q * q;                // same expression as ba_lm_kernel (g2o: pow(2 rho - 1, 3), <= 1 ulp apart)
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2; currentChi = tempChi; ++st.accepted;
        ba->cur = 1 - ba->cur;                      // discardTop: trial state becomes current
        // keep the other buffer's untouched landmarks coherent: psi_trial of landmarks without
        // edges on this rank was never written, but both buffers started equal and only owned
        // landmarks ever change, so nothing to copy.
      } else {
        lambda *= ni; ni *= 2;                      // pop: current state stays
      }
      ++qmax;
    } while (rho < 0 && qmax < prm.max_trials);
    ++st.iterations;
    st.chi2_final = currentChi;
    if (qmax == prm.max_trials || rho == 0) { ok = false; st.terminated = 1; }
  }
  st.lambda_final = lambda;
  if (stats) *stats = st;
  return SVS_OK;
}

extern "C" int svs_ba_optimize(svs_ba *ba, svs_allreduce_fn allreduce, void *user, svs_ba_stats *stats) {
  OptRun R;
  int rc = optimize_begin(ba, allreduce, user, R);
  if (rc) return rc;
  return optimize_finish(ba, R, stats);
}

// Throughput mode: n independent windows, each on its own context (= its own stream), all enqueued before the first wait -- the
// kernels of different windows overlap on the device (a 50-keyframe window keeps 2 of 256 CUs busy during its solve).
extern "C" int svs_ba_optimize_batch(svs_ba *const *bas, int n, svs_ba_stats *stats) {
  if (!bas || n < 1) return SVS_ERR_INVALID;
  std::vector<OptRun> runs((size_t)n);
  std::vector<char> alone((size_t)n, 0);
  for (int i = 0; i < n; ++i) {
    if (!bas[i]) return SVS_ERR_INVALID;
    for (int j = 0; j < i; ++j) if (bas[j]->ctx == bas[i]->ctx) { bas[i]->ctx->err = "svs_ba_optimize_batch: every window needs its own context (stream)"; return SVS_ERR_INVALID; }
    // a window on the multi-workgroup solve needs (nearly) all CUs resident at once for its grid-wide arrivals: two of them in flight, or one next to
    // the Schur kernels of other windows, would stall each other until the bounded spin gives up -- such windows run by themselves, after the others
    alone[i] = !bas[i]->use_fused_solve && !bas[i]->use_lds_solve && bas[i]->grid_G > 0;
    if (alone[i]) continue;
    const int rc = optimize_begin(bas[i], nullptr, nullptr, runs[i]);
    if (rc) return rc;
  }
  int rc_all = SVS_OK;
  for (int i = 0; i < n; ++i) {
    if (alone[i]) continue;
    const int rc = optimize_finish(bas[i], runs[i], stats ? stats + i : nullptr);
    if (rc && !rc_all) rc_all = rc;
  }
  for (int i = 0; i < n; ++i) {
    if (!alone[i]) continue;
    int rc = optimize_begin(bas[i], nullptr, nullptr, runs[i]);
    if (!rc) rc = optimize_finish(bas[i], runs[i], stats ? stats + i : nullptr);
    if (rc && !rc_all) rc_all = rc;
  }
  return rc_all;
}

extern "C" int svs_ba_get_state(svs_ba *ba, double *h_poses, double *h_psi) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && ba->d_poses[0] && ba->problem_valid);
  SVS_DEVICE(ctx);
  // both arrays land in a pinned area of the library's own, the two copies back to back, ONE wait; then a host copy into the caller's (pageable) arrays.  Straight
  // into pageable memory every copy is a synchronous, staged transfer of its own: 40 us of host turn-around between them in the drop-in call's timeline
  const size_t nb_poses = h_poses ? sizeof(double) * 12 * (size_t)ba->P : 0, nb_psi = (h_psi && ba->L) ? sizeof(double) * 3 * (size_t)ba->L : 0;
  if (nb_poses + nb_psi > ba->h_state_cap) {
    if (ba->h_state) (void)hipHostFree(ba->h_state);
    ba->h_state = nullptr; ba->h_state_cap = 0;
    const size_t want = (nb_poses + nb_psi) + (nb_poses + nb_psi) / 4 + 4096;
    SVS_HIP(ctx, hipHostMalloc((void **)&ba->h_state, want, hipHostMallocDefault));
    ba->h_state_cap = want;
  }
  if (nb_poses) SVS_HIP(ctx, hipMemcpyAsync(ba->h_state, ba->d_poses[ba->cur], nb_poses, hipMemcpyDeviceToHost, ctx->stream));
  if (nb_psi) SVS_HIP(ctx, hipMemcpyAsync(ba->h_state + nb_poses, ba->d_psi[ba->cur], nb_psi, hipMemcpyDeviceToHost, ctx->stream));
  SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (nb_poses) std::memcpy(h_poses, ba->h_state, nb_poses);
  if (nb_psi) std::memcpy(h_psi, ba->h_state + nb_poses, nb_psi);
  return SVS_OK;
}

extern "C" int svs_ba_set_comm(svs_ba *ba, svs_comm *comm) {
  if (!ba) return SVS_ERR_INVALID;
  ba->comm = comm;
  ba->profile_ready = false;            // the co-visibility pattern must be exchanged over the new communicator
  return SVS_OK;
}

extern "C" int svs_ba_set_option(svs_ba *ba, const char *name, int value) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && name);
  auto clamp = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
  BaOptions &o = ba->opt;
  const std::string n(name);
  if (n == "no_speculation") o.no_speculation = value != 0;
  else if (n == "no_order") o.no_order = value != 0;
  else if (n == "one_front") o.one_front = value != 0;
  else if (n == "no_fused_solve") o.no_fused_solve = value != 0;
  else if (n == "no_lds_panel") o.no_lds_panel = value != 0;
  else if (n == "no_graph") o.no_graph = value != 0;
  else if (n == "no_lds_solve") o.no_lds_solve = value != 0;
  else if (n == "no_fused_cons") o.no_fused_cons = value != 0;
  else if (n == "no_grid_solve") o.no_grid_solve = value != 0;
  else if (n == "no_tile_solve") o.no_tile_solve = value != 0;
  else if (n == "debug") o.debug = clamp(value, 0, 2);
  else if (n == "nw") o.nw = value == 0 ? 0 : clamp(value, 4, 8);
  else if (n == "p1") o.p1 = value < 0 ? -1 : clamp(value, 0, SOLVE_MAX_P);
  else if (n == "group") o.group = value == 0 ? 0 : clamp(value, 1, WIN);
  else if (n == "grid_g") o.grid_g = clamp(value, 0, 1024);
  else if (n == "host_threads") o.host_threads = clamp(value, 0, 64);
  else if (n == "host_marshal") o.host_marshal = clamp(value, 0, 2);
  else SVS_REQUIRE(ctx, !"unknown option");
  ba->profile_ready = false;            // solve-kernel choice depends on the switches
  return SVS_OK;
}

extern "C" int svs_ba_graph_stats(svs_ba *ba, int64_t *launches, int64_t *captures) {
  if (!ba) return SVS_ERR_INVALID;
  if (launches) *launches = ba->n_graph_launches;
  if (captures) *captures = ba->n_graph_captures;
  return SVS_OK;
}
extern "C" int svs_ba_info(svs_ba *ba, int32_t *solve_kind, int32_t *envelope_rows, int32_t *n_chunks, int32_t *n_wide) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && ba->problem_valid);
  if (!ba->profile_ready) { const int rc = ensure_profile(ba, ba->comm ? svs_comm_allreduce_hook : nullptr, ba->comm); if (rc) return rc; }
  if (solve_kind) *solve_kind = ba->use_fused_solve ? (ba->fuse_P1 > 0 ? 3 : 2) : (ba->use_lds_solve ? 1 : (ba->grid_G > 0 ? (ba->tiles_G > 0 ? 5 : 4) : 0));
  if (envelope_rows) *envelope_rows = ba->env_R;
  if (n_chunks) *n_chunks = ba->n_chunks;
  if (n_wide) *n_wide = ba->n_wide;
  return SVS_OK;
}

extern "C" int svs_ba_order_info(svs_ba *ba, int32_t *ordered, int32_t *envelope_rows_caller_order, int32_t *h_perm) {
  svs_ctx *ctx = ba ? ba->ctx : nullptr;
  SVS_REQUIRE(ctx, ba && ba->problem_valid);
  if (!ba->profile_ready) { const int rc = ensure_profile(ba, ba->comm ? svs_comm_allreduce_hook : nullptr, ba->comm); if (rc) return rc; }
  if (ordered) *ordered = ba->perm_active ? 1 : 0;
  if (envelope_rows_caller_order) *envelope_rows_caller_order = ba->env_R_natural;
  if (h_perm) for (int k = 0; k < ba->P; ++k) h_perm[k] = ba->perm_active ? ba->h_perm[k] : k;
  return SVS_OK;
}

extern "C" int svs_ba_set_timing(svs_ba *ba, int on) {
  if (!ba) return SVS_ERR_INVALID;
  ba->timing = on != 0;
  return SVS_OK;
}

extern "C" int svs_ba_kernel_times(svs_ba *ba, float *reduce_ms, float *solve_ms, float *backsub_ms, int32_t *n_launches) {
  if (!ba) return SVS_ERR_INVALID;
  if (reduce_ms) *reduce_ms = ba->t_reduce;
  if (solve_ms) *solve_ms = ba->t_solve;
  if (backsub_ms) *backsub_ms = ba->t_backsub;
  if (n_launches) *n_launches = ba->n_reduce;
  return SVS_OK;
}
