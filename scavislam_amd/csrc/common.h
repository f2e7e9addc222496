// common.h -- context, error plumbing and wave64 helpers shared by the gfx950 kernels.
// Target: MI355X (gfx950, CDNA4, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/scavislam_hip.h"

struct svs_ctx {
  int device = 0;
  int n_cu = 256;                 // compute units of the device (MI355X: 256)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  // device scratch owned by the context (block partials, cross-workgroup hand-off words): grown on demand, freed with the ctx
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
  // the matcher's per-call tables (relative poses per (stream, keyframe), predictions per point): its own buffer, because the trackers'
  // scratch above carries hand-off words across the launches of one call chain
  void *match_scratch = nullptr;
  size_t match_scratch_bytes = 0;
  // switches read ONCE at svs_ctx_create (debug / experiment only; never per call)
  int trk_nwg = 0;            // SVS_TRK_NWG: workgroups per stream of the latency-mode quarter-grid tracker (0 = automatic)
  int trk_balance = 1;        // big batches (dense.hip, BAL): 1 = grid order by the last frame's LM work, 2 = also 2..4 workgroups for the longest streams (experimental), 0 = stream order
  int trk_regs = 0;           // SVS_TRK_ONE_PER_CU (1) / SVS_TRK_TWO_PER_CU (2): register budget of that tracker (0 = automatic)
  int full_nwg = 0;           // SVS_FULL_NWG: workgroups per stream of the full-resolution tracker (0 = automatic)
  int match_legacy = 0;       // "match_legacy": 0 = four points per wave (match_kernel3), 1 = the round-1/2 kernel (one wave per point, ballots), 2 = one wave per point with the lean scan
  int mo_spec = 1;            // "mo_spec": calcFastMotionOnly's trials behind a rejection run a chi2-only sweep (the full one follows if the trial is accepted after all); 0: always full
  int fe_fuse_tail = 1;       // "fe_fuse_tail": the one-call front end runs the gate and the dense clouds inside the refinement kernel (0: four more launches)
  int match_order = 1;        // "match_order": match_kernel3 takes the points of a stream in image order (counting sort by cell of the predicted position), 0: list order
  int fe_pipeline = 1;        // "fe_pipeline" (read at svs_frontend_create and per call): svs_frontend_process_frames on caller-owned device frames builds the pyramid
                              // of frame N+1 on the side stream while frame N's pose refinement / gate / cloud run (frontend.hip).  Results identical.
  int fe_overlap = 1;         // "fe_overlap": the one-call front end runs FAST / block matching on a side stream beside the dense tracker (0: one stream)
  int trk_seq_chi2 = 0;       // "trk_seq_chi2": the quarter-grid tracker decides accept / reject on the reference's own sequential f32 chi2 sums (dense.hip; slow: parity runs)
  int trk_lazy_chi2 = 1;      // "trk_lazy_chi2" (default): the same decisions at full speed -- the f64 sums decide wherever their difference is outside the rigorous error bound of the
                              // reference's float sums, and inside it the float sums are formed bit for bit without the sequential chain (seqsum.h).  0: f64 sums alone (rounds 1-4)
  int trk_flat = 1;           // "trk_flat": big batches run the flat state-machine tracker kernel (dense.hip, round 6: the sweep inlined, LM state in LDS); 0: the round-5 kernel (sweep as a call) -- same bits
  int trk_split = 10;         // "trk_split" (with trk_flat): K -- a stream of a big batch that is still iterating after K trials on the finest level parks and is finished by a
                              // second launch with several workgroups per stream (dense.hip: the continuation launch); 0: off.  Measured on the bench batch (512
                              // streams): tracker stage 1.40 ms unsplit, 1.30 at K = 8..9, 1.33 at K = 10; the whole step 2.66-2.69 unsplit, 2.75 at K = 8 (the side
                              // stream's FAST no longer finds the tail to run in), 2.63 at K = 10
  int trk_cont_slots = 0;     // "trk_cont_slots": workgroup slots the continuation launch sizes itself for (0: two per CU)
  // set by the one-call front end around svs_match (few keyframes): the tracked pose / the active keyframe's pose per stream, from which the matcher's prediction kernel forms
  // T_cur_from_w / T_w_from_actkey and the per-keyframe relative poses itself (match.hip) -- two small launches less between tracker and matcher
  const double *match_src_T = nullptr, *match_src_Ta = nullptr;
  void *seq_buf = nullptr; size_t seq_buf_bytes = 0;      // the per-pass term buffers of both modes
  void *seq_stats = nullptr;                              // device: [0] exact float sums formed, [1] of those by the fallback chain (svs_ctx_get_stat)
  hipEvent_t spin_ev = nullptr;      // "a device-filling kernel of mine has finished" (svs_spin_enter / svs_spin_leave)
  hipEvent_t lane_ev = nullptr;      // the same behind my latest priority-lane launch
  bool spin_lane = false;            // the launch between the last svs_spin_enter and its svs_spin_leave took the priority lane
  int spin_demand = 0;               // compute units that launch may hold (one per workgroup)
  long long spin_n_lane = 0, spin_n_gated = 0;      // svs_ctx_get_stat "spin_lane_launches" / "spin_gated_launches"
  int xcd_swizzle = 1;        // "xcd_swizzle": tile kernels whose neighbouring tiles share image lines (FAST score, block matching, ...) hand every XCD a CONTIGUOUS range of
                              // the linear workgroup index (xcd_contiguous below) so the shared lines are fetched into one L2, not into 2-3; 0: the dispatcher's round robin (A/B)
  int mo_legacy = 0;          // "mo_legacy": the record-walking motion-only kernel of rounds 1-2 instead of the fused one (A/B experiments)
};
// Kernels whose workgroups wait for each other INSIDE one launch (the latency-mode trackers, the multi-workgroup Cholesky) size their grids to a device they have
// to themselves: every workgroup must become resident while its siblings spin.  Next to ordinary kernels of other contexts that always happens -- those finish in
// bounded time and free their slots -- but two such kernels from two contexts can each hold part of the device and starve the other's missing workgroups until the
// bounded spins give up (SVS_ERR_BUSY).  The reference runs exactly this concurrency: the front end on the main thread, optimize + re-registration on the backend
// thread (stereo_slam.cpp:196, backend.cpp:157-224), so the library keeps at most ONE of them on the device at a time: a launch between svs_spin_enter / svs_spin_leave
// first makes its stream wait for the event the previous device-filling launch of ANOTHER context left behind (stream-side; the host does not block), then leaves its own.
// n_workgroups: the REAL size of the launch (0: the whole device).  Small launches (<= 16 workgroups) skip the gate while they provably fit beside everything of this
// kind that is in flight (image.hip: the invariant and its book-keeping).
int svs_spin_enter(svs_ctx *ctx, int n_workgroups = 0);
int svs_spin_leave(svs_ctx *ctx);
// the pair as a scope.  Normal path: `if (int rc = gate.leave()) return rc;` right behind the launch (the event record can fail, and then the NEXT context's launch would
// run ungated without anybody knowing); the destructor is the safety net of the early returns only.
struct SvsSpinScope {
  svs_ctx *ctx; int rc; bool open;
  SvsSpinScope(svs_ctx *c, int n_workgroups = 0) : ctx(c), rc(svs_spin_enter(c, n_workgroups)), open(rc == 0) {}
  int leave() { if (!open) return SVS_OK; open = false; return svs_spin_leave(ctx); }
  ~SvsSpinScope() { if (open) (void)svs_spin_leave(ctx); }
  SvsSpinScope(const SvsSpinScope &) = delete;
  SvsSpinScope &operator=(const SvsSpinScope &) = delete;
};
// returns ctx-owned device scratch of at least `bytes` (contents undefined); may synchronise the stream when it has to grow
int svs_ctx_scratch(svs_ctx *ctx, size_t bytes, void **out);
int svs_ctx_match_scratch(svs_ctx *ctx, size_t bytes, void **out);
// svs_dense_track_cpu_sem with the state of the balanced launch of big batches (dense.hip: per-stream LM work of the last frame -> workgroups per stream);
// d_bal_state: svs_dense_track_balance_bytes(batch) bytes of device memory, initialised once by svs_dense_track_balance_init; may be NULL
size_t svs_dense_track_balance_bytes(int batch);
const int *svs_dense_track_balance_order(svs_ctx *ctx, void *d_bal_state, int batch);
int svs_dense_track_balance_init(svs_ctx *ctx, void *d_bal_state, int batch);
int svs_dense_track_cpu_sem_work(svs_ctx *ctx, const svs_dense_track_args *a, double *d_T_io, int32_t *d_passes_out, int batch, void *d_bal_state);
// every entry point that allocates or launches runs on the context's device, whatever the calling thread's current device is
#define SVS_DEVICE(ctx) SVS_HIP(ctx, hipSetDevice((ctx)->device))

#define SVS_HIP(ctx, call)                                                              \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      char buf_[512];                                                                   \
      snprintf(buf_, sizeof buf_, "%s:%d %s -> %s", __FILE__, __LINE__, #call,          \
               hipGetErrorString(e_));                                                  \
      (ctx)->err = buf_;                                                                \
      return SVS_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define SVS_REQUIRE(ctx, cond)                                                          \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      char buf_[512];                                                                   \
      snprintf(buf_, sizeof buf_, "%s:%d requirement failed: %s", __FILE__, __LINE__,   \
               #cond);                                                                  \
      if (ctx) (ctx)->err = buf_;                                                       \
      return SVS_ERR_INVALID;                                                           \
    }                                                                                   \
  } while (0)

#define SVS_LAUNCH_CHECK(ctx) SVS_HIP(ctx, hipGetLastError())

// internal: svs_pyr_down_u8 that also copies its source image (the front end's copy-in of a caller's device frame rides on the first pyramid step)
int svs_pyr_down_u8_copy(svs_ctx *ctx, const uint8_t *d_src, int w, int h, int sstride, size_t s_bstride, uint8_t *d_dst, int dstride, size_t d_bstride,
                         uint8_t *d_copy, int cstride, size_t c_bstride, int batch);
// internal (not exported through the header): svs_process_matched_points with the record count of the new-feature lists per stream, on the device
// the work a stream's refinement workgroup does after its LM loop when the front end fuses the stages (dense.hip: motion_only_fused_kernel<true>)
struct svs_mo_tail {
  const svs_candidate_point *pts; size_t pts_b; const int32_t *n_new; float mre; svs_gated_point *gated; size_t gated_b; svs_point_stats *ptstats;      // processMatchedPoints' gate
  const float *disp; int ds; size_t disp_b; svs_cam cams[3]; float *cloud[3]; size_t cloud_b[3];                                                     // computeDensePointCloudCpu, 3 levels
  const int *order;                                                                                                                                  // optional: workgroup -> stream << 4 (a permutation)
};
int svs_motion_only_gate_cloud(svs_ctx *ctx, const svs_match_result *d_results, int n, size_t res_bstride, const svs_cam *cam, const svs_pose_opt_params *prm,
                               double *d_T_io, svs_pose_opt_stats *d_stats, const svs_mo_tail *tail, int batch);
int svs_process_matched_points_dev(svs_ctx *ctx, const svs_match_result *d_results, const svs_candidate_point *d_pts, int n, size_t res_bstride,
                                   size_t pts_bstride, const int32_t *d_n_new_records, const svs_cam *cam, const double *d_T, float max_reproj_error,
                                   svs_gated_point *d_gated, size_t gated_bstride, svs_point_stats *d_stats, int batch);

// internal: the dense clouds of the three levels in one launch (dense.hip)
int svs_pointcloud_cpu_sem_levels(svs_ctx *ctx, const float *d_disp, int disp_stride, size_t disp_bstride, const svs_cam *cams, const double *d_T, float *const *d_cloud,
                                  const size_t *cloud_bstride, int batch);

__host__ __device__ static inline int div_up(int a, int b) { return (a + b - 1) / b; }

// Workgroup b of a launch is observed on XCD b % 8 (nothing promises it: speed only, never correctness).  Kernels whose neighbouring tiles share cache lines take their tile from
// this bijective remap of the linear workgroup index instead: XCD x works through the contiguous range [x * n / 8, (x + 1) * n / 8) in dispatch order, so the lines two neighbours
// share arrive in ONE L2 (any n, also n % 8 != 0).
__device__ __forceinline__ unsigned xcd_contiguous(unsigned id, unsigned n) {
  const unsigned q = n >> 3, r = n & 7u, x = id & 7u, k = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// ---- wave64 reductions (DPP/bpermute via __shfl; no LDS, no volatile warp idioms) -----------
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
