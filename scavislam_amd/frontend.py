"""Host-side mirror of the reference's per-frame front-end call surfaces, on top of the C ABI.

Same names / argument meaning as the reference (so the parity tests read like calls into
stereo_frontend.cpp), device memory held in torch tensors:

  FramePyramid.preprocessing()           <- FrameGrabber::preprocessing   frame_grabber.cpp:285-336
  FastGrid.detectAdaptively / .detect    <- FastGrid                      fast_grid.h:40-51
  GuidedMatcher.match                    <- GuidedMatcher<StereoCamera>   matcher.hpp:67-83
  DenseTracker.denseTrackingCpu / computeDensePointCloudCpu               dense_tracking.h:59-79
  GpuTracker.jacobianReduction / chi2 / computePointCloud                 gpu/dense_tracking.cuh:281-342
  GpuFrameData.preprocessing()           <- FrameGrabber::preprocessing, CUDA branch  frame_grabber.cpp:291-313
  DenseTrackerGpu.denseTrackingGpu / computeDensePointCloudGpu            dense_tracking.cpp:60-215
  StereoMatcher.calcDisparityCpu         <- StereoFrontend::calcDisparityCpu  stereo_frontend.cpp:620-653
  PoseOptimizer.calcFastMotionOnly       <- BA_SE3_XYZ_STEREO::calcFastMotionOnly  pose_optimizer.h:134-298

The HIP library does all the arithmetic; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .ctypes_types import (CANDIDATE_DTYPE, DENSE_LM_RECORD_DTYPE, DENSE_SUMS_DTYPE, GATED_POINT_DTYPE, KEYFRAME_DTYPE, MATCH_RESULT_DTYPE,
                           POINT_STATS_DTYPE, Cam,
                           FastGrid as FastGridPOD, PoseOptParams, PoseOptStats, StereoParams, level_cams)

NUM_PYR_LEVELS = 3  # global.h:107


def _pose_mul(A, B):
    """3x4 pose product with the operation order of the C code on both sides of the boundary ((a0 b0 + a1 b1) + a2 b2,
    then + t; no BLAS, no FMA): the matcher's truncating warp makes the last bit of the pose visible once in ~10^3 points."""
    A, B = [float(v) for v in np.asarray(A, np.float64).reshape(12)], [float(v) for v in np.asarray(B, np.float64).reshape(12)]
    out = [0.0] * 12
    for i in range(3):
        for j in range(4):
            out[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j]
        out[4 * i + 3] += A[4 * i + 3]
    return np.array(out)


def _pose_inv(A):
    A = [float(v) for v in np.asarray(A, np.float64).reshape(12)]
    out = [0.0] * 12
    for i in range(3):
        for j in range(3):
            out[4 * i + j] = A[4 * j + i]
    for i in range(3):
        out[4 * i + 3] = -(out[4 * i] * A[3] + out[4 * i + 1] * A[7] + out[4 * i + 2] * A[11])
    return np.array(out)


def _round_up(a, b):
    return (a + b - 1) // b * b


def fastgrid_for_level(w, h, level):
    """Grid parameters of StereoFrontend::initialize (stereo_frontend.cpp:73-88) + the FastGrid
    constructor (fast_grid.cpp:23-58).  Host-side integer bookkeeping, no image arithmetic."""
    dim = max(3 - int(level * 0.5), 1)
    inv_fac = 1.0 / (1 << level)
    total = int(2000 * inv_fac * inv_fac)
    per_cell = total // (dim * dim)
    bound = max(per_cell // 3, 10)
    g = FastGridPOD()
    g.gx = g.gy = dim
    g.min_inner = int(per_cell - bound * 0.33)
    g.min_outer = per_cell - bound
    g.max_inner = int(per_cell + bound * 0.33)
    g.max_outer = per_cell + bound
    g.cell_w, g.cell_h = w // dim, h // dim
    g.fast_min, g.fast_max = 10, 40
    for i in range(len(g.thr)):
        g.thr[i] = 25
    return g


class FramePyramid:
    """Device-resident frame data of `batch` independent camera streams (FrameData<StereoCamera>,
    frame_grabber.hpp:93-155): u8 pyramid, f32 pyramid + Sobel images, f32 disparity."""

    def __init__(self, ctx, stream, cam, batch=1, with_float=True):
        self.ctx, self.stream, self.cam, self.batch = ctx, stream, cam, batch
        self.cams = level_cams(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"], NUM_PYR_LEVELS)
        dev = torch.device("cuda", ctx.device)
        self.w = [self.cams[l].w for l in range(NUM_PYR_LEVELS)]
        self.h = [self.cams[l].h for l in range(NUM_PYR_LEVELS)]
        self.stride = [_round_up(w, 64) for w in self.w]
        with torch.cuda.stream(stream):
            self.pyr = [torch.zeros((batch, self.h[l], self.stride[l]), dtype=torch.uint8, device=dev)
                        for l in range(NUM_PYR_LEVELS)]
            self.disp = torch.zeros((batch, self.h[0], self.stride[0]), dtype=torch.float32, device=dev)
            self.f32 = self.dx = self.dy = None
            if with_float:
                self.f32 = [torch.zeros((batch, self.h[l], self.stride[l]), dtype=torch.float32, device=dev)
                            for l in range(NUM_PYR_LEVELS)]
                self.dx = [torch.zeros_like(t) for t in self.f32]
                self.dy = [torch.zeros_like(t) for t in self.f32]

    def bstride(self, l):
        return self.h[l] * self.stride[l]

    def upload(self, images, disp=None):
        """images: [batch,h,w] u8 (numpy); disp: [batch,h,w] f32 or None."""
        with torch.cuda.stream(self.stream):
            img = torch.as_tensor(np.ascontiguousarray(images)).to(self.pyr[0].device, non_blocking=False)
            self.pyr[0][:, :, :self.w[0]] = img.reshape(self.batch, self.h[0], self.w[0])
            if disp is not None:
                d = torch.as_tensor(np.ascontiguousarray(disp, dtype=np.float32)).to(self.disp.device)
                self.disp[:, :, :self.w[0]] = d.reshape(self.batch, self.h[0], self.w[0])

    def preprocessing(self, with_float=True):
        """FrameGrabber::preprocessing: cv::buildPyramid + (CPU path) convertTo/Sobel per level.
        with_float=False skips the f32/Sobel images (the dense tracker can form them on the fly
        from the u8 pyramid, DenseTracker.track_args(from_u8=True))."""
        for l in range(1, NUM_PYR_LEVELS):
            self.ctx.call("svs_pyr_down_u8", self.pyr[l - 1].data_ptr(), self.w[l - 1], self.h[l - 1],
                          self.stride[l - 1], self.bstride(l - 1), self.pyr[l].data_ptr(), self.stride[l],
                          self.bstride(l), self.batch)
        if self.f32 is not None and with_float:
            for l in range(NUM_PYR_LEVELS):
                self.ctx.call("svs_convert_sobel_f32", self.pyr[l].data_ptr(), self.w[l], self.h[l], self.stride[l],
                              self.bstride(l), self.f32[l].data_ptr(), self.dx[l].data_ptr(), self.dy[l].data_ptr(),
                              self.stride[l], self.bstride(l), self.batch)

    def level_host(self, l, slot=0):
        self.ctx.sync()
        return self.pyr[l][slot, :, :self.w[l]].cpu().numpy()

    def clone_pyramid(self):
        """Frame::clone (keyframes.h:72-83): deep copy of the u8 pyramid, kept on the device."""
        with torch.cuda.stream(self.stream):
            return [t.clone() for t in self.pyr]


class FastGrid:
    """FastGrid for all matching levels and `batch` threshold states (fast_grid.h:27-63)."""

    def __init__(self, ctx, frame, n_levels=NUM_PYR_LEVELS, corner_cap=8192, grids=None):
        self.ctx, self.frame, self.n_levels, self.cap = ctx, frame, n_levels, corner_cap
        self.grids = (FastGridPOD * n_levels)()
        for l in range(n_levels):
            self.grids[l] = grids[l] if grids is not None else fastgrid_for_level(frame.w[l], frame.h[l], l)
        w = (C.c_int32 * n_levels)(*frame.w[:n_levels])
        h = (C.c_int32 * n_levels)(*frame.h[:n_levels])
        self.h = C.c_void_p()
        ctx.check(ctx.lib.svs_fast_create(ctx.h, n_levels, w, h, self.grids, frame.batch, corner_cap, C.byref(self.h)))
        ctx.children.add(self)

    def _detect(self, pyr, trials, n_batch):
        n = self.n_levels
        imgs = (C.c_void_p * n)(*[pyr[l].data_ptr() for l in range(n)])
        strides = (C.c_int32 * n)(*self.frame.stride[:n])
        bstrides = (C.c_size_t * n)(*[self.frame.bstride(l) for l in range(n)])
        self.ctx.check(self.ctx.lib.svs_fast_detect(self.h, imgs, strides, bstrides, n_batch or self.frame.batch, trials))

    def detectAdaptively(self, pyr=None, trials=6, n_batch=None):
        """FastGrid::detectAdaptively(img, trials, qt) on every level (stereo_frontend.cpp:657-679)."""
        assert trials >= 1
        self._detect(pyr or self.frame.pyr, trials, n_batch)

    def detect(self, pyr=None, n_batch=None):
        """FastGrid::detect(img, cell_grid2d, qt): one pass at the stored thresholds."""
        self._detect(pyr or self.frame.pyr, 0, n_batch)

    def corners(self, slot=0, level=0):
        """(xy int16 [n,2] in quadtree insertion order, cell_count, emit_thr, thr_state)."""
        g = self.grids[level]
        nc = g.gx * g.gy
        xy = np.zeros((self.cap, 2), np.int16)
        n = C.c_int32()
        cc, et, ts = np.zeros(nc, np.int32), np.zeros(nc, np.int32), np.zeros(nc, np.int32)
        self.ctx.check(self.ctx.lib.svs_fast_download(self.h, slot, level, xy.ctypes.data, self.cap, C.byref(n),
                                                      cc.ctypes.data, et.ctypes.data, ts.ctypes.data))
        return xy[:n.value].copy(), cc, et, ts

    def set_thresholds(self, slot, level, thr):
        thr = np.ascontiguousarray(thr, np.int32)
        self.ctx.check(self.ctx.lib.svs_fast_set_thresholds(self.h, slot, level, thr.ctypes.data))

    def corner_bits(self, slot=0, level=0):
        """The level's corner bitmap as a bool image [h][w] (svs_fast_device_view, API 7): True = a corner of the last detection -- what
        GuidedMatcher::match tests its window positions against (the reference asks a quadtree of these corners, matcher.cpp:351-357)."""
        import torch
        d_bits, stride, bstride, colbits = C.c_void_p(), C.c_int32(), C.c_size_t(), C.c_int32()
        self.ctx.check(self.ctx.lib.svs_fast_device_view(self.h, level, C.byref(d_bits), C.byref(stride), C.byref(bstride), C.byref(colbits), None, None))
        self.ctx.sync()
        g, w, h = self.grids[level], self.frame.w[level], self.frame.h[level]
        raw = np.zeros((h, stride.value), np.uint8)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = hip.hipMemcpy(raw.ctypes.data, d_bits.value + slot * bstride.value, raw.nbytes, 2)      # hipMemcpyDeviceToHost
        assert rc == 0, rc
        bits = np.unpackbits(raw, axis=1, bitorder="little")
        x = np.arange(w)
        ci = np.minimum(x // g.cell_w, g.gx - 1)
        return bits[:, x + (colbits.value - g.cell_w) * ci].astype(bool)

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.lib.svs_fast_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GuidedMatcher:
    """GuidedMatcher<StereoCamera>::match (matcher.hpp:67-83)."""

    def __init__(self, ctx, frame, fast):
        self.ctx, self.frame, self.fast = ctx, frame, fast

    def match(self, keyframes, T_cur_from_actkey, T_actkey_from_w, points, search_radius=8, thr_mean=22,
              thr_std=10):
        """keyframes: list of (pyr device tensors [3] with frame strides, slot, T_anchor_from_w [12]);
        T_cur_from_actkey / T_actkey_from_w: [batch,12] (or [12]); points: CANDIDATE_DTYPE array
        [batch,n] or [n].  Returns MATCH_RESULT_DTYPE [batch,n] (appending to TrackData is host
        bookkeeping left to the caller, matcher.cpp:183-214)."""
        a = self.prepare(keyframes, T_cur_from_actkey, T_actkey_from_w, points, search_radius, thr_mean, thr_std)
        self.launch(a)
        return self.download()

    def prepare(self, keyframes, T_cur_from_actkey, T_actkey_from_w, points, search_radius=8, thr_mean=22, thr_std=10):
        """Upload the candidate points / keyframe table / poses once; returns the svs_match_args."""
        fr, B = self.frame, self.frame.batch
        dev = fr.pyr[0].device
        kfs = np.zeros(len(keyframes), KEYFRAME_DTYPE)
        for i, (pyr, slot, T) in enumerate(keyframes):
            kfs[i]["T_anchor_from_w"] = np.asarray(T, np.float64).reshape(12)
            for l in range(NUM_PYR_LEVELS):
                kfs[i]["pyr"][l] = pyr[l].data_ptr() + slot * fr.bstride(l)
                kfs[i]["stride"][l] = fr.stride[l]
        pts = np.ascontiguousarray(points, CANDIDATE_DTYPE)
        if pts.ndim == 1:
            pts = np.broadcast_to(pts, (B, len(pts))).copy()
        n = pts.shape[1]
        Tc = np.broadcast_to(np.asarray(T_cur_from_actkey, np.float64).reshape(-1, 12), (B, 12))
        Ta = np.broadcast_to(np.asarray(T_actkey_from_w, np.float64).reshape(-1, 12), (B, 12))
        T_cur_from_w = np.zeros((B, 12))
        T_w_from_actkey = np.zeros((B, 12))
        for b in range(B):  # two 3x4 pose products per frame: host bookkeeping (matcher.cpp:326-330)
            T_cur_from_w[b, :] = _pose_mul(Tc[b], Ta[b])
            T_w_from_actkey[b, :] = _pose_inv(Ta[b])
        with torch.cuda.stream(fr.stream):
            d_kfs = torch.as_tensor(kfs.view(np.uint8)).to(dev)
            d_pts = torch.as_tensor(pts.view(np.uint8).reshape(-1)).to(dev)
            d_Tc = torch.as_tensor(T_cur_from_w).to(dev)
            d_Ta = torch.as_tensor(T_w_from_actkey).to(dev)
            d_out = torch.zeros(B * n * MATCH_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        a = capi.MatchArgs()
        a.d_kfs, a.n_kf, a.d_pts, a.n_pts = d_kfs.data_ptr(), len(kfs), d_pts.data_ptr(), n
        a.d_T_cur_from_w, a.d_T_w_from_actkey = d_Tc.data_ptr(), d_Ta.data_ptr()
        for l in range(NUM_PYR_LEVELS):
            a.d_cur_pyr[l], a.cur_stride[l], a.cur_bstride[l] = fr.pyr[l].data_ptr(), fr.stride[l], fr.bstride(l)
            a.cam_vec[l] = fr.cams[l]
        a.d_disp, a.disp_stride, a.disp_bstride = fr.disp.data_ptr(), fr.stride[0], fr.bstride(0)
        a.search_radius, a.thr_mean, a.thr_std, a.n_batch = search_radius, thr_mean, thr_std, B
        self._keep = (d_kfs, d_pts, d_Tc, d_Ta, d_out, keyframes)
        self._n = n
        return a

    def launch(self, a):
        self.ctx.check(self.ctx.lib.svs_match(self.ctx.h, C.byref(a), self.fast.h, self._keep[4].data_ptr()))

    def download(self):
        self.ctx.sync()
        return self._keep[4].cpu().numpy().view(MATCH_RESULT_DTYPE).reshape(self.frame.batch, self._n)


class PoseOptimizer:
    """BA_SE3_XYZ_STEREO (PoseOptimizer<SE3,6,IdObs<3>,3>, pose_optimizer.h:486): motion-only refinement of
    T_cur_from_actkey over the matcher's TrackData, device resident (stereo_frontend.cpp:1058-1063)."""

    def __init__(self, ctx, frame):
        self.ctx, self.frame = ctx, frame
        dev = frame.pyr[0].device
        with torch.cuda.stream(frame.stream):
            self.d_T = torch.zeros((frame.batch, 12), dtype=torch.float64, device=dev)
            self.d_stats = torch.zeros(frame.batch * C.sizeof(PoseOptStats), dtype=torch.uint8, device=dev)

    def calcFastMotionOnly(self, matcher, T_cur_from_actkey, params=None, download=True):
        """obs_list / point_list = the status-OK results of `matcher` (device resident).  T: [batch,12] or [12]."""
        fr = self.frame
        prm = params or PoseOptParams.reference()
        T = np.broadcast_to(np.asarray(T_cur_from_actkey, np.float64).reshape(-1, 12), (fr.batch, 12))
        with torch.cuda.stream(fr.stream):
            self.d_T.copy_(torch.as_tensor(np.array(T, np.float64)))
        self.launch(matcher, prm)
        return self.download() if download else None

    def launch(self, matcher, prm):
        fr = self.frame
        self.ctx.call("svs_motion_only", matcher._keep[4].data_ptr(), matcher._n, matcher._n, C.byref(fr.cams[0]), C.byref(prm),
                      self.d_T.data_ptr(), self.d_stats.data_ptr(), fr.batch)

    def download(self):
        self.ctx.sync()
        T = self.d_T.cpu().numpy().reshape(self.frame.batch, 3, 4)
        raw = self.d_stats.cpu().numpy()
        stats = [PoseOptStats.from_buffer_copy(raw[i * C.sizeof(PoseOptStats):(i + 1) * C.sizeof(PoseOptStats)].tobytes())
                 for i in range(self.frame.batch)]
        return T, stats

    def processMatchedPoints(self, matcher, n_new_records, max_reproj_error=2.0, download=True):
        """StereoFrontend::processMatchedPoints (stereo_frontend.cpp:834-974), data-parallel part, at the pose this
        optimiser holds (i.e. right behind calcFastMotionOnly, no host round trip): per-record gate flags and
        pyramid-level positions + PointStatistics.  Returns (GATED_POINT_DTYPE[batch, n], POINT_STATS_DTYPE[batch])."""
        fr = self.frame
        n = matcher._n
        dev = fr.pyr[0].device
        with torch.cuda.stream(fr.stream):
            self.d_gated = torch.zeros(fr.batch * max(n, 1) * GATED_POINT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            self.d_pstats = torch.zeros(fr.batch * POINT_STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self.ctx.call("svs_process_matched_points", matcher._keep[4].data_ptr(), matcher._keep[1].data_ptr(), n, n, n,
                      int(n_new_records), C.byref(fr.cams[0]), self.d_T.data_ptr(), float(max_reproj_error),
                      self.d_gated.data_ptr(), n, self.d_pstats.data_ptr(), fr.batch)
        if not download:
            return None
        self.ctx.sync()
        gated = self.d_gated.cpu().numpy().view(GATED_POINT_DTYPE)[:fr.batch * n].reshape(fr.batch, n)
        return gated, self.d_pstats.cpu().numpy().view(POINT_STATS_DTYPE)


class DenseTracker:
    """DenseTracker, CPU-path semantics (dense_tracking.cpp:222-423)."""

    def __init__(self, ctx, frame):
        self.ctx, self.frame = ctx, frame
        dev = frame.pyr[0].device
        with torch.cuda.stream(frame.stream):
            self.ref_dense_points = [torch.zeros((frame.batch, frame.h[l] // 4, frame.w[l] // 4, 4),
                                                 dtype=torch.float32, device=dev) for l in range(NUM_PYR_LEVELS)]
            self.d_T = torch.zeros((frame.batch, 12), dtype=torch.float64, device=dev)
            self.d_passes = torch.zeros(frame.batch, dtype=torch.int32, device=dev)
            self.d_sums = torch.zeros(frame.batch * DENSE_SUMS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            self.d_T_jac = torch.zeros((frame.batch, NUM_PYR_LEVELS, 12), dtype=torch.float64, device=dev)
            self.d_rec = torch.zeros(frame.batch * 128 * DENSE_LM_RECORD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            self.d_nrec = torch.zeros(frame.batch, dtype=torch.int32, device=dev)
            # DenseTracker's constructor: residual_img[level].setTo((0,0,0,1)) (dense_tracking.cpp:52-54)
            self.residual_img = [torch.zeros_like(c) for c in self.ref_dense_points]
            for r in self.residual_img:
                r[..., 3] = 1.0

    def _set_T(self, T):
        T = np.broadcast_to(np.asarray(T, np.float64).reshape(-1, 12), (self.frame.batch, 12))
        with torch.cuda.stream(self.frame.stream):
            self.d_T.copy_(torch.as_tensor(np.array(T, dtype=np.float64)))

    def computeDensePointCloudCpu(self, T_cur_from_actkey):
        fr = self.frame
        self._set_T(T_cur_from_actkey)
        for l in range(NUM_PYR_LEVELS):
            cb = (fr.h[l] // 4) * (fr.w[l] // 4) * 4
            self.ctx.call("svs_pointcloud_cpu_sem", fr.disp.data_ptr(), fr.stride[0], fr.bstride(0),
                          C.byref(fr.cams[l]), l, self.d_T.data_ptr(), self.ref_dense_points[l].data_ptr(), cb, fr.batch)

    def pass_sums(self, level, prev_pyr, T, do_jac):
        """One chi2 (do_jac=False) or H,b (True) pass of the loop body; returns DENSE_SUMS_DTYPE[batch]."""
        fr = self.frame
        self._set_T(T)
        cb = (fr.h[level] // 4) * (fr.w[level] // 4) * 4
        self.ctx.call("svs_dense_pass_cpu_sem", self.ref_dense_points[level].data_ptr(), cb,
                      prev_pyr[level].data_ptr(), fr.stride[level], fr.bstride(level), fr.f32[level].data_ptr(),
                      fr.dx[level].data_ptr(), fr.dy[level].data_ptr(), fr.stride[level], fr.bstride(level),
                      C.byref(fr.cams[level]), self.d_T.data_ptr(), int(do_jac), self.d_sums.data_ptr(), fr.batch)
        self.ctx.sync()
        return self.d_sums.cpu().numpy().view(DENSE_SUMS_DTYPE).copy()

    def track_args(self, prev_pyr, from_u8=False):
        """from_u8=True: track straight from the current u8 pyramid (fused convert+Sobel taps)."""
        fr = self.frame
        a = capi.DenseTrackArgs()
        for l in range(NUM_PYR_LEVELS):
            a.d_cloud[l] = self.ref_dense_points[l].data_ptr()
            a.cloud_bstride[l] = (fr.h[l] // 4) * (fr.w[l] // 4) * 4
            a.d_prev_u8[l], a.pstride[l], a.p_bstride[l] = prev_pyr[l].data_ptr(), fr.stride[l], fr.bstride(l)
            if from_u8:
                a.d_cur_u8[l], a.c8stride[l], a.c8_bstride[l] = fr.pyr[l].data_ptr(), fr.stride[l], fr.bstride(l)
            else:
                a.d_cur[l], a.d_dx[l], a.d_dy[l] = fr.f32[l].data_ptr(), fr.dx[l].data_ptr(), fr.dy[l].data_ptr()
            a.fstride[l], a.f_bstride[l] = fr.stride[l], fr.bstride(l)
            a.cam_vec[l] = fr.cams[l]
        a.d_T_jac_out = self.d_T_jac.data_ptr()
        a.d_record_out, a.record_cap, a.d_n_record_out = self.d_rec.data_ptr(), 128, self.d_nrec.data_ptr()
        return a

    def lm_records(self):
        """accept / reject record of the last denseTrackingCpu call: list (per stream) of DENSE_LM_RECORD_DTYPE arrays"""
        self.ctx.sync()
        n = self.d_nrec.cpu().numpy()
        rec = self.d_rec.cpu().numpy().view(DENSE_LM_RECORD_DTYPE).reshape(self.frame.batch, 128)
        return [rec[b, :min(int(n[b]), 128)].copy() for b in range(self.frame.batch)]

    def computeResidualImages(self, prev_pyr, from_u8=False):
        """Fill the public residual_img[level] member as denseTrackingCpu leaves it (dense_tracking.cpp:279-329):
        the image of the last H,b pass of each level, whose pose the tracker recorded in d_T_jac."""
        fr = self.frame
        for l in range(NUM_PYR_LEVELS):
            cb = (fr.h[l] // 4) * (fr.w[l] // 4) * 4
            self.ctx.call("svs_dense_residual_image_cpu_sem", self.ref_dense_points[l].data_ptr(), cb,
                          prev_pyr[l].data_ptr(), fr.stride[l], fr.bstride(l),
                          None if from_u8 else fr.f32[l].data_ptr(), fr.stride[l], fr.bstride(l),
                          fr.pyr[l].data_ptr() if from_u8 else None, fr.stride[l], fr.bstride(l),
                          C.byref(fr.cams[l]), self.d_T_jac.data_ptr() + 96 * l, 36,
                          self.residual_img[l].data_ptr(), cb, fr.batch)
        self.ctx.sync()
        return [r.cpu().numpy() for r in self.residual_img]

    def denseTrackingCpu(self, prev_pyr, T_cur_from_actkey, args=None, download=True, from_u8=False):
        """DenseTracker::denseTrackingCpu(SE3*): in/out pose; whole LM loop in one launch."""
        fr = self.frame
        if T_cur_from_actkey is not None:
            self._set_T(T_cur_from_actkey)
        a = args or self.track_args(prev_pyr, from_u8=from_u8)
        self.ctx.check(self.ctx.lib.svs_dense_track_cpu_sem(self.ctx.h, C.byref(a), self.d_T.data_ptr(),
                                                            self.d_passes.data_ptr(), fr.batch))
        if not download:
            return None
        self.ctx.sync()
        return self.d_T.cpu().numpy().reshape(fr.batch, 3, 4), self.d_passes.cpu().numpy()


class GpuTracker:
    """GpuTracker / computePointCloud (gpu/dense_tracking.cuh:281-342): full-resolution f32 passes."""

    def __init__(self, ctx, stream, w, h):
        self.ctx, self.stream, self.w, self.h = ctx, stream, w, h
        self.d_sums = torch.zeros(DENSE_SUMS_DTYPE.itemsize, dtype=torch.uint8, device=torch.device("cuda", ctx.device))
        self._tex = None

    def bindTexture(self, I_cur, dx, dy, w, h, stride):
        self._tex = (I_cur, dx, dy, w, h, stride)

    def _pass(self, I_prev, cloud, T34_colmajor, f, cx, cy, w, h, stride_f, stride_f4, do_jac):
        I_cur, dx, dy, _, _, _ = self._tex
        T = np.ascontiguousarray(T34_colmajor, np.float32).reshape(12)
        self.ctx.call("svs_dense_pass_full", cloud.data_ptr(), w, h, stride_f4, I_prev.data_ptr(), I_cur.data_ptr(),
                      dx.data_ptr(), dy.data_ptr(), stride_f, float(f), float(cx), float(cy), T.ctypes.data,
                      int(do_jac), self.d_sums.data_ptr())
        self.ctx.sync()
        return self.d_sums.cpu().numpy().view(DENSE_SUMS_DTYPE)[0].copy()

    def jacobianReduction(self, I_prev, cloud, T, f, cx, cy, w, h, stride_f, stride_f4):
        return self._pass(I_prev, cloud, T, f, cx, cy, w, h, stride_f, stride_f4, True)

    def chi2(self, I_prev, cloud, T, f, cx, cy, w, h, stride_f, stride_f4):
        return float(self._pass(I_prev, cloud, T, f, cx, cy, w, h, stride_f, stride_f4, False)["chi2"])

    def residualImage(self, I_prev, cloud, T34_colmajor, f, cx, cy, w, h, stride_f, stride_f4, res_img):
        I_cur = self._tex[0]
        T = np.ascontiguousarray(T34_colmajor, np.float32).reshape(12)
        self.ctx.call("svs_dense_residual_image_full", cloud.data_ptr(), w, h, stride_f4, I_prev.data_ptr(), I_cur.data_ptr(),
                      stride_f, float(f), float(cx), float(cy), T.ctypes.data, res_img.data_ptr())
        self.ctx.sync()

    def computePointCloud(self, TQ_colmajor, disp, w, h, stride_in, stride_out, factor, cloud):
        TQ = np.ascontiguousarray(TQ_colmajor, np.float32).reshape(16)
        self.ctx.call("svs_pointcloud_full", TQ.ctypes.data, disp.data_ptr(), w, h, stride_in, stride_out, factor,
                      cloud.data_ptr())


class GpuFrameData:
    """The CUDA build's per-frame device data (FrameData members gpu_pyr_float32 / gpu_pyr_float32_dx / _dy / gpu_disp_32f,
    frame_grabber.hpp:139-150) for `batch` independent streams."""

    def __init__(self, ctx, stream, cam, batch=1):
        self.ctx, self.stream, self.cam, self.batch = ctx, stream, cam, batch
        self.cams = level_cams(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"], NUM_PYR_LEVELS)
        dev = torch.device("cuda", ctx.device)
        self.w = [self.cams[l].w for l in range(NUM_PYR_LEVELS)]
        self.h = [self.cams[l].h for l in range(NUM_PYR_LEVELS)]
        self.stride = [_round_up(w, 64) for w in self.w]
        with torch.cuda.stream(stream):
            self.uint8 = torch.zeros((batch, self.h[0], self.stride[0]), dtype=torch.uint8, device=dev)
            self.disp = torch.zeros((batch, self.h[0], self.stride[0]), dtype=torch.float32, device=dev)
            self.f32 = [torch.zeros((batch, self.h[l], self.stride[l]), dtype=torch.float32, device=dev) for l in range(NUM_PYR_LEVELS)]
            self.dx = [torch.zeros_like(t) for t in self.f32]
            self.dy = [torch.zeros_like(t) for t in self.f32]

    def bstride(self, l):
        return self.h[l] * self.stride[l]

    def upload(self, images, disp=None):
        with torch.cuda.stream(self.stream):
            img = torch.as_tensor(np.ascontiguousarray(images)).to(self.uint8.device)
            self.uint8[:, :, :self.w[0]] = img.reshape(self.batch, self.h[0], self.w[0])
            if disp is not None:
                d = torch.as_tensor(np.ascontiguousarray(disp, dtype=np.float32)).to(self.disp.device)
                self.disp[:, :, :self.w[0]] = d.reshape(self.batch, self.h[0], self.w[0])

    def preprocessing(self):
        """gpu convertTo + gpu::pyrDown on f32 + the REPLICATE derivative filters (frame_grabber.cpp:291-313)."""
        n = NUM_PYR_LEVELS
        P = C.c_void_p * n
        self.ctx.call("svs_preprocess_gpu_sem", self.uint8.data_ptr(), self.w[0], self.h[0], self.stride[0], self.bstride(0),
                      P(*[t.data_ptr() for t in self.f32]), P(*[t.data_ptr() for t in self.dx]), P(*[t.data_ptr() for t in self.dy]),
                      (C.c_int32 * n)(*self.stride), (C.c_size_t * n)(*[self.bstride(l) for l in range(n)]), n, self.batch)


class DenseTrackerGpu:
    """DenseTracker of the CUDA build (dense_tracking.cpp:25-215): full-resolution clouds dev_ref_dense_points_[l], damped LM."""
    RECORD_CAP = 128

    def __init__(self, ctx, frame):
        self.ctx, self.frame = ctx, frame
        dev = frame.f32[0].device
        B = frame.batch
        with torch.cuda.stream(frame.stream):
            self.dev_ref_dense_points = [torch.zeros((B, frame.h[l], frame.w[l], 4), dtype=torch.float32, device=dev) for l in range(NUM_PYR_LEVELS)]
            self.dev_residual_img = [torch.zeros_like(c) for c in self.dev_ref_dense_points]
            for r in self.dev_residual_img:
                r[..., 3] = 1.0                                   # setTo(Scalar(0,0,0,1)) (dense_tracking.cpp:40)
            self.d_T = torch.zeros((B, 12), dtype=torch.float64, device=dev)
            self.d_passes = torch.zeros(B, dtype=torch.int32, device=dev)
            self.d_T_jac = torch.zeros((B, NUM_PYR_LEVELS, 12), dtype=torch.float64, device=dev)
            self.d_rec = torch.zeros(B * self.RECORD_CAP * DENSE_LM_RECORD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            self.d_nrec = torch.zeros(B, dtype=torch.int32, device=dev)

    def _set_T(self, T):
        T = np.broadcast_to(np.asarray(T, np.float64).reshape(-1, 12), (self.frame.batch, 12))
        with torch.cuda.stream(self.frame.stream):
            self.d_T.copy_(torch.as_tensor(np.array(T, dtype=np.float64)))

    def computeDensePointCloudGpu(self, T_cur_from_actkey):
        """computePointCloud per level with TQ = T^-1 * cam_vec[level].Q() and factor 2^level (dense_tracking.cpp:195-215);
        slot 0's pose for every slot (one launch per level and slot: the reference's call surface is per frame)."""
        fr = self.frame
        T = np.asarray(T_cur_from_actkey, np.float64).reshape(3, 4)
        Ti = np.vstack([_pose_inv(T).reshape(3, 4), [0, 0, 0, 1]])
        for l in range(NUM_PYR_LEVELS):
            c = fr.cams[l]
            Q = np.array([[1, 0, 0, -c.cx], [0, 1, 0, -c.cy], [0, 0, 0, c.f], [0, 0, 1.0 / c.b, 0]])
            TQ = np.ascontiguousarray((Ti @ Q).T.reshape(16), np.float32)
            for b in range(fr.batch):
                self.ctx.call("svs_pointcloud_full", TQ.ctypes.data, fr.disp[b].data_ptr(), fr.w[l], fr.h[l], fr.stride[0], fr.w[l],
                              1 << l, self.dev_ref_dense_points[l][b].data_ptr())

    def track_args(self, prev, fuse_gradients=False, record=True):
        fr = self.frame
        a = capi.DenseTrackFullArgs()
        for l in range(NUM_PYR_LEVELS):
            a.d_cloud4[l], a.stride_f4[l], a.cloud_bstride[l] = self.dev_ref_dense_points[l].data_ptr(), fr.w[l], fr.w[l] * fr.h[l]
            a.d_prev[l], a.d_cur[l] = prev.f32[l].data_ptr(), fr.f32[l].data_ptr()
            if not fuse_gradients:
                a.d_dx[l], a.d_dy[l] = fr.dx[l].data_ptr(), fr.dy[l].data_ptr()
            a.stride_f[l], a.f_bstride[l] = fr.stride[l], fr.bstride(l)
            a.w[l], a.h[l] = fr.w[l], fr.h[l]
            a.f[l], a.cx[l], a.cy[l] = fr.cams[l].f, fr.cams[l].cx, fr.cams[l].cy
        a.d_T_jac_out = self.d_T_jac.data_ptr()
        if record:
            a.d_record_out, a.record_cap, a.d_n_record_out = self.d_rec.data_ptr(), self.RECORD_CAP, self.d_nrec.data_ptr()
        return a

    def denseTrackingGpu(self, prev, T_cur_from_actkey, args=None, download=True, fuse_gradients=False):
        """DenseTracker::denseTrackingGpu(SE3*): in/out pose, whole damped LM in one launch.  prev: the previous GpuFrameData.
        Returns (T [batch,3,4], fused sweeps [batch], records: list of DENSE_LM_RECORD_DTYPE arrays)."""
        fr = self.frame
        if T_cur_from_actkey is not None:
            self._set_T(T_cur_from_actkey)
        a = args or self.track_args(prev, fuse_gradients)
        self.ctx.check(self.ctx.lib.svs_dense_track_full(self.ctx.h, C.byref(a), self.d_T.data_ptr(), self.d_passes.data_ptr(), fr.batch))
        if not download:
            return None
        self.ctx.sync()
        n = self.d_nrec.cpu().numpy()
        rec = self.d_rec.cpu().numpy().view(DENSE_LM_RECORD_DTYPE).reshape(fr.batch, self.RECORD_CAP)
        return (self.d_T.cpu().numpy().reshape(fr.batch, 3, 4), self.d_passes.cpu().numpy(),
                [rec[b, :min(int(n[b]), self.RECORD_CAP)].copy() for b in range(fr.batch)])

    def residualImages(self, prev):
        """dev_residual_img[l] as denseTrackingGpu leaves it: residualImage at the pose of the level's last jacobianReduction
        (dense_tracking.cpp:177-186)."""
        fr = self.frame
        self.ctx.sync()
        Tj = self.d_T_jac.cpu().numpy().reshape(fr.batch, NUM_PYR_LEVELS, 3, 4)
        for l in range(NUM_PYR_LEVELS):
            for b in range(fr.batch):
                T34 = np.ascontiguousarray(Tj[b, l].T.reshape(12), np.float32)          # GpuMatrix34: column-major
                self.ctx.call("svs_dense_residual_image_full", self.dev_ref_dense_points[l][b].data_ptr(), fr.w[l], fr.h[l], fr.w[l],
                              prev.f32[l][b].data_ptr(), fr.f32[l][b].data_ptr(), fr.stride[l], float(fr.cams[l].f), float(fr.cams[l].cx),
                              float(fr.cams[l].cy), T34.ctypes.data, self.dev_residual_img[l][b].data_ptr())
        self.ctx.sync()
        return [r.cpu().numpy() for r in self.dev_residual_img]


class StereoFrontend:
    """StereoFrontend::processFrame / processFirstFrame (stereo_frontend.h:88-95): one library call per frame (svs_frontend_*), no host round
    trip between the stages.  n_streams == 1: host arrays in and out, the way stereo_slam calls it.  n_streams > 1: that many independent front
    ends whose stages run as one launch each, frames in device memory (torch tensors), results fetched per stream."""

    def __init__(self, ctx, cam, max_points=4096, max_keyframes=8, params=None, n_streams=1):
        self.ctx, self.cam, self.n_streams = ctx, cam, n_streams
        self.params = params or capi.FrontendParams.reference()
        camc = Cam(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
        self.h = C.c_void_p()
        ctx.check(ctx.lib.svs_frontend_create_batch(ctx.h, C.byref(camc), C.byref(self.params), max_points, max_keyframes, n_streams, C.byref(self.h)))
        ctx.children.add(self)
        self.n_points = [0] * n_streams

    @staticmethod
    def _img(a, dtype):
        return None if a is None else np.ascontiguousarray(a, dtype)

    @staticmethod
    def _ptr(a):
        return a.ctypes.data if a is not None else None

    def processFirstFrame(self, left=None, right=None, disp=None):
        left, right, disp = self._img(left, np.uint8), self._img(right, np.uint8), self._img(disp, np.float32)
        w = self.cam["w"]
        self.ctx.check(self.ctx.lib.svs_frontend_first_frame(self.h, self._ptr(left), w, self._ptr(right), w, self._ptr(disp), w))

    def keepKeyframe(self, slot, T_kf_from_w, stream=0):
        T = np.ascontiguousarray(T_kf_from_w, np.float64).reshape(12)
        self.ctx.check(self.ctx.lib.svs_frontend_keep_keyframe_of(self.h, stream, slot, T.ctypes.data))

    def keepKeyframes(self, slot, T_kf_from_w):
        """all streams at once: the frame each stream processed last -> its keyframe slot `slot`; poses [n_streams][12]"""
        T = np.ascontiguousarray(T_kf_from_w, np.float64).reshape(self.n_streams, 12)
        self.ctx.check(self.ctx.lib.svs_frontend_keep_keyframes(self.h, slot, T.ctypes.data))

    def setCandidateListsAll(self, pts_per_stream, group_end_per_stream):
        """matchAndTrack's lists of ALL streams in one staged upload (every stream with the same number of groups)"""
        ge = np.ascontiguousarray(group_end_per_stream, np.int32).reshape(self.n_streams, -1)
        n = np.array([len(p) for p in pts_per_stream], np.int32)
        pts = np.ascontiguousarray(np.concatenate([np.asarray(p, CANDIDATE_DTYPE) for p in pts_per_stream]), CANDIDATE_DTYPE)
        self.ctx.check(self.ctx.lib.svs_frontend_set_candidates_all(self.h, pts.ctypes.data, n.ctypes.data, ge.ctypes.data, ge.shape[1]))
        for b in range(self.n_streams):
            self.n_points[b] = int(n[b])

    def setCandidates(self, pts, n_new_records, stream=0):
        self.setCandidateLists(pts, [int(n_new_records), len(pts)], stream)

    def setCandidateLists(self, pts, group_end, stream=0):
        """matchAndTrack's lists in the order it walks them: group_end[0] = end of newpoint_map[actkey], then one end per neighbour (strength order),
        last = len(pts) = end of the neighbourhood's point_list"""
        pts = np.ascontiguousarray(pts, CANDIDATE_DTYPE)
        ge = np.ascontiguousarray(group_end, np.int32)
        self.ctx.check(self.ctx.lib.svs_frontend_set_candidates_grouped(self.h, stream, pts.ctypes.data, len(pts), ge.ctypes.data, len(ge)))
        self.n_points[stream] = len(pts)

    def stagingView(self):
        """numpy views (left u8 [h, w], right u8 [h, w], disp f32 [h, w]) of the pinned buffers the next frame is staged in: a frame written into
        them and passed on as it is needs no host-side copy"""
        p = [C.c_void_p() for _ in range(3)]
        self.ctx.check(self.ctx.lib.svs_frontend_staging_view(self.h, C.byref(p[0]), C.byref(p[1]), C.byref(p[2])))
        h, w = self.cam["h"], self.cam["w"]
        mk = lambda ptr, ct, dt: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(h, w)).view(dt)
        return mk(p[0], C.c_uint8, np.uint8), mk(p[1], C.c_uint8, np.uint8), mk(p[2], C.c_float, np.float32)

    def prefetchFrame(self, left, right=None, disp=None):
        left, right, disp = self._img(left, np.uint8), self._img(right, np.uint8), self._img(disp, np.float32)
        w = self.cam["w"]
        self._keep = (left, right, disp)
        self.ctx.check(self.ctx.lib.svs_frontend_prefetch_frame(self.h, self._ptr(left), w, self._ptr(right), w, self._ptr(disp), w))

    def submitFrame(self, left, T_cur_from_actkey, T_actkey_from_w, right=None, disp=None):
        left, right, disp = self._img(left, np.uint8), self._img(right, np.uint8), self._img(disp, np.float32)
        w = self.cam["w"]
        Tc = np.ascontiguousarray(T_cur_from_actkey, np.float64).reshape(12)
        Ta = np.ascontiguousarray(T_actkey_from_w, np.float64).reshape(12)
        self.ctx.check(self.ctx.lib.svs_frontend_submit_frame(self.h, self._ptr(left), w, self._ptr(right), w, self._ptr(disp), w, Tc.ctypes.data, Ta.ctypes.data, 1, 1))

    def waitFrame(self):
        res = capi.FrameResult()
        m = np.zeros(self.n_points[0], MATCH_RESULT_DTYPE)
        g = np.zeros(self.n_points[0], GATED_POINT_DTYPE)
        self.ctx.check(self.ctx.lib.svs_frontend_wait_frame(self.h, C.byref(res), m.ctypes.data, g.ctypes.data))
        return res, m, g

    def processFrame(self, left, T_cur_from_actkey, T_actkey_from_w, right=None, disp=None):
        """returns (FrameResult, MATCH_RESULT_DTYPE[n], GATED_POINT_DTYPE[n]); left=None: the frame was prefetched"""
        left, right, disp = self._img(left, np.uint8), self._img(right, np.uint8), self._img(disp, np.float32)
        w = self.cam["w"]
        Tc = np.ascontiguousarray(T_cur_from_actkey, np.float64).reshape(12)
        Ta = np.ascontiguousarray(T_actkey_from_w, np.float64).reshape(12)
        res = capi.FrameResult()
        m = np.zeros(self.n_points[0], MATCH_RESULT_DTYPE)
        g = np.zeros(self.n_points[0], GATED_POINT_DTYPE)
        self.ctx.check(self.ctx.lib.svs_frontend_process_frame(self.h, self._ptr(left), w, self._ptr(right), w, self._ptr(disp), w, Tc.ctypes.data, Ta.ctypes.data,
                                                               C.byref(res), m.ctypes.data, g.ctypes.data))
        return res, m, g

    # ---- all streams, frames in device memory
    def _frames(self, left, right, disp, ready_event=None):
        """torch tensors [n_streams][h][stride] (u8, u8, f32) -> svs_frames_dev (None = written in place through inputView).  ready_event: a torch.cuda.Event recorded
        behind whatever produced the frames (None: they are produced by work on the context's stream enqueued before the call)"""
        if left is None:
            return None
        fr = capi.FramesDev()
        if ready_event is not None:
            fr.ready_event = ready_event.cuda_event
        fr.d_left, fr.lstride, fr.l_bstride = left.data_ptr(), left.stride(1), left.stride(0)
        if right is not None:
            fr.d_right, fr.rstride, fr.r_bstride = right.data_ptr(), right.stride(1), right.stride(0)
        if disp is not None:
            fr.d_disp, fr.dstride, fr.d_bstride = disp.data_ptr(), disp.stride(1), disp.stride(0)
        return fr

    def inputView(self):
        """(left, right, disp): device pointer, row stride, stream stride of the buffers the next frames may be written into"""
        p = [C.c_void_p() for _ in range(3)]
        st = [C.c_int32() for _ in range(3)]
        bs = [C.c_size_t() for _ in range(3)]
        self.ctx.check(self.ctx.lib.svs_frontend_input_view(self.h, C.byref(p[0]), C.byref(st[0]), C.byref(bs[0]), C.byref(p[1]), C.byref(st[1]), C.byref(bs[1]),
                                                            C.byref(p[2]), C.byref(st[2]), C.byref(bs[2])))
        return [(p[i].value, st[i].value, bs[i].value) for i in range(3)]

    def processFirstFrames(self, left=None, right=None, disp=None):
        fr = self._frames(left, right, disp)
        self.ctx.check(self.ctx.lib.svs_frontend_first_frames(self.h, C.byref(fr) if fr is not None else None))

    def processFrames(self, T_cur_from_actkey, T_actkey_from_w, left=None, right=None, disp=None, ready_event=None):
        """asynchronous; poses [n_streams][12]"""
        fr = self._frames(left, right, disp, ready_event)
        Tc = np.ascontiguousarray(T_cur_from_actkey, np.float64).reshape(self.n_streams, 12)
        Ta = np.ascontiguousarray(T_actkey_from_w, np.float64).reshape(self.n_streams, 12)
        self.ctx.check(self.ctx.lib.svs_frontend_process_frames(self.h, C.byref(fr) if fr is not None else None, Tc.ctypes.data, Ta.ctypes.data))

    def results(self, stream=0):
        res = capi.FrameResult()
        m = np.zeros(self.n_points[stream], MATCH_RESULT_DTYPE)
        g = np.zeros(self.n_points[stream], GATED_POINT_DTYPE)
        self.ctx.check(self.ctx.lib.svs_frontend_results(self.h, stream, C.byref(res), m.ctypes.data, g.ctypes.data))
        return res, m, g

    def poses(self):
        T = np.zeros((self.n_streams, 12))
        ok = np.zeros(self.n_streams, np.int32)
        self.ctx.check(self.ctx.lib.svs_frontend_poses(self.h, T.ctypes.data, ok.ctypes.data))
        return T.reshape(-1, 3, 4), ok

    STAGES = ("preprocess", "dense_tracking", "stereo", "fast", "match", "pose_refinement", "process_points", "pointcloud")

    def setTiming(self, on):
        self.ctx.check(self.ctx.lib.svs_frontend_set_timing(self.h, int(on)))

    def stageTimes(self):
        """ms per stage of the last timed call (the reference's per_mon_ stages, stereo_frontend.cpp:190-302)"""
        ms = np.zeros(len(self.STAGES), np.float32)
        self.ctx.check(self.ctx.lib.svs_frontend_stage_times(self.h, ms.ctypes.data))
        return dict(zip(self.STAGES, [float(v) for v in ms]))

    def denseRecords(self, stream=0):
        rec = np.zeros(64, DENSE_LM_RECORD_DTYPE)
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.svs_frontend_dense_records(self.h, stream, rec.ctypes.data, 64, C.byref(n)))
        return rec[:min(n.value, 64)].copy()

    def recomputeCloud(self, T_cur_from_actkey):
        """one pose for every stream ([12]) or one per stream ([n_streams][12])"""
        T = np.ascontiguousarray(T_cur_from_actkey, np.float64).reshape(-1, 12)
        T = np.ascontiguousarray(np.broadcast_to(T, (self.n_streams, 12)))
        self.ctx.check(self.ctx.lib.svs_frontend_recompute_cloud(self.h, T.ctypes.data))

    def cloud_host(self, level, stream=0):
        """reference cloud (quarter grid; full resolution in the CUDA build) the last frame left for the next one"""
        clouds = (C.c_void_p * 3)()
        self.ctx.check(self.ctx.lib.svs_frontend_device_view(self.h, stream, None, None, None, clouds, None))
        q = 1 if self.params.cuda_build else 4
        w, h = (self.cam["w"] >> level) // q, (self.cam["h"] >> level) // q
        out = np.zeros((h, w, 4), np.float32)
        self.ctx.call("svs_memcpy_d2h", out.ctypes.data, clouds[level], out.nbytes)
        return out

    def fast_handle(self):
        f = C.c_void_p()
        self.ctx.check(self.ctx.lib.svs_frontend_device_view(self.h, 0, None, None, None, None, C.byref(f)))
        return f

    def corners(self, stream=0, level=0, cap=8192):
        """corner list of the last frame (quadtree insertion order), per-cell counts, emit thresholds, persistent thresholds"""
        f = self.fast_handle()
        xy = np.zeros((cap, 2), np.int16)
        n = C.c_int32()
        cc, et, ts = np.zeros(64, np.int32), np.zeros(64, np.int32), np.zeros(64, np.int32)
        self.ctx.check(self.ctx.lib.svs_fast_download(f, stream, level, xy.ctypes.data, cap, C.byref(n), cc.ctypes.data, et.ctypes.data, ts.ctypes.data))
        return xy[:n.value].copy(), cc, et, ts

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.lib.svs_frontend_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StereoMatcher:
    """StereoFrontend::calcDisparityCpu (stereo_frontend.cpp:620-653): cv::StereoBM on the level-0 left image
    and the right image, float disparity written into the frame's `disp` (-1 where filtered)."""

    def __init__(self, ctx, frame, params=None):
        self.ctx, self.frame = ctx, frame
        self.params = params or StereoParams.reference()
        self.h = C.c_void_p()
        ctx.call("svs_stereo_create", frame.w[0], frame.h[0], frame.batch, C.byref(self.params), C.byref(self.h))
        ctx.children.add(self)
        with torch.cuda.stream(frame.stream):
            self.right = torch.zeros_like(frame.pyr[0])

    def upload_right(self, images):
        with torch.cuda.stream(self.frame.stream):
            img = torch.as_tensor(np.ascontiguousarray(images)).to(self.right.device)
            self.right[:, :, :self.frame.w[0]] = img.reshape(self.frame.batch, self.frame.h[0], self.frame.w[0])

    def calcDisparityCpu(self, n_batch=None):
        f = self.frame
        self.ctx.check(self.ctx.lib.svs_stereo_compute(self.h, f.pyr[0].data_ptr(), f.stride[0], f.bstride(0),
                                                       self.right.data_ptr(), f.stride[0], f.bstride(0), f.disp.data_ptr(),
                                                       f.stride[0], f.bstride(0), n_batch or f.batch))

    def disparity_host(self, slot=0):
        self.ctx.sync()
        return self.frame.disp[slot, :, :self.frame.w[0]].cpu().numpy()

    def close(self):
        if self.h:
            self.ctx.lib.svs_stereo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
