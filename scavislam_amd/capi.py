"""ctypes binding of libscavislam_hip.so (include/scavislam_hip.h).

The library is the product: there is NO CPU fallback.  Importing this module never touches the
GPU; `load()` raises if the HIP extension has not been built, and every call raises SvsError on a
non-zero status.  PyTorch is used only as plumbing (device buffers, streams, torch.distributed).
"""
import ctypes as C
import os
import weakref

from .ctypes_types import BaParams, BaStats, Cam, FastGrid, PoseOptParams, PoseOptStats, StereoParams

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVS_LIB_PATH") or os.path.join(_HERE, "libscavislam_hip.so")   # override = kernel A/B experiments only
_LIB = None

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)


class SvsError(RuntimeError):
    pass


class DenseTrackArgs(C.Structure):
    _fields_ = [("d_cloud", C.c_void_p * 3), ("cloud_bstride", C.c_size_t * 3),
                ("d_prev_u8", C.c_void_p * 3), ("pstride", C.c_int32 * 3), ("p_bstride", C.c_size_t * 3),
                ("d_cur", C.c_void_p * 3), ("d_dx", C.c_void_p * 3), ("d_dy", C.c_void_p * 3),
                ("fstride", C.c_int32 * 3), ("f_bstride", C.c_size_t * 3), ("cam_vec", Cam * 3),
                ("d_cur_u8", C.c_void_p * 3), ("c8stride", C.c_int32 * 3), ("c8_bstride", C.c_size_t * 3),
                ("d_T_jac_out", C.c_void_p), ("d_record_out", C.c_void_p), ("record_cap", C.c_int32), ("d_n_record_out", C.c_void_p)]


class DenseTrackFullArgs(C.Structure):
    _fields_ = [("d_cloud4", C.c_void_p * 3), ("stride_f4", C.c_int32 * 3), ("cloud_bstride", C.c_size_t * 3),
                ("d_prev", C.c_void_p * 3), ("d_cur", C.c_void_p * 3), ("d_dx", C.c_void_p * 3), ("d_dy", C.c_void_p * 3),
                ("stride_f", C.c_int32 * 3), ("f_bstride", C.c_size_t * 3),
                ("w", C.c_int32 * 3), ("h", C.c_int32 * 3),
                ("f", C.c_double * 3), ("cx", C.c_double * 3), ("cy", C.c_double * 3),
                ("d_T_jac_out", C.c_void_p), ("d_record_out", C.c_void_p), ("record_cap", C.c_int32),
                ("d_n_record_out", C.c_void_p)]


class FrontendParams(C.Structure):
    _fields_ = [("fast_trials", C.c_int32), ("search_radius", C.c_int32), ("thr_mean", C.c_int32), ("thr_std", C.c_int32),
                ("max_reproj_error", C.c_float), ("use_block_matching", C.c_int32), ("pose_opt", PoseOptParams), ("stereo", StereoParams),
                ("n_levels", C.c_int32), ("num_max_points", C.c_int32), ("min_matches", C.c_int32), ("cuda_build", C.c_int32)]

    @classmethod
    def reference(cls, use_block_matching=False, cuda_build=False, n_levels=3, search_radius=None, thr_mean=22, thr_std=10, num_max_points=300):
        """the values StereoFrontend uses (stereo_frontend.cpp:232, :989-1004, :845-846, :1061, :620-653); cuda_build = its SCAVISLAM_CUDA_SUPPORT
        build (full-resolution tracker, matcher search radius 4, :1043-1047).  n_levels: use_n_levels_in_frontent (code default 2, shipped cfgs 3)"""
        if search_radius is None:
            search_radius = 4 if cuda_build else 8
        return cls(6, search_radius, thr_mean, thr_std, 2.0, int(use_block_matching), PoseOptParams.reference(), StereoParams.reference(),
                   n_levels, num_max_points, 20, int(cuda_build))


class FramesDev(C.Structure):
    """svs_frames_dev: images of all streams in device memory"""
    _fields_ = [("d_left", C.c_void_p), ("lstride", C.c_int32), ("l_bstride", C.c_size_t),
                ("d_right", C.c_void_p), ("rstride", C.c_int32), ("r_bstride", C.c_size_t),
                ("d_disp", C.c_void_p), ("dstride", C.c_int32), ("d_bstride", C.c_size_t),
                ("ready_event", C.c_void_p)]


class PointStatsC(C.Structure):
    _fields_ = [("num_points_grid2x2", C.c_int32 * 4), ("num_points_grid3x3", C.c_int32 * 9), ("num_matched_points", C.c_int32 * 3),
                ("num_track_points", C.c_int32), ("num_obs", C.c_int32), ("pad_", C.c_int32 * 2), ("sum_track_length", C.c_double)]


class FrameResult(C.Structure):
    _fields_ = [("T_cur_from_actkey", C.c_double * 12), ("dense_passes", C.c_int32), ("n_points", C.c_int32), ("n_matched", C.c_int32),
                ("tracking_ok", C.c_int32), ("pose_stats", PoseOptStats), ("point_stats", PointStatsC)]


class MatchArgs(C.Structure):
    _fields_ = [("d_kfs", C.c_void_p), ("n_kf", C.c_int32),
                ("d_pts", C.c_void_p), ("n_pts", C.c_int32),
                ("d_T_cur_from_w", C.c_void_p), ("d_T_w_from_actkey", C.c_void_p),
                ("d_cur_pyr", C.c_void_p * 3), ("cur_stride", C.c_int32 * 3), ("cur_bstride", C.c_size_t * 3),
                ("d_disp", C.c_void_p), ("disp_stride", C.c_int32), ("disp_bstride", C.c_size_t),
                ("cam_vec", Cam * 3),
                ("search_radius", C.c_int32), ("thr_mean", C.c_int32), ("thr_std", C.c_int32),
                ("n_batch", C.c_int32),
                ("kf_bstride", C.c_size_t), ("pts_bstride", C.c_size_t), ("out_bstride", C.c_size_t)]


_SIGS = {
    "svs_ctx_create": [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)],
    "svs_ctx_destroy": [C.c_void_p],
    "svs_ctx_sync": [C.c_void_p],
    "svs_ctx_set_option": [C.c_void_p, C.c_char_p, C.c_int],
    "svs_ctx_get_stat": [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)],
    "svs_malloc": [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)],
    "svs_free": [C.c_void_p, C.c_void_p],
    "svs_memcpy_h2d": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "svs_memcpy_d2h": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "svs_timer_start": [C.c_void_p],
    "svs_timer_stop_ms": [C.c_void_p, C.POINTER(C.c_float)],
    "svs_pyr_down_u8": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int,
                        C.c_size_t, C.c_int],
    "svs_convert_sobel_f32": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int],
    "svs_fast_create": [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(FastGrid),
                        C.c_int, C.c_int, C.POINTER(C.c_void_p)],
    "svs_fast_destroy": [C.c_void_p],
    "svs_fast_detect": [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_size_t),
                        C.c_int, C.c_int],
    "svs_fast_download": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32),
                          C.c_void_p, C.c_void_p, C.c_void_p],
    "svs_fast_set_thresholds": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "svs_fast_device_view": [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_size_t), C.POINTER(C.c_int32),
                             C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
    "svs_match": [C.c_void_p, C.POINTER(MatchArgs), C.c_void_p, C.c_void_p],
    "svs_pointcloud_cpu_sem": [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(Cam), C.c_int,
                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_int],
    "svs_dense_pass_cpu_sem": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(Cam),
                               C.c_void_p, C.c_int, C.c_void_p, C.c_int],
    "svs_dense_seq_sum_f32": [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "svs_dense_track_cpu_sem": [C.c_void_p, C.POINTER(DenseTrackArgs), C.c_void_p, C.c_void_p, C.c_int],
    "svs_dense_residual_image_cpu_sem": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t,
                                         C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t,
                                         C.POINTER(Cam), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int],
    "svs_dense_residual_image_full": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p],
    "svs_preprocess_gpu_sem": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                               C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_size_t), C.c_int, C.c_int],
    "svs_dense_track_full": [C.c_void_p, C.POINTER(DenseTrackFullArgs), C.c_void_p, C.c_void_p, C.c_int],
    "svs_dense_pixel_terms_full": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p],
    "svs_dense_pass_full": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p,
                            C.c_int, C.c_void_p],
    "svs_pointcloud_full": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_void_p],
    "svs_motion_only": [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(Cam), C.POINTER(PoseOptParams), C.c_void_p, C.c_void_p,
                        C.c_int],
    "svs_process_matched_points": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(Cam),
                                   C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int],
    "svs_stereo_create": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(StereoParams), C.POINTER(C.c_void_p)],
    "svs_stereo_destroy": [C.c_void_p],
    "svs_stereo_compute": [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int,
                           C.c_size_t, C.c_int],
    "svs_frontend_create": [C.c_void_p, C.POINTER(Cam), C.POINTER(FrontendParams), C.c_int, C.c_int, C.POINTER(C.c_void_p)],
    "svs_frontend_create_batch": [C.c_void_p, C.POINTER(Cam), C.POINTER(FrontendParams), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)],
    "svs_frontend_keep_keyframe_of": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "svs_frontend_set_candidates_grouped": [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int],
    "svs_frontend_keep_keyframes": [C.c_void_p, C.c_int, C.c_void_p],
    "svs_frontend_set_candidates_all": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int],
    "svs_frontend_submit_frame": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int],
    "svs_frontend_wait_frame": [C.c_void_p, C.POINTER(FrameResult), C.c_void_p, C.c_void_p],
    "svs_frontend_prefetch_frame": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int],
    "svs_frontend_staging_view": [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)],
    "svs_frontend_input_view": [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_size_t)],
    "svs_frontend_first_frames": [C.c_void_p, C.POINTER(FramesDev)],
    "svs_frontend_process_frames": [C.c_void_p, C.POINTER(FramesDev), C.c_void_p, C.c_void_p],
    "svs_frontend_results": [C.c_void_p, C.c_int, C.POINTER(FrameResult), C.c_void_p, C.c_void_p],
    "svs_frontend_poses": [C.c_void_p, C.c_void_p, C.c_void_p],
    "svs_frontend_set_timing": [C.c_void_p, C.c_int],
    "svs_frontend_stage_times": [C.c_void_p, C.c_void_p],
    "svs_frontend_dense_records": [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32)],
    "svs_pointcloud_full_pose": [C.c_void_p, C.c_void_p, C.POINTER(Cam), C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                 C.c_void_p, C.c_int],
    "svs_frontend_destroy": [C.c_void_p],
    "svs_frontend_first_frame": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int],
    "svs_frontend_keep_keyframe": [C.c_void_p, C.c_int, C.c_void_p],
    "svs_frontend_set_candidates": [C.c_void_p, C.c_void_p, C.c_int, C.c_int],
    "svs_frontend_process_frame": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.POINTER(FrameResult), C.c_void_p, C.c_void_p],
    "svs_frontend_recompute_cloud": [C.c_void_p, C.c_void_p],
    "svs_frontend_device_view": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "svs_ba_create": [C.c_void_p, C.POINTER(C.c_void_p)],
    "svs_ba_destroy": [C.c_void_p],
    "svs_ba_set_problem": [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                           C.c_int, C.c_void_p, C.POINTER(Cam), C.POINTER(BaParams), C.c_int],
    "svs_ba_window_update": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                             C.c_void_p, C.POINTER(Cam), C.POINTER(BaParams)],
    "svs_ba_window_reset": [C.c_void_p],
    "svs_ba_window_forget_keyframes": [C.c_void_p, C.c_void_p, C.c_int],
    "svs_ba_optimize": [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(BaStats)],
    "svs_ba_optimize_batch": [C.POINTER(C.c_void_p), C.c_int, C.POINTER(BaStats)],
    "svs_ba_get_state": [C.c_void_p, C.c_void_p, C.c_void_p],
    "svs_ba_reset_state": [C.c_void_p, C.c_void_p, C.c_void_p],
    "svs_ba_reduced_system": [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p],
    "svs_ba_info": [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "svs_ba_order_info": [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p],
    "svs_ba_set_timing": [C.c_void_p, C.c_int],
    "svs_ba_set_option": [C.c_void_p, C.c_char_p, C.c_int],
    "svs_ba_set_comm": [C.c_void_p, C.c_void_p],
    "svs_comm_get_unique_id": [C.c_void_p, C.c_void_p],
    "svs_comm_create": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)],
    "svs_comm_destroy": [C.c_void_p],
    "svs_comm_allreduce_f64": [C.c_void_p, C.c_void_p, C.c_size_t],
    "svs_comm_stats": [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "svs_comm_create_p2p": [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p],
    "svs_comm_connect_p2p": [C.c_void_p, C.c_void_p],
    "svs_comm_transport": [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)],
    "svs_ba_kernel_times": [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                            C.POINTER(C.c_int32)],
    "svs_ba_graph_stats": [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
}
EXPORTS = sorted(list(_SIGS) + ["svs_ctx_stream", "svs_last_error", "svs_api_version", "svs_pose_opt_params_default"])
API_VERSION = 7      # SVS_API_VERSION of include/scavislam_hip.h this binding was written against


def load():
    """dlopen the HIP extension; raises (never falls back) if it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SvsError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        # torch bundles its own libamdhip64; importing it first makes the loader bind this library to that same HIP
        # runtime instance (two runtimes in one process do not share devices, streams or allocations)
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        lib.svs_ctx_stream.argtypes = [C.c_void_p]
        lib.svs_ctx_stream.restype = C.c_void_p
        lib.svs_last_error.argtypes = [C.c_void_p]
        lib.svs_last_error.restype = C.c_char_p
        lib.svs_api_version.argtypes = []
        lib.svs_api_version.restype = C.c_int
        lib.svs_pose_opt_params_default.argtypes = [C.c_void_p]
        lib.svs_pose_opt_params_default.restype = None
        if lib.svs_api_version() != API_VERSION:
            raise SvsError(f"{LIB_PATH} was built with SVS_API_VERSION {lib.svs_api_version()}, this binding expects {API_VERSION}: rebuild (__graft_entry__.build())")
        _LIB = lib
    return _LIB


class Context:
    """One svs_ctx (HIP stream + scratch) per calling thread, as in SURVEY.md 8b."""

    def __init__(self, device=0, stream_ptr=None):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.svs_ctx_create(int(device), C.c_void_p(stream_ptr) if stream_ptr else None, C.byref(h))
        if rc:
            raise SvsError(f"svs_ctx_create failed with status {rc} (no gfx950 device visible?)")
        self.h = h
        self.device = device
        self.children = weakref.WeakSet()   # objects holding a pointer to this ctx; closed first

    def check(self, rc):
        if rc:
            raise SvsError(f"status {rc}: {self.lib.svs_last_error(self.h).decode(errors='replace')}")

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self.h, *args))

    def sync(self):
        self.check(self.lib.svs_ctx_sync(self.h))

    def set_option(self, name, value):
        """experiment / test switches of the context (see svs_ctx_set_option in the header)"""
        self.call("svs_ctx_set_option", name.encode(), int(value))

    def get_stat(self, name):
        """counters of the context (svs_ctx_get_stat): "trk_exact_sums", "trk_exact_fallbacks"; blocking"""
        v = C.c_longlong(0)
        self.call("svs_ctx_get_stat", name.encode(), C.byref(v))
        return int(v.value)

    def timer_start(self):
        self.check(self.lib.svs_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        self.check(self.lib.svs_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            for ch in list(self.children):
                ch.close()
            self.lib.svs_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def torch_context(device=0):
    """Context running on a dedicated torch stream, so torch copies/collectives and our kernels
    are ordered on the same HIP stream.  Returns (ctx, torch_stream)."""
    import torch
    torch.cuda.set_device(device)
    s = torch.cuda.Stream(device=device)
    return Context(device, s.cuda_stream), s
