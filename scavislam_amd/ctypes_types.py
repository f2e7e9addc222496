"""ctypes mirrors of the POD structs in include/scavislam_hip.h (the test oracle declares the same layouts independently).

Names follow the reference's domain: FastGridCell grids (keyframes.h:31-44), CandidatePoint
(data_structures.h:37-69), the BA edge records of SlamGraph::copyDataToG2o
(slam_graph.cpp:983-1032).
"""
import ctypes as C

import numpy as np

SVS_MAX_CELLS = 64


class Cam(C.Structure):
    """Per-level StereoCamera (frame_grabber-impl.cpp:48-60)."""
    _fields_ = [("f", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("b", C.c_double),
                ("w", C.c_int32), ("h", C.c_int32)]


class FastGrid(C.Structure):
    """FastGrid members (fast_grid.cpp:23-58)."""
    _fields_ = [("gx", C.c_int32), ("gy", C.c_int32), ("cell_w", C.c_int32), ("cell_h", C.c_int32),
                ("min_inner", C.c_int32), ("min_outer", C.c_int32),
                ("max_inner", C.c_int32), ("max_outer", C.c_int32),
                ("fast_min", C.c_int32), ("fast_max", C.c_int32),
                ("thr", C.c_int32 * SVS_MAX_CELLS)]


CANDIDATE_DTYPE = np.dtype([("xyz_anchor", "<f8", 3), ("anchor_obs_pyr", "<f8", 3),
                            ("anchor_level", "<i4"), ("kf_index", "<i4"),
                            ("point_id", "<i4"), ("pad_", "<i4")], align=True)
assert CANDIDATE_DTYPE.itemsize == 64

MATCH_RESULT_DTYPE = np.dtype([("status", "<i4"), ("u", "<i4"), ("v", "<i4"), ("znssd", "<i4"),
                               ("obs", "<f8", 3), ("xyz_actkey", "<f8", 3)], align=True)
assert MATCH_RESULT_DTYPE.itemsize == 64

KEYFRAME_DTYPE = np.dtype([("T_anchor_from_w", "<f8", 12), ("pyr", "<u8", 3),
                           ("stride", "<i4", 3), ("pad_", "<i4")], align=True)
assert KEYFRAME_DTYPE.itemsize == 136

DENSE_SUMS_DTYPE = np.dtype([("H", "<f8", 21), ("b", "<f8", 6), ("chi2", "<f8"),
                             ("n_valid", "<i8")], align=True)
assert DENSE_SUMS_DTYPE.itemsize == 232

# svs_dense_lm_record: one entry per chi2 evaluation of DenseTracker::denseTrackingGpu (dense_tracking.cpp:60-193)
DENSE_LM_RECORD_DTYPE = np.dtype([("level", "<i4"), ("accepted", "<i4"), ("chi2", "<f4"), ("new_chi2", "<f4")])
assert DENSE_LM_RECORD_DTYPE.itemsize == 16

BA_EDGE_DTYPE = np.dtype([("obs", "<f8", 3), ("info", "<f8", 3), ("point", "<i4"),
                          ("pose", "<i4"), ("anchor", "<i4"), ("pad_", "<i4")], align=True)
assert BA_EDGE_DTYPE.itemsize == 64

BA_CONSTRAINT_DTYPE = np.dtype([("T_21", "<f8", 12), ("info", "<f8", 36),
                                ("pose1", "<i4"), ("pose2", "<i4")], align=True)
assert BA_CONSTRAINT_DTYPE.itemsize == 392


class BaParams(C.Structure):
    """OptParams (slam_graph.hpp:36-50) + the g2o settings of setupG2o/optimize."""
    _fields_ = [("num_iters", C.c_int32), ("use_robust", C.c_int32), ("huber_delta", C.c_double),
                ("lambda_init", C.c_double), ("max_trials", C.c_int32),
                ("self_edge_mode", C.c_int32)]

    @classmethod
    def reference_defaults(cls):
        # backend.cpp:187 OptParams(2,true,3); delta stays 1 (slam_graph-impl.cpp:86-90);
        # lambda 50 (slam_graph.cpp:338); maxTrialsAfterFailure 5 (slam_graph.cpp:1073)
        return cls(2, 1, 1.0, 50.0, 5, 0)


class BaStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("accepted", C.c_int32),
                ("terminated", C.c_int32), ("chi2_init", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double)]


MATCH_OK, MATCH_NO_ANCHOR, MATCH_BORDER, MATCH_DEPTH, MATCH_TEXTURE, MATCH_NONE, MATCH_NO_DISP, MATCH_SKIPPED = range(8)


def level_cams(f, cx, cy, b, w, h, levels=3):
    """cam_vec of FrameGrabber<StereoCamera> (frame_grabber-impl.cpp:48-60)."""
    out = (Cam * levels)()
    for l in range(levels):
        s = float(1 << l)
        out[l] = Cam(f / s, cx / s, cy / s, b * (1 << l), int(w / s), int(h / s))
    return out


class StereoParams(C.Structure):
    """svs_stereo_params: cv::StereoBM state as set at stereo_frontend.cpp:620-653."""
    _fields_ = [("prefilter_cap", C.c_int32), ("sad_window", C.c_int32), ("min_disparity", C.c_int32),
                ("num_disparities", C.c_int32), ("texture_threshold", C.c_int32), ("uniqueness_ratio", C.c_int32),
                ("speckle_window", C.c_int32), ("speckle_range", C.c_int32), ("disp12_max_diff", C.c_int32)]

    @classmethod
    def reference(cls, num_disp16=2):
        return cls(31, 7, 0, 16 * num_disp16, 10, 15, 100, 32, 1)


class PoseOptParams(C.Structure):
    """svs_pose_opt_params: PoseOptimizerParams (pose_optimizer.h:36-58)."""
    _fields_ = [("robust_kernel", C.c_int32), ("num_iter", C.c_int32), ("kernel_param", C.c_double),
                ("initial_mu", C.c_double), ("tau", C.c_double), ("min_obs", C.c_int32), ("pad_", C.c_int32)]

    @classmethod
    def reference(cls):
        """PoseOptimizerParams(true, 2, 15) as passed at stereo_frontend.cpp:1061"""
        return cls(1, 15, 2.0, -1.0, 1e-5)


class PoseOptStats(C.Structure):
    _fields_ = [("initial_chi2", C.c_double), ("chi2", C.c_double), ("max_err", C.c_double),
                ("num_obs", C.c_int32), ("status", C.c_int32)]


# svs_gated_point / svs_point_stats: StereoFrontend::processMatchedPoints (stereo_frontend.cpp:834-974)
GATED_POINT_DTYPE = np.dtype([("accepted", "<i4"), ("is_new", "<i4"), ("uv_pyr", "<f8", 2), ("curkey_uv_pyr", "<f8", 2)])
assert GATED_POINT_DTYPE.itemsize == 40
POINT_STATS_DTYPE = np.dtype([("num_points_grid2x2", "<i4", 4), ("num_points_grid3x3", "<i4", 9), ("num_matched_points", "<i4", 3),
                              ("num_track_points", "<i4"), ("num_obs", "<i4"), ("pad_", "<i4", 2), ("sum_track_length", "<f8")])
assert POINT_STATS_DTYPE.itemsize == 88
