"""GPU parity of the motion-only pose refinement (PoseOptimizer::calcFastMotionOnly, pose_optimizer.h:134-298) behind the
guided matcher (stereo_frontend.cpp:1058-1063).  f64 sums in a different order than the sequential oracle => tolerance."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _synthetic_results(rng, cam, T_true, n, outliers=0.1, n_fail=30):
    """matcher-like result records: points in the active keyframe frame, stereo observations in the current frame."""
    from scavislam_amd.ctypes_types import MATCH_RESULT_DTYPE
    res = np.zeros(n, MATCH_RESULT_DTYPE)
    xyz = np.stack([rng.uniform(-3, 3, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2.5, 15, n)], 1)
    p = xyz @ T_true[:, :3].T + T_true[:, 3]
    u = p[:, 0] / p[:, 2] * cam["f"] + cam["cx"]
    v = p[:, 1] / p[:, 2] * cam["f"] + cam["cy"]
    ur = (p[:, 0] - cam["b"]) / p[:, 2] * cam["f"] + cam["cx"]
    obs = np.stack([u, v, ur], 1) + rng.normal(0, 0.4, (n, 3))
    bad = rng.random(n) < outliers
    obs[bad] += rng.uniform(-25, 25, (bad.sum(), 3))
    res["obs"], res["xyz_actkey"] = obs, xyz
    res["status"][rng.choice(n, n_fail, replace=False)] = rng.integers(1, 7, n_fail)      # matcher rejections are skipped
    return res


def _run(gpu_ctx, cam, res_batch, T0, prm=None):
    import torch
    from scavislam_amd.ctypes_types import Cam, PoseOptParams, PoseOptStats
    ctx, stream = gpu_ctx
    prm = prm or PoseOptParams.reference()
    B, n = res_batch.shape
    with torch.cuda.stream(stream):
        d_res = torch.as_tensor(res_batch.view(np.uint8).reshape(-1)).cuda()
        d_T = torch.as_tensor(np.tile(np.asarray(T0, np.float64).reshape(12), (B, 1))).cuda()
        d_st = torch.zeros(B * C.sizeof(PoseOptStats), dtype=torch.uint8, device="cuda")
    camc = Cam(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
    ctx.call("svs_motion_only", d_res.data_ptr(), n, n, C.byref(camc), C.byref(prm), d_T.data_ptr(), d_st.data_ptr(), B)
    ctx.sync()
    raw = d_st.cpu().numpy()
    sz = C.sizeof(PoseOptStats)
    return d_T.cpu().numpy().reshape(B, 3, 4), [PoseOptStats.from_buffer_copy(raw[i * sz:(i + 1) * sz].tobytes()) for i in range(B)], camc


@pytest.mark.parametrize("shape,n", [("fused", 400), ("fused", 2600), ("legacy", 400)])
def test_motion_only_matches_oracle(gpu_ctx, shape, n):
    """the fused kernel with lists short and longer than what its registers hold (4 observations per lane of 512: the rest is re-read through the index
    list), and the record-walking kernel of rounds 1-2"""
    import oracle as O
    from scavislam_amd import synth
    rng = np.random.default_rng(17)
    cam = synth.CAM_DEFAULT
    T_true = synth.pose(synth.so3_exp(np.array([0.01, -0.02, 0.005])), np.array([0.03, -0.01, 0.08]))
    T0 = synth.pose(np.eye(3), np.zeros(3))
    batch = np.stack([_synthetic_results(rng, cam, T_true, n) for _ in range(3)])
    ctx = gpu_ctx[0]
    ctx.set_option("mo_legacy", 1 if shape == "legacy" else 0)
    try:
        Tg, stg, camc = _run(gpu_ctx, cam, batch, T0)
    finally:
        ctx.set_option("mo_legacy", 0)
    for b in range(3):
        Tr, sr = O.motion_only(batch[b], camc, T0)
        assert stg[b].status == 0 and stg[b].num_obs == sr.num_obs == int((batch[b]["status"] == 0).sum())
        np.testing.assert_allclose(Tg[b], Tr, rtol=0, atol=1e-9)
        np.testing.assert_allclose([stg[b].initial_chi2, stg[b].chi2, stg[b].max_err], [sr.initial_chi2, sr.chi2, sr.max_err], rtol=1e-9)
        assert stg[b].chi2 < stg[b].initial_chi2
        assert np.abs(Tg[b] - T_true).max() < 5e-3      # robust kernel keeps the 10 % outliers from biasing the pose


@pytest.mark.parametrize("n", [400, 1800, 2600])
def test_motion_only_speculative_trials_leave_the_same_bits(gpu_ctx, n):
    """Context option "mo_spec": a trial that follows a rejection runs a chi2-only sweep (a third of all sweeps: every refinement ends in five rejections at the
    noise floor of its chi2); the full sweep follows only if such a trial is accepted after all.  Pose and statistics must be the bits of the always-full loop:
    24 problems with different motions, starting poses near and far (far starts make mid-run rejections that ARE followed by accepted trials)."""
    from scavislam_amd import synth
    rng = np.random.default_rng(23)
    cam = synth.CAM_DEFAULT
    ctx = gpu_ctx[0]
    n_diff_paths = 0
    for k in range(8):
        T_true = synth.pose(synth.so3_exp(rng.normal(0, 0.01, 3)), rng.normal(0, 0.04, 3))
        batch = np.stack([_synthetic_results(rng, cam, T_true, n) for _ in range(3)])
        far = k % 2 == 1
        T0 = synth.pose(np.eye(3), np.zeros(3)) if far else synth.pose_mul(synth.pose(synth.so3_exp(rng.normal(0, 0.002, 3)), rng.normal(0, 0.005, 3)), T_true)
        out = {}
        for spec in (0, 1):
            ctx.set_option("mo_spec", spec)
            try:
                out[spec] = _run(gpu_ctx, cam, batch, T0)
            finally:
                ctx.set_option("mo_spec", 1)
        T_a, st_a, _ = out[0]
        T_b, st_b, _ = out[1]
        assert T_a.tobytes() == T_b.tobytes(), k
        for a, b in zip(st_a, st_b):
            assert bytes(a) == bytes(b), k
            assert a.status == 0 and a.chi2 < a.initial_chi2


def test_motion_only_non_robust_and_fixed_mu(gpu_ctx):
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.ctypes_types import PoseOptParams
    rng = np.random.default_rng(4)
    cam = synth.CAM_NEWCOLLEGE
    T_true = synth.pose(synth.so3_exp(np.array([-0.02, 0.01, 0.0])), np.array([-0.05, 0.0, 0.1]))
    res = _synthetic_results(rng, cam, T_true, 120, outliers=0.0, n_fail=5)[None]
    for prm in (PoseOptParams(0, 15, 2.0, -1.0, 1e-5), PoseOptParams(1, 3, 1.0, 1e-3, 1e-5)):
        Tg, stg, camc = _run(gpu_ctx, cam, res, np.eye(3, 4), prm)
        Tr, sr = O.motion_only(res[0], camc, np.eye(3, 4), prm)
        np.testing.assert_allclose(Tg[0], Tr, rtol=0, atol=1e-9)
        np.testing.assert_allclose(stg[0].chi2, sr.chi2, rtol=1e-9)


def test_motion_only_degenerate_lists(gpu_ctx):
    """empty list (the reference asserts): status 1, pose untouched; already converged pose: first step rejected path."""
    import oracle as O
    from scavislam_amd import synth
    rng = np.random.default_rng(8)
    cam = synth.CAM_DEFAULT
    T_true = synth.pose(np.eye(3), np.array([0.0, 0.0, 0.05]))
    res = _synthetic_results(rng, cam, T_true, 64, outliers=0.0, n_fail=0)
    empty = res.copy()
    empty["status"] = 5
    Tg, stg, camc = _run(gpu_ctx, cam, np.stack([empty, res]), T_true)
    assert stg[0].status == 1 and stg[0].num_obs == 0 and np.array_equal(Tg[0], T_true)
    Tr, sr = O.motion_only(res, camc, T_true)
    np.testing.assert_allclose(Tg[1], Tr, rtol=0, atol=1e-9)
    assert sr.status == 0 and O.motion_only(empty, camc, T_true)[1].status == 1


def test_motion_only_behind_the_matcher(gpu_ctx, scene_frames):
    """matchAndTrack's tail: GuidedMatcher::match results feed calcFastMotionOnly on the device (no host round trip)."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import FastGrid, FramePyramid, GuidedMatcher, PoseOptimizer
    ctx, stream = gpu_ctx
    cam, poses, frames = scene_frames["cam"], scene_frames["poses"], scene_frames["frames"]
    kf = FramePyramid(ctx, stream, cam, batch=1, with_float=False)
    cur = FramePyramid(ctx, stream, cam, batch=1, with_float=False)
    kf.upload(frames[0][0][None], frames[0][1][None])
    cur.upload(frames[2][0][None], frames[2][1][None])
    kf.preprocessing(); cur.preprocessing()
    fast = FastGrid(ctx, cur)
    fast.detectAdaptively(trials=5)
    pts = synth.candidate_points(np.random.default_rng(1), cam, frames[0][1], poses[0], (600, 300, 100))
    T_true = synth.pose_mul(poses[2], synth.pose_inv(poses[0]))
    T_guess = synth.pose_mul(synth.pose(synth.so3_exp(np.array([0.002, -0.003, 0.001])), np.array([0.01, 0.0, -0.01])), T_true)
    m = GuidedMatcher(ctx, cur, fast)
    res = m.match([(kf.pyr, 0, poses[0].reshape(12))], T_guess.reshape(12), poses[0].reshape(12), pts)[0]
    assert (res["status"] == 0).sum() >= 20      # matchAndTrack returns false below 20 observations (:1053)
    T, stats = PoseOptimizer(ctx, cur).calcFastMotionOnly(m, T_guess.reshape(12))
    Tr, sr = O.motion_only(res, cur.cams[0], T_guess)
    np.testing.assert_allclose(T[0], Tr, rtol=0, atol=1e-9)
    assert stats[0].num_obs == sr.num_obs and stats[0].chi2 <= stats[0].initial_chi2
    assert np.abs(T[0] - T_true).max() < 0.05      # stays near the truth (the candidate points carry a few px of noise)
    # processMatchedPoints right behind it, at the refined pose that never left the device
    po = PoseOptimizer(ctx, cur)
    T2, _ = po.calcFastMotionOnly(m, T_guess.reshape(12))
    gated, pstats = po.processMatchedPoints(m, n_new_records=600)
    g_ref, s_ref = O.process_matched_points(res, pts, 600, cur.cams[0], T2[0])
    _check_gate(gated[0], pstats[0], g_ref, s_ref)
    assert pstats[0]["num_track_points"] >= 20


def _check_gate(gated, stats, gated_ref, stats_ref):
    for k in ("accepted", "is_new", "uv_pyr", "curkey_uv_pyr"):
        assert np.array_equal(gated[k], gated_ref[k]), k
    for k in ("num_points_grid2x2", "num_points_grid3x3", "num_matched_points", "num_track_points", "num_obs"):
        assert np.array_equal(stats[k], stats_ref[k]), k
    np.testing.assert_allclose(stats["sum_track_length"], stats_ref["sum_track_length"], rtol=1e-12)


def test_process_matched_points_matches_oracle(gpu_ctx):
    """StereoFrontend::processMatchedPoints (stereo_frontend.cpp:834-974): gate flags, level positions and integer
    statistics bit-exact; the track-length sum within 1e-12 (parallel summation order)."""
    import torch
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.ctypes_types import CANDIDATE_DTYPE, GATED_POINT_DTYPE, POINT_STATS_DTYPE, Cam
    ctx, stream = gpu_ctx
    rng = np.random.default_rng(5)
    cam = synth.CAM_DEFAULT
    camc = Cam(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
    T = synth.pose(synth.so3_exp(np.array([0.01, -0.02, 0.005])), np.array([0.03, -0.01, 0.08]))
    B, n, n_new = 3, 700, 260
    res = np.stack([_synthetic_results(rng, cam, T, n, outliers=0.25) for _ in range(B)])
    # residuals straddling the three thresholds (2 * 2^level px in u and v, 6 px in u_right)
    res["obs"] += rng.choice([0.0, 1.9, 2.1, 3.9, 4.1, 5.9, 6.1, 7.9, 8.1], size=res["obs"].shape) * rng.choice([-1, 1], size=res["obs"].shape)
    pts = np.zeros((B, n), CANDIDATE_DTYPE)
    pts["anchor_level"] = rng.integers(0, 3, (B, n))
    with torch.cuda.stream(stream):
        d_res = torch.as_tensor(res.view(np.uint8).reshape(-1)).cuda()
        d_pts = torch.as_tensor(pts.view(np.uint8).reshape(-1)).cuda()
        d_T = torch.as_tensor(np.tile(T.reshape(12), (B, 1))).cuda()
        d_g = torch.zeros(B * n * GATED_POINT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        d_s = torch.zeros(B * POINT_STATS_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    for mre in (2.0, 0.75, 5.0):
        ctx.call("svs_process_matched_points", d_res.data_ptr(), d_pts.data_ptr(), n, n, n, n_new, C.byref(camc), d_T.data_ptr(), mre,
                 d_g.data_ptr(), n, d_s.data_ptr(), B)
        ctx.sync()
        gated = d_g.cpu().numpy().view(GATED_POINT_DTYPE).reshape(B, n)
        stats = d_s.cpu().numpy().view(POINT_STATS_DTYPE)
        for b in range(B):
            g_ref, s_ref = O.process_matched_points(res[b], pts[b], n_new, camc, T, mre)
            _check_gate(gated[b], stats[b], g_ref, s_ref)
            assert 0 < s_ref["num_track_points"] < s_ref["num_obs"] and s_ref["num_points_grid3x3"].sum() == s_ref["num_track_points"]
            assert g_ref["is_new"].sum() == g_ref["accepted"][:n_new].sum()
    # empty list
    ctx.call("svs_process_matched_points", None, None, 0, 0, 0, 0, C.byref(camc), d_T.data_ptr(), 2.0, None, 0, d_s.data_ptr(), B)
    ctx.sync()
    stats = d_s.cpu().numpy().view(POINT_STATS_DTYPE)
    assert (stats["num_obs"] == 0).all() and (stats["sum_track_length"] == 0).all()
