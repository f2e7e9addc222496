"""svs_frontend_process_frame -- StereoFrontend::processFrame as ONE library call per frame (host buffers in, host buffers out) -- against
the same stages driven one by one through their own entry points (which tests/test_gpu_newcollege_chain.py and the per-stage tests
hold against the oracle): every output must be IDENTICAL, because both run the same kernels on the same inputs in the same order.
Additionally the refined pose is checked against the oracle's calcFastMotionOnly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block_matching", [False, True])
def test_process_frame_equals_staged_chain(gpu_ctx, block_matching):
    import oracle as O
    from scavislam_amd import capi, synth
    from scavislam_amd.frontend import (DenseTracker, FastGrid, FramePyramid, GuidedMatcher, PoseOptimizer, StereoFrontend, StereoMatcher)
    ctx, stream = gpu_ctx
    cam = synth.CAM_NEWCOLLEGE
    sc = synth.Scene(2011)
    traj = synth.trajectory(8)
    kf_i, prev_i, cur_i = 0, 4, 5
    fr = {n: synth.render_stereo(sc, cam, traj[i], seed=s) for n, i, s in (("kf", kf_i, 1), ("prev", prev_i, 2), ("cur", cur_i, 3))}
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    T_guess = synth.pose_mul(synth.pose(synth.so3_exp(np.array([0.001, -0.002, 0.0005])), np.array([0.004, 0.0, -0.004])),
                             synth.pose_mul(traj[cur_i], synth.pose_inv(traj[prev_i])))          # motion-model guess T_cur_from_actkey (actkey = prev)

    # ---- the stages one by one
    frames = {}
    for name in ("kf", "prev", "cur"):
        L, R, d_true = fr[name]
        f = FramePyramid(ctx, stream, cam, batch=1, with_float=False)
        if block_matching:
            f.upload(L[None])
            sm = StereoMatcher(ctx, f)
            sm.upload_right(R[None])
            sm.calcDisparityCpu()
            ctx.sync()
            sm.close()
        else:
            f.upload(L[None], d_true[None])
        f.preprocessing(with_float=False)
        frames[name] = f
    kf, prev, cur = frames["kf"], frames["prev"], frames["cur"]
    fast = FastGrid(ctx, cur)
    fast.detectAdaptively(pyr=kf.pyr, trials=5)          # the frontend object's FastGrid saw: first frame (kf), first frame again (prev), cur
    fast.detectAdaptively(pyr=prev.pyr, trials=5)
    dtp = DenseTracker(ctx, prev)
    dtp.computeDensePointCloudCpu(I.reshape(12))
    dt = DenseTracker(ctx, cur)
    dt.ref_dense_points = dtp.ref_dense_points
    T_trk, passes = dt.denseTrackingCpu(prev.pyr, T_guess.reshape(12), from_u8=True)
    fast.detectAdaptively(trials=6)
    rng = np.random.default_rng(7)
    disp_kf = kf.disp[0, :, :kf.w[0]].cpu().numpy()
    pts = synth.candidate_points(rng, cam, np.maximum(disp_kf, 0), traj[kf_i], (500, 250, 80))
    m = GuidedMatcher(ctx, cur, fast)
    res = m.match([(kf.pyr, 0, traj[kf_i].reshape(12))], T_trk[0].reshape(12), traj[prev_i].reshape(12), pts)[0]
    po = PoseOptimizer(ctx, cur)
    T_mo, st = po.calcFastMotionOnly(m, T_trk[0].reshape(12))
    gated, pstats = po.processMatchedPoints(m, n_new_records=500)
    dt.computeDensePointCloudCpu(T_mo[0].reshape(12))
    ctx.sync()
    clouds = [dt.ref_dense_points[l][0].cpu().numpy() for l in range(3)]

    # ---- one call per frame
    fe = StereoFrontend(ctx, cam, max_points=2048, max_keyframes=4, params=capi.FrontendParams.reference(use_block_matching=block_matching))
    kw = (lambda n: dict(right=fr[n][1])) if block_matching else (lambda n: dict(disp=fr[n][2]))
    fe.processFirstFrame(fr["kf"][0], **kw("kf"))
    fe.keepKeyframe(0, traj[kf_i])
    fe.processFirstFrame(fr["prev"][0], **kw("prev"))                 # the active keyframe: reference cloud at the identity
    fe.setCandidates(pts, 500)
    out, matches, g = fe.processFrame(fr["cur"][0], T_guess, traj[prev_i], **kw("cur"))

    assert out.dense_passes == passes[0] and out.n_points == len(pts)
    assert np.array_equal(np.array(out.T_cur_from_actkey).reshape(3, 4), T_mo[0])
    for k in ("status", "u", "v", "znssd", "obs", "xyz_actkey"):
        assert np.array_equal(matches[k], res[k]), k
    ok = res["status"] == 0
    assert out.n_matched == int(ok.sum()) == st[0].num_obs and out.tracking_ok == 1
    for k in ("accepted", "is_new", "uv_pyr", "curkey_uv_pyr"):
        assert np.array_equal(g[k][ok], gated[0][k][ok]), k
    for k in ("num_points_grid2x2", "num_points_grid3x3", "num_matched_points"):
        assert np.array_equal(np.array(getattr(out.point_stats, k)), pstats[0][k]), k
    assert out.point_stats.num_track_points == pstats[0]["num_track_points"] and out.point_stats.num_obs == pstats[0]["num_obs"]
    assert out.pose_stats.chi2 == st[0].chi2 and out.pose_stats.initial_chi2 == st[0].initial_chi2
    for l in range(3):
        assert np.array_equal(fe.cloud_host(l), clouds[l]), f"reference cloud level {l}"
    # and against the oracle where the chain test does not already: the refined pose
    T_mo_ref, _ = O.motion_only(res, cur.cams[0], T_trk[0])
    np.testing.assert_allclose(np.array(out.T_cur_from_actkey).reshape(3, 4), T_mo_ref, rtol=0, atol=1e-9)
    fe.close()
