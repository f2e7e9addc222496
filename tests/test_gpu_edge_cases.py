"""Edge cases through the C ABI on the GPU: empty / ragged / degenerate inputs must behave like the oracle
(or fail with a status code), never crash or hang."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cam(c):
    from scavislam_amd.ctypes_types import Cam
    return Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])


def test_ba_degenerate_windows(gpu_ctx):
    """(a) landmarks observed only by their anchor (pure self edges), (b) landmarks without any edge,
    (c) a window with constraints only, (d) a single pose -- all must follow the oracle."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BA_EDGE_DTYPE, BaParams
    ctx, stream = gpu_ctx
    prob = synth.ba_window(6, 120, seed=41, n_outer=2)
    cam = _cam(prob["cam"])
    e = prob["edges"]
    cases = {}
    keep = np.ones(len(e), bool)
    keep[(e["point"] < 40) & (e["pose"] != e["anchor"])] = False          # (a) first 40 landmarks: self edge only
    keep[(e["point"] >= 100)] = False                                      # (b) last 20 landmarks: no edges at all
    cases["ragged"] = (prob["poses"], prob["psi"], e[keep], prob["cons"])
    cases["constraints_only"] = (prob["poses"], prob["psi"], e[:0], prob["cons"])
    one = e[(e["pose"] == 0) & (e["anchor"] == 0)].copy()
    cases["single_pose"] = (prob["poses"][:1], prob["psi"], one, prob["cons"][:0])
    for name, (poses, psi, edges, cons) in cases.items():
        for mode in (0, 1):
            prm = BaParams.reference_defaults()
            prm.self_edge_mode = mode
            opt = SlamGraphOptimizer(ctx, stream)
            opt.copyDataToG2o(poses, psi, edges, cons, cam, prm)
            st = opt.optimize()
            p, s = opt.restoreDataFromG2o()
            pr, sr, str_ = O.ba_optimize(poses, psi, edges, cons, cam, prm)
            assert (st.iterations, st.trials, st.accepted, st.terminated) == (str_.iterations, str_.trials, str_.accepted, str_.terminated), name
            scale_p = max(np.abs(pr - poses).max(), 1e-12)
            scale_s = max(np.abs(sr - psi).max(), 1e-12)
            assert np.abs(p - pr).max() <= 1e-6 * scale_p + 1e-12, (name, mode)
            assert np.abs(s - sr).max() <= 1e-6 * scale_s + 1e-12, (name, mode)
            opt.close()


def test_ba_rejects_unsupported_and_bad_input(gpu_ctx):
    from scavislam_amd import capi, synth
    from scavislam_amd.backend import SlamGraphOptimizer
    ctx, stream = gpu_ctx
    prob = synth.ba_window(5, 30, seed=2, n_outer=0)
    cam = _cam(prob["cam"])
    bad = prob["edges"].copy()
    bad["pose"][0] = 99                                                    # pose index out of range
    opt = SlamGraphOptimizer(ctx, stream)
    with pytest.raises(capi.SvsError):
        opt.copyDataToG2o(prob["poses"], prob["psi"], bad, prob["cons"], cam)
    bad = prob["edges"].copy()
    pt = bad["point"][0]
    idx = np.nonzero(bad["point"] == pt)[0]
    if len(idx) > 1:
        bad["anchor"][idx[1]] = (bad["anchor"][idx[1]] + 1) % 5              # two anchors for one point
        with pytest.raises(capi.SvsError):
            opt.copyDataToG2o(prob["poses"], prob["psi"], bad, prob["cons"], cam)
    big = np.tile(prob["poses"][:1], (300, 1))                             # P > 256: documented limit, status 5
    with pytest.raises(capi.SvsError, match="status 5"):
        opt.copyDataToG2o(big, prob["psi"], prob["edges"], prob["cons"], cam)
    opt.close()


def test_matcher_empty_and_all_rejected(gpu_ctx, scene_frames):
    """n = 0 candidate points; and a set where every point is rejected (no corners in any window)."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.ctypes_types import CANDIDATE_DTYPE
    from scavislam_amd.frontend import FastGrid, FramePyramid, GuidedMatcher
    ctx, stream = gpu_ctx
    cam = scene_frames["cam"]
    flat = np.full((cam["h"], cam["w"]), 90, np.uint8)                      # no texture, no corners
    disp = np.full((cam["h"], cam["w"]), 5.0, np.float32)
    fr = FramePyramid(ctx, stream, cam, batch=1, with_float=False)
    fr.upload(flat[None], disp[None])
    fr.preprocessing()
    fg = FastGrid(ctx, fr)
    fg.detectAdaptively(trials=6)
    gm = GuidedMatcher(ctx, fr, fg)
    I = np.hstack([np.eye(3), np.zeros((3, 1))]).reshape(12)
    res = gm.match([(fr.pyr, 0, I)], I, I, np.zeros(0, CANDIDATE_DTYPE))
    assert res.shape == (1, 0)
    rng = np.random.default_rng(0)
    pts = synth.candidate_points(rng, cam, disp, I.reshape(3, 4), (50, 20, 10))
    res = gm.match([(fr.pyr, 0, I)], I, I, pts)[0]
    pyr = O.build_pyramid(flat)
    trees = [O.QuadTree(pyr[l].shape[1], pyr[l].shape[0]) for l in range(3)]
    ref = O.match([pyr], [I], I.reshape(3, 4), I.reshape(3, 4), pyr, disp, trees, fr.cams, pts)
    # the reference's "texture gate" compares sumA^2 - sumAA (sic, not a variance): a flat patch PASSES it
    # (matcher.cpp:384-386); with no corners in the window every point then ends as "no candidate"
    assert np.array_equal(res["status"], ref["status"]) and np.all(ref["status"] == 5)


def test_dense_tracker_without_valid_depth(gpu_ctx, scene_frames):
    """Disparity <= 0 everywhere: the cloud is all-invalid, H is singular, the reference's undamped solve
    produces NaN and every step is rejected -- the pose must come back unchanged, as in the oracle."""
    import oracle as O
    from scavislam_amd.frontend import DenseTracker, FramePyramid
    ctx, stream = gpu_ctx
    cam = scene_frames["cam"]
    img = scene_frames["frames"][1][0]
    disp = np.zeros((cam["h"], cam["w"]), np.float32)
    fr = FramePyramid(ctx, stream, cam, batch=1)
    fr.upload(img[None], disp[None])
    fr.preprocessing()
    dt = DenseTracker(ctx, fr)
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    dt.computeDensePointCloudCpu(I.reshape(12))
    for from_u8 in (False, True):
        T, passes = dt.denseTrackingCpu(fr.pyr, I.reshape(12), from_u8=from_u8)
        assert np.array_equal(T[0], I)
    clouds = [O.pointcloud_cpu(disp, fr.cams[l], l, I) for l in range(3)]
    pyr = O.build_pyramid(img)
    fl = [O.convert_sobel(p) for p in pyr]
    T_ref, _ = O.dense_tracking_cpu(clouds, pyr, [f[0] for f in fl], [f[1] for f in fl], [f[2] for f in fl], fr.cams, I)
    assert np.array_equal(T_ref, I)


def test_fast_small_image_and_capacity(gpu_ctx):
    """Smallest sensible frame (cells barely larger than the 3-px ROI border) and a corner capacity
    that is too small: counts stay exact, the download reports SVS_ERR_CAPACITY (status 4)."""
    import oracle as O
    from scavislam_amd import capi, synth
    from scavislam_amd.frontend import FastGrid, FramePyramid
    ctx, stream = gpu_ctx
    w, h = 64, 48
    cam = dict(synth.CAM_DEFAULT, w=w, h=h, cx=32.0, cy=24.0)
    img = synth.noise_image(w, h, 77)
    fr = FramePyramid(ctx, stream, cam, batch=1, with_float=False)
    fr.upload(img[None])
    fr.preprocessing()
    fg = FastGrid(ctx, fr)
    pyr = O.build_pyramid(img)
    grids = [O.fastgrid_for_level(pyr[l].shape[1], pyr[l].shape[0], l) for l in range(3)]
    for it in range(4):
        fg.detectAdaptively(trials=6)
        for l in range(3):
            xy_ref, cc_ref, et_ref = O.fastgrid_detect_adaptively(grids[l], pyr[l], 6)
            xy, cc, et, ts = fg.corners(0, l)
            assert np.array_equal(xy, xy_ref) and np.array_equal(cc, cc_ref) and np.array_equal(et, et_ref)
    img2 = synth.noise_image(640, 480, 5)
    fr2 = FramePyramid(ctx, stream, synth.CAM_DEFAULT, batch=1, with_float=False)
    fr2.upload(img2[None])
    fr2.preprocessing()
    fg2 = FastGrid(ctx, fr2, corner_cap=100)
    fg2.detectAdaptively(trials=6)
    with pytest.raises(capi.SvsError, match="status 4"):
        fg2.corners(0, 0)
    # a cell whose corner bitmap does not fit the compaction's LDS (2560 x 1440 pixels as ONE cell: 461 KB of bits) is refused at create, before anything is
    # allocated -- SVS_ERR_INVALID (status 1), not a launch failure later
    from scavislam_amd.frontend import fastgrid_for_level
    camb = dict(synth.CAM_DEFAULT, w=2560, h=1440, cx=1280.0, cy=720.0)
    frb = FramePyramid(ctx, stream, camb, batch=1, with_float=False)
    gb = []
    for l in range(3):
        g = fastgrid_for_level(frb.w[l], frb.h[l], l)
        g.gx = g.gy = 1
        g.cell_w, g.cell_h = frb.w[l], frb.h[l]
        gb.append(g)
    with pytest.raises(capi.SvsError, match="status 1"):
        FastGrid(ctx, frb, grids=gb)
