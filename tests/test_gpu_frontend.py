"""GPU parity of the per-frame front-end against the CPU oracle (all through the C ABI).

Bars (BASELINE.json north_star): integer paths bit-exact (pyramid, FAST corner lists incl. order
and thresholds, ZNSSD best match incl. tie-breaks, u8 key-patch warp); f32/f64 dense-tracking sums
within the tolerances written next to each assert.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame(ctx, stream, cam, imgs, disps=None, with_float=True):
    from scavislam_amd.frontend import FramePyramid
    fr = FramePyramid(ctx, stream, cam, batch=len(imgs), with_float=with_float)
    fr.upload(np.stack(imgs), None if disps is None else np.stack(disps))
    fr.preprocessing()
    return fr


@pytest.mark.parametrize("w,h", [(640, 480), (512, 384), (322, 242)])
def test_pyramid_and_sobel_bit_exact(gpu_ctx, w, h):
    import oracle as O
    from scavislam_amd import synth
    ctx, stream = gpu_ctx
    cam = dict(synth.CAM_DEFAULT, w=w, h=h)
    imgs = [synth.noise_image(w, h, 11 + i) for i in range(2)]
    if (w, h) == (322, 242):           # odd level sizes exercise REFLECT_101 on both borders
        from scavislam_amd.frontend import FramePyramid
        import torch
        fr = FramePyramid(ctx, stream, dict(cam, w=320, h=240), batch=1, with_float=False)
        # direct C-ABI call on an odd-sized image
        src = torch.as_tensor(imgs[0]).cuda()
        dst = torch.zeros(((h + 1) // 2, (w + 1) // 2), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.call("svs_pyr_down_u8", src.data_ptr(), w, h, w, w * h, dst.data_ptr(), (w + 1) // 2, 0, 1)
        ctx.sync()
        assert np.array_equal(dst.cpu().numpy(), O.pyr_down_u8(imgs[0]))
        return
    fr = _frame(ctx, stream, cam, imgs)
    ctx.sync()
    for b, img in enumerate(imgs):
        pyr = O.build_pyramid(img)
        for l in range(3):
            assert np.array_equal(fr.level_host(l, b), pyr[l]), f"pyramid level {l}"
            f, dx, dy = O.convert_sobel(pyr[l])
            wl = fr.w[l]
            assert np.array_equal(fr.f32[l][b, :, :wl].cpu().numpy(), f)      # bit-exact f32
            assert np.array_equal(fr.dx[l][b, :, :wl].cpu().numpy(), dx)
            assert np.array_equal(fr.dy[l][b, :, :wl].cpu().numpy(), dy)


@pytest.mark.parametrize("w,h", [(640, 480), (512, 384)])
def test_fastgrid_adaptive_sequence_bit_exact(gpu_ctx, w, h):
    """Six consecutive frames: corner lists (order included), per-cell counts, emit thresholds and
    the persistent thresholds must equal the oracle's after every frame, for two camera streams."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import FastGrid
    ctx, stream = gpu_ctx
    cam = dict(synth.CAM_DEFAULT, w=w, h=h)
    seqs = [[synth.noise_image(w, h, 100 * s + i) for i in range(6)] for s in range(2)]
    fr = _frame(ctx, stream, cam, [seqs[0][0], seqs[1][0]], with_float=False)
    fg = FastGrid(ctx, fr)
    grids = [[O.fastgrid_for_level(fr.w[l], fr.h[l], l) for l in range(3)] for _ in range(2)]
    for i in range(6):
        fr.upload(np.stack([seqs[0][i], seqs[1][i]]))
        fr.preprocessing()
        trials = 5 if i == 0 else 6            # first frame uses 5 (stereo_frontend.cpp:131-136)
        fg.detectAdaptively(trials=trials)
        for s in range(2):
            pyr = O.build_pyramid(seqs[s][i])
            for l in range(3):
                xy_ref, cc_ref, et_ref = O.fastgrid_detect_adaptively(grids[s][l], pyr[l], trials)
                xy, cc, et, ts = fg.corners(s, l)
                nc = grids[s][l].gx * grids[s][l].gy
                assert np.array_equal(cc, cc_ref), (i, s, l, cc, cc_ref)
                assert np.array_equal(et, et_ref)
                assert np.array_equal(ts, np.array(grids[s][l].thr[:nc]))
                assert np.array_equal(xy, xy_ref)
    # FastGrid::detect (static thresholds) on the last frame
    fg.detect()
    for s in range(2):
        pyr = O.build_pyramid(seqs[s][5])
        for l in range(3):
            xy_ref, cc_ref = O.fastgrid_detect(grids[s][l], pyr[l])
            xy, cc, et, ts = fg.corners(s, l)
            assert np.array_equal(xy, xy_ref) and np.array_equal(cc, cc_ref)
            # the corner bitmap the matcher reads (svs_fast_device_view: 1 bit per pixel, cell columns on dword boundaries) holds exactly these corners
            bits = fg.corner_bits(s, l)
            ref_bits = np.zeros_like(bits)
            ref_bits[xy_ref[:, 1], xy_ref[:, 0]] = True
            assert np.array_equal(bits, ref_bits), (s, l, int(bits.sum()), len(xy_ref))


def test_fast_empty_and_saturated_images(gpu_ctx):
    """Edge cases: constant image (no corners, thresholds walk down to fast_min) and a checkerboard
    (far too many corners, thresholds walk up to fast_max)."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import FastGrid
    ctx, stream = gpu_ctx
    w, h = 640, 480
    flat = np.full((h, w), 77, np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    checker = (((xx // 4) + (yy // 4)) % 2 * 200 + 20).astype(np.uint8)
    fr = _frame(ctx, stream, dict(synth.CAM_DEFAULT), [flat, checker], with_float=False)
    fg = FastGrid(ctx, fr, corner_cap=60000)
    grids = [[O.fastgrid_for_level(fr.w[l], fr.h[l], l) for l in range(3)] for _ in range(2)]
    for it in range(10):
        fg.detectAdaptively(trials=6)
        for s, img in enumerate((flat, checker)):
            pyr = O.build_pyramid(img)
            for l in range(3):
                xy_ref, cc_ref, et_ref = O.fastgrid_detect_adaptively(grids[s][l], pyr[l], 6, cap=1 << 17)
                xy, cc, et, ts = fg.corners(s, l)
                assert np.array_equal(cc, cc_ref) and np.array_equal(et, et_ref)
                assert np.array_equal(xy, xy_ref)
    assert all(t == 10 for t in grids[0][0].thr[:9])


def _match_setup(scene_frames, ctx, stream, n_per_level=(300, 150, 60), seed=5):
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import FastGrid
    cam = scene_frames["cam"]
    (img_k, disp_k), (img_p, disp_p), (img_c, disp_c) = scene_frames["frames"]
    T_k, T_p, T_c = scene_frames["poses"]
    fr = _frame(ctx, stream, cam, [img_c], [disp_c])
    fg = FastGrid(ctx, fr)
    for _ in range(3):
        fg.detectAdaptively(trials=6)
    rng = np.random.default_rng(seed)
    pts = synth.candidate_points(rng, cam, disp_k, T_k, n_per_level)
    # a few degenerate candidates: unknown anchor, border anchor, far too close, behind the camera
    pts[0]["kf_index"] = -1
    pts[1]["anchor_obs_pyr"][:2] = (2.0, 2.0)
    pts[2]["xyz_anchor"] *= 0.05
    pts[3]["xyz_anchor"][2] = -1.0
    return cam, fr, fg, pts, (img_k, disp_k, T_k), (img_c, disp_c, T_c)


@pytest.mark.parametrize("legacy", [0, 1, 2])
@pytest.mark.parametrize("radius,thr_mean,thr_std", [(8, 22, 10), (4, 22, 10), (5, 12, 0), (10, 22, 10)])
def test_matcher_bit_exact(gpu_ctx, scene_frames, radius, thr_mean, thr_std, legacy):
    """GuidedMatcher::match: status, best corner, ZNSSD score bit-exact; obs / xyz_actkey to 1e-12.  The three parameter sets the oracle is pinned on against
    the reference-compiled matcher (tests/test_ref_pin_cpu.py): radius 8 = the CPU build, radius 4 = the CUDA build (stereo_frontend.cpp:1043-1047), and
    (12, 0, 5) = non-default thresholds, radius 10 = a window wider than 17 (beyond the four-points-per-wave kernel: the library falls back to one wave
    per point); each with the kernel choice left to the library (0) and forced to the round-1/2 kernel (1) and the lean one-wave-per-point kernel (2)."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import FramePyramid, GuidedMatcher
    ctx, stream = gpu_ctx
    cam, fr, fg, pts, (img_k, disp_k, T_k), (img_c, disp_c, T_c) = _match_setup(scene_frames, ctx, stream)
    kf = FramePyramid(ctx, stream, cam, batch=1, with_float=False)
    kf.upload(img_k[None])
    kf.preprocessing()
    # active keyframe = anchor keyframe; predicted pose = true relative pose perturbed by ~1 px
    T_cur_from_actkey = synth.pose_mul(T_c, synth.pose_inv(T_k))
    T_cur_from_actkey[:, 3] += np.array([0.004, -0.003, 0.002])
    gm = GuidedMatcher(ctx, fr, fg)
    ctx.set_option("match_legacy", legacy)
    try:
        res = gm.match([(kf.pyr, 0, T_k.reshape(12))], T_cur_from_actkey.reshape(12), T_k.reshape(12), pts, radius, thr_mean, thr_std)[0]
    finally:
        ctx.set_option("match_legacy", 0)
    # oracle on the same corners (taken from the oracle's own FAST, proven equal in the FAST test)
    pyr_c, pyr_k = O.build_pyramid(img_c), O.build_pyramid(img_k)
    trees = []
    for l in range(3):
        xy, cc, et, ts = fg.corners(0, l)
        trees.append(O.quadtree_from_corners(xy, cc, pyr_c[l].shape[1], pyr_c[l].shape[0]))
    ref = O.match([pyr_k], [T_k.reshape(12)], T_cur_from_actkey, T_k, pyr_c, disp_c, trees, fr.cams, pts, radius, thr_mean, thr_std)
    assert np.array_equal(res["status"], ref["status"])
    ok = ref["status"] == 0
    assert ok.sum() > (100 if thr_std else 40), "test scene should produce plenty of matches"
    found = (ref["status"] == 0) | (ref["status"] == 6)
    assert np.array_equal(res["u"][found], ref["u"][found]) and np.array_equal(res["v"][found], ref["v"][found])
    assert np.array_equal(res["znssd"], ref["znssd"])
    assert np.array_equal(res["obs"][ok], ref["obs"][ok])          # exact: ints and one f32 disparity
    np.testing.assert_allclose(res["xyz_actkey"][found], ref["xyz_actkey"][found], rtol=1e-12, atol=1e-12)
    if (radius, thr_mean, thr_std) == (8, 22, 10):
        assert set(np.unique(ref["status"])) >= {0, 1, 2, 3}


def test_matcher_tie_break_follows_quadtree_order(gpu_ctx):
    """Periodic texture => many candidates with identical ZNSSD; the winner must be the first one
    in QuadTree::query DFS order, which the kernel reproduces with quadrant keys."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.ctypes_types import CANDIDATE_DTYPE
    from scavislam_amd.frontend import FastGrid, FramePyramid, GuidedMatcher
    ctx, stream = gpu_ctx
    cam = dict(synth.CAM_DEFAULT)
    h, w = cam["h"], cam["w"]
    yy, xx = np.mgrid[0:h, 0:w]
    tile = np.random.default_rng(3).integers(0, 256, (8, 8)).astype(np.uint8)
    img = tile[yy % 8, xx % 8]                      # exactly 8-periodic: ZNSSD ties everywhere
    disp = np.full((h, w), 6.0, np.float32)
    fr = _frame(ctx, stream, cam, [img], [disp])
    fg = FastGrid(ctx, fr, corner_cap=320 * 1024)
    fg.detectAdaptively(trials=6)
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    z = cam["f"] * cam["b"] / 6.0
    rng = np.random.default_rng(9)
    pts = np.zeros(200, CANDIDATE_DTYPE)
    for k in range(200):
        u0, v0 = int(rng.integers(40, w - 40)), int(rng.integers(40, h - 40))
        pts[k]["xyz_anchor"] = ((u0 - cam["cx"]) / cam["f"] * z, (v0 - cam["cy"]) / cam["f"] * z, z)
        pts[k]["anchor_obs_pyr"] = (u0, v0, u0 - 6.0)
    gm = GuidedMatcher(ctx, fr, fg)
    res = gm.match([(fr.pyr, 0, I.reshape(12))], I.reshape(12), I.reshape(12), pts, thr_std=0)[0]
    pyr = O.build_pyramid(img)
    trees = []
    for l in range(3):
        xy, cc, et, ts = fg.corners(0, l)
        trees.append(O.quadtree_from_corners(xy, cc, pyr[l].shape[1], pyr[l].shape[0]))
    ref = O.match([pyr], [I.reshape(12)], I, I, pyr, disp, trees, fr.cams, pts, thr_std=0)
    assert np.array_equal(res["status"], ref["status"])
    assert (ref["status"] == 0).sum() > 50
    assert np.array_equal(res["u"], ref["u"]) and np.array_equal(res["v"], ref["v"])
    assert np.array_equal(res["znssd"], ref["znssd"])


@pytest.mark.parametrize("legacy", [0, 1, 2])
def test_matcher_one_cell_column_and_windows_over_the_left_border(gpu_ctx, legacy):
    """The matcher reads FAST's corner bitmap, whose cell columns start on dword boundaries (pad bits between them).  Two corners of that layout: a grid with ONE cell
    column whose width is no multiple of 32 (pad bits but no column boundary to cross), and search windows that hang over the image's left / right border (bits left of
    column 0 read as 0, the window's first in-image column is column 0).  All three kernels against the oracle on the same corners."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.ctypes_types import CANDIDATE_DTYPE
    from scavislam_amd.frontend import FastGrid, GuidedMatcher, fastgrid_for_level
    ctx, stream = gpu_ctx
    cam = dict(synth.CAM_DEFAULT)
    h, w = cam["h"], cam["w"]
    img = np.random.default_rng(41).integers(0, 256, (h, w)).astype(np.uint8)      # white noise: corners everywhere, up to the first and last column a cell can have
    disp = np.full((h, w), 6.0, np.float32)
    fr = _frame(ctx, stream, cam, [img], [disp])
    grids = []
    for l in range(3):
        g = fastgrid_for_level(fr.w[l], fr.h[l], l)
        g.gx, g.gy = 1, 2
        g.cell_w, g.cell_h = fr.w[l] - 10, fr.h[l] // 2          # 630 / 310 / 150 columns: 10 / 10 / 10 pad bits behind the only cell column
        grids.append(g)
    fg = FastGrid(ctx, fr, corner_cap=320 * 1024, grids=grids)
    for l in range(3):
        fg.set_thresholds(0, l, [24, 24])                       # FastGrid::detect at a low static threshold: a dense corner set (windows with dozens of hits: more than the
    fg.detect()                                                 # 16 a point's lanes take at a time), corners in the first and last columns a cell can have
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    z = cam["f"] * cam["b"] / 6.0
    rng = np.random.default_rng(19)
    rows, n_left, n_right = [], 0, 0
    for lvl in range(3):
        xy = fg.corners(0, lvl)[0].astype(np.int64)
        s = 1 << lvl
        inner = xy[(xy[:, 1] >= 12) & (xy[:, 1] < (h >> lvl) - 12)]
        # anchors ON corners (identity motion: the corner itself is the perfect match): the ones within 8 pixels of the left border (their window starts left of the
        # image), the ones at the right end of the cell column, and a sample of the rest
        left, right = inner[inner[:, 0] < 8], inner[inner[:, 0] >= grids[lvl].cell_w - 12]
        rest = inner[rng.permutation(len(inner))[:60]]
        n_left += len(left[:40]); n_right += len(right[:40])
        for ul, vl in np.concatenate([left[:40], right[:40], rest]):
            r = np.zeros(1, CANDIDATE_DTYPE)
            u0, v0 = ul * s, vl * s
            r["xyz_anchor"] = ((u0 - cam["cx"]) / cam["f"] * z, (v0 - cam["cy"]) / cam["f"] * z, z)
            r["anchor_obs_pyr"] = (ul, vl, (u0 - 6.0) / s)
            r["anchor_level"] = lvl
            rows.append(r)
    pts = np.concatenate(rows)
    assert n_left >= 10 and n_right >= 10, (n_left, n_right)      # (the pyramid smooths the noise: most of them on level 0)
    gm = GuidedMatcher(ctx, fr, fg)
    ctx.set_option("match_legacy", legacy)
    try:
        res = gm.match([(fr.pyr, 0, I.reshape(12))], I.reshape(12), I.reshape(12), pts, thr_std=0)[0]
    finally:
        ctx.set_option("match_legacy", 0)
    pyr = O.build_pyramid(img)
    trees = []
    for l in range(3):
        xy, cc, et, ts = fg.corners(0, l)
        trees.append(O.quadtree_from_corners(xy, cc, pyr[l].shape[1], pyr[l].shape[0]))
        bits = fg.corner_bits(0, l)
        ref_bits = np.zeros_like(bits)
        ref_bits[xy[:, 1], xy[:, 0]] = True
        assert np.array_equal(bits, ref_bits), l
    ref = O.match([pyr], [I.reshape(12)], I, I, pyr, disp, trees, fr.cams, pts, thr_std=0)
    assert np.array_equal(res["status"], ref["status"])
    assert (ref["status"] == 0).sum() > 100, np.bincount(ref["status"])
    assert np.array_equal(res["u"], ref["u"]) and np.array_equal(res["v"], ref["v"])
    assert np.array_equal(res["znssd"], ref["znssd"])


def _dense_setup(scene_frames, ctx, stream):
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import DenseTracker, FramePyramid
    cam = scene_frames["cam"]
    (_, _), (img_p, disp_p), (img_c, disp_c) = scene_frames["frames"]
    _, T_p, T_c = scene_frames["poses"]
    prev = _frame(ctx, stream, cam, [img_p], [disp_p])
    cur = _frame(ctx, stream, cam, [img_c], [disp_c])
    return cam, prev, cur, (img_p, disp_p, T_p), (img_c, disp_c, T_c)


def test_dense_pointcloud_and_single_pass(gpu_ctx, scene_frames):
    """computeDensePointCloudCpu bit-exact (f32 out of identical f64 ops); one H,b / chi2 pass of the
    loop body: H and b within 1e-9 relative of the serial oracle (only the f64 summation order
    differs), n_valid exact, chi2 within 1e-4 relative (reference accumulates a float serially)."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import DenseTracker
    ctx, stream = gpu_ctx
    cam, prev, cur, (img_p, disp_p, T_p), (img_c, disp_c, T_c) = _dense_setup(scene_frames, ctx, stream)
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    # the cloud is built from the PREVIOUS frame's disparity and used while tracking the current one
    dt_prev = DenseTracker(ctx, prev)
    dt_prev.computeDensePointCloudCpu(I.reshape(12))
    ctx.sync()
    clouds_ref = [O.pointcloud_cpu(disp_p, prev.cams[l], l, I) for l in range(3)]
    for l in range(3):
        got = dt_prev.ref_dense_points[l][0].cpu().numpy()
        assert np.array_equal(got, clouds_ref[l]), f"cloud level {l}"
    dt = DenseTracker(ctx, cur)
    dt.ref_dense_points = dt_prev.ref_dense_points
    T_true = synth.pose_mul(T_c, synth.pose_inv(T_p))
    pyr_p = O.build_pyramid(img_p)
    pyr_c = O.build_pyramid(img_c)
    for l in range(3):
        f, dx, dy = O.convert_sobel(pyr_c[l])
        for T in (I, T_true):
            for do_jac in (False, True):
                ref = O.dense_pass_cpu(clouds_ref[l], pyr_p[l], f, dx, dy, cur.cams[l], T, do_jac)
                got = dt.pass_sums(l, prev.pyr, T.reshape(12), do_jac)[0]
                assert got["n_valid"] == ref["n_valid"] and ref["n_valid"] > 100
                np.testing.assert_allclose(got["chi2"], ref["chi2"], rtol=1e-4)   # oracle sums ~2e4 floats serially
                if do_jac:
                    scale = np.abs(ref["H"]).max()
                    np.testing.assert_allclose(got["H"], ref["H"], rtol=0, atol=1e-9 * scale)
                    np.testing.assert_allclose(got["b"], ref["b"], rtol=0, atol=1e-9 * np.abs(ref["b"]).max() + 1e-12)


def test_dense_tracking_device_resident_lm(gpu_ctx, scene_frames):
    """Whole denseTrackingCpu in one launch vs the oracle: the same LM trajectory record for record (the accept test takes the
    reference's decisions, SURVEY.md B-9 / csrc/seqsum.h), final pose within 1e-9.  Both must improve on the start pose."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import DenseTracker
    ctx, stream = gpu_ctx
    cam, prev, cur, (img_p, disp_p, T_p), (img_c, disp_c, T_c) = _dense_setup(scene_frames, ctx, stream)
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    dt = DenseTracker(ctx, cur)
    dtp = DenseTracker(ctx, prev)
    dtp.computeDensePointCloudCpu(I.reshape(12))
    dt.ref_dense_points = dtp.ref_dense_points
    T_gpu, passes = dt.denseTrackingCpu(prev.pyr, I.reshape(12))
    # fused source: f32 image + Sobel taps formed on the fly from the u8 pyramid -- same values, same
    # summation order => the tracked pose must be BIT-identical to the f32-pyramid path
    T_u8, passes_u8 = dt.denseTrackingCpu(prev.pyr, I.reshape(12), from_u8=True)
    assert np.array_equal(T_u8, T_gpu) and np.array_equal(passes_u8, passes)
    clouds = [O.pointcloud_cpu(disp_p, prev.cams[l], l, I) for l in range(3)]
    pyr_p, pyr_c = O.build_pyramid(img_p), O.build_pyramid(img_c)
    fl = [O.convert_sobel(p) for p in pyr_c]
    T_ref, passes_ref, rec_ref = O.dense_tracking_cpu(clouds, pyr_p, [f[0] for f in fl], [f[1] for f in fl], [f[2] for f in fl],
                                                      cur.cams, I, want_rec=True)
    T_true = synth.pose_mul(T_c, synth.pose_inv(T_p))
    err_gpu = np.abs(T_gpu[0] - T_true).max()
    err_ref = np.abs(T_ref - T_true).max()
    err_start = np.abs(I - T_true).max()
    assert err_ref < 0.5 * err_start and err_gpu < 0.5 * err_start
    np.testing.assert_allclose(T_gpu[0], T_ref, rtol=0, atol=1e-4)
    assert 3 <= passes[0] <= 3 * (1 + 15 * 2 * 3)
    # the LM trajectory itself (dense_tracking.cpp:367-385)
    _check_cpu_sem_trajectory(dt.lm_records()[0], passes[0], rec_ref, passes_ref, "scene_frames")
    np.testing.assert_allclose(T_gpu[0], T_ref, rtol=0, atol=1e-9)


def _check_cpu_sem_trajectory(rec, passes, rec_ref, passes_ref, label):
    """Accept / reject sequence and the chi2 of every trial of the device-resident denseTrackingCpu loop vs the oracle's -- EVERY record, to the end.
    The reference keeps chi2 in a `float` accumulated serially over the samples (dense_tracking.cpp:229,335) and accepts a step iff chi2 - new_chi2 > 0; its
    last steps of a level are decided by the rounding of those sums.  The default accept test of the library takes the same decisions: f64 sums
    where the difference is outside the rigorous error bound of the float sums, the float sums themselves (bit for bit, csrc/seqsum.h) inside it -- a record whose
    chi2 pair is bit-equal to the oracle's was decided on the float sums, the others carry the f64 sums narrowed (2e-5: the float sum's own noise).
    The reference repeats a rejected trial (identical step, identical rejection) before it stops at trial == 2; the device loop records it once."""
    ref = rec_ref.copy()
    dup = np.zeros(len(ref), bool)
    for k in range(1, len(ref)):
        if ref[k, 1] == 0 and ref[k - 1, 1] == 0 and ref[k, 0] == ref[k - 1, 0]:
            assert ref[k, 2] == ref[k - 1, 2] and ref[k, 3] == ref[k - 1, 3]      # the repeated trial really is identical
            dup[k] = True
    ref = ref[~dup]
    is_trial = ref[:, 1] < 2
    assert passes == len(rec), label                   # one fused pass per chi2 evaluation
    assert len(rec) == len(ref), (label, len(rec), len(ref))
    assert np.array_equal(rec["level"], ref[:, 0].astype(np.int32)), label
    assert np.array_equal(rec["accepted"], ref[:, 1].astype(np.int32)), (label, rec["accepted"], ref[:, 1])
    np.testing.assert_allclose(rec["chi2"], ref[:, 2], rtol=2e-5)
    np.testing.assert_allclose(rec["new_chi2"], ref[:, 3], rtol=2e-5)
    on_float_sums = is_trial & (rec["chi2"].astype(np.float32) == ref[:, 2].astype(np.float32)) & (rec["new_chi2"].astype(np.float32) == ref[:, 3].astype(np.float32))
    # the reference's pass count for this trajectory: per level 1 chi2 pass, 2 passes per trial, rejected trials twice
    assert passes_ref == 3 + 2 * (int(is_trial.sum()) + int(dup.sum())), label
    print(f"{label}: {int(is_trial.sum())} trials, all {len(ref)} records equal; {int(on_float_sums.sum())} trials decided on the float sums")
    return int(on_float_sums.sum())


@pytest.mark.parametrize("B", [8, 24, 40])
def test_dense_tracking_lm_trajectory_many_scenes(gpu_ctx, B):
    """B streams with different scenes / motions (B = 8: eight workgroups per stream, 24: four, 40: one): EVERY stream's accept / reject record equals the oracle's to the
    end -- in the default mode, the one bench.py times -- and the pose agrees to 1e-9 (same LM trajectory; the sums differ in their last bits)."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import DenseTracker, FramePyramid
    ctx, stream = gpu_ctx
    cam = synth.CAM_DEFAULT
    NS = 8                                            # distinct scenes; the streams beyond repeat them
    sc = synth.Scene(2011)
    rng = np.random.default_rng(5)
    base = synth.trajectory(6)
    T_prev = [base[b % 6] for b in range(NS)]
    T_cur = [synth.pose_mul(synth.pose(synth.so3_exp([rng.normal(0, 5e-4), np.deg2rad(rng.uniform(0.05, 0.5)), rng.normal(0, 5e-4)]),
                                       [rng.normal(0, 0.003), rng.normal(0, 0.002), -rng.uniform(0.02, 0.08)]), T_prev[b]) for b in range(NS)]
    prev_f = [sc.render(cam, T_prev[b], seed=500 + b) for b in range(NS)]
    cur_f = [sc.render(cam, T_cur[b], seed=600 + b) for b in range(NS)]
    prev = FramePyramid(ctx, stream, cam, batch=B)
    cur = FramePyramid(ctx, stream, cam, batch=B)
    prev.upload(np.stack([prev_f[b % NS][0] for b in range(B)]), np.stack([prev_f[b % NS][1] for b in range(B)]))
    cur.upload(np.stack([cur_f[b % NS][0] for b in range(B)]), np.stack([cur_f[b % NS][1] for b in range(B)]))
    prev.preprocessing(); cur.preprocessing()
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    dtp = DenseTracker(ctx, prev)
    dtp.computeDensePointCloudCpu(I.reshape(12))
    dt = DenseTracker(ctx, cur)
    dt.ref_dense_points = dtp.ref_dense_points
    n0 = ctx.get_stat("trk_exact_sums")
    T, passes = dt.denseTrackingCpu(prev.pyr, I.reshape(12), from_u8=True)
    recs = dt.lm_records()
    assert ctx.get_stat("trk_exact_sums") > n0 and ctx.get_stat("trk_exact_fallbacks") == 0
    n_float = 0
    refs = {}
    for b in range(B):
        s_ = b % NS
        if s_ not in refs:
            clouds = [O.pointcloud_cpu(prev_f[s_][1], prev.cams[l], l, I) for l in range(3)]
            pyr_p, pyr_c = O.build_pyramid(prev_f[s_][0]), O.build_pyramid(cur_f[s_][0])
            fl = [O.convert_sobel(p) for p in pyr_c]
            refs[s_] = O.dense_tracking_cpu(clouds, pyr_p, [f[0] for f in fl], [f[1] for f in fl], [f[2] for f in fl], cur.cams, I, want_rec=True)
        T_ref, passes_ref, rec_ref = refs[s_]
        n_float += _check_cpu_sem_trajectory(recs[b], passes[b], rec_ref, passes_ref, f"stream {b}")
        np.testing.assert_allclose(T[b], T_ref, rtol=0, atol=1e-9)
    assert n_float >= B, "the float sums were hardly ever needed: is the accept test's band too narrow?"
    # the same launch with every sum formed by the literal sequential chain ("trk_seq_chi2"): same records, and -- one workgroup per stream on both sides -- the same BITS
    ctx.set_option("trk_seq_chi2", 1)
    try:
        T_c, passes_c = dt.denseTrackingCpu(prev.pyr, I.reshape(12), from_u8=True)
        recs_c = dt.lm_records()
    finally:
        ctx.set_option("trk_seq_chi2", 0)
    for b in range(B):
        assert np.array_equal(recs_c[b]["accepted"], recs[b]["accepted"]) and np.array_equal(recs_c[b]["level"], recs[b]["level"]), b
    if B > 32:
        assert np.array_equal(T_c, T) and np.array_equal(passes_c, passes)
    else:
        np.testing.assert_allclose(T_c, T, rtol=0, atol=1e-12)


@pytest.mark.parametrize("nwg", [2, 16])
def test_dense_tracking_latency_mode_other_workgroup_counts(gpu_ctx, nwg):
    """Latency mode with 2 and 16 workgroups per stream (context option "trk_nwg"; the default of a one-stream call is 8): the per-sweep exchange as flagged 8-byte words
    and the two-workgroup hand-over of a near-tie's float sums (round 6) do not depend on the count -- every LM record of four streams equals the oracle's, pose within 1e-9."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import DenseTracker, FramePyramid
    ctx, stream = gpu_ctx
    cam = synth.CAM_DEFAULT
    B = 4
    sc = synth.Scene(2011)
    rng = np.random.default_rng(11)
    base = synth.trajectory(6)
    T_prev = [base[b % 6] for b in range(B)]
    T_cur = [synth.pose_mul(synth.pose(synth.so3_exp([rng.normal(0, 5e-4), np.deg2rad(rng.uniform(0.05, 0.5)), rng.normal(0, 5e-4)]),
                                       [rng.normal(0, 0.003), rng.normal(0, 0.002), -rng.uniform(0.02, 0.08)]), T_prev[b]) for b in range(B)]
    prev_f = [sc.render(cam, T_prev[b], seed=700 + b) for b in range(B)]
    cur_f = [sc.render(cam, T_cur[b], seed=800 + b) for b in range(B)]
    prev = FramePyramid(ctx, stream, cam, batch=B)
    cur = FramePyramid(ctx, stream, cam, batch=B)
    prev.upload(np.stack([f[0] for f in prev_f]), np.stack([f[1] for f in prev_f]))
    cur.upload(np.stack([f[0] for f in cur_f]), np.stack([f[1] for f in cur_f]))
    prev.preprocessing(); cur.preprocessing()
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    dtp = DenseTracker(ctx, prev)
    dtp.computeDensePointCloudCpu(I.reshape(12))
    dt = DenseTracker(ctx, cur)
    dt.ref_dense_points = dtp.ref_dense_points
    ctx.set_option("trk_nwg", nwg)
    try:
        T, passes = dt.denseTrackingCpu(prev.pyr, I.reshape(12), from_u8=True)
        recs = dt.lm_records()
    finally:
        ctx.set_option("trk_nwg", 0)
    assert (passes > 0).all()
    for b in range(B):
        clouds = [O.pointcloud_cpu(prev_f[b][1], prev.cams[l], l, I) for l in range(3)]
        pyr_p, pyr_c = O.build_pyramid(prev_f[b][0]), O.build_pyramid(cur_f[b][0])
        fl = [O.convert_sobel(p) for p in pyr_c]
        T_ref, passes_ref, rec_ref = O.dense_tracking_cpu(clouds, pyr_p, [f[0] for f in fl], [f[1] for f in fl], [f[2] for f in fl], cur.cams, I, want_rec=True)
        _check_cpu_sem_trajectory(recs[b], passes[b], rec_ref, passes_ref, f"{nwg} workgroups per stream, stream {b}")
        np.testing.assert_allclose(T[b], T_ref, rtol=0, atol=1e-9)


def test_dense_tracking_big_batch_continuation_launch(gpu_ctx):
    """A batch of more than one stream per CU runs the flat tracker kernel, and with "trk_split" = K its streams that are still iterating after K trials on the finest
    level PARK and are finished by a second launch with 1 / 2 / 4 / 8 workgroups per stream (dense.hip: the continuation launch).  300 streams over 8 scenes, K = 0 (off),
    1, 3, 5, 8: however many streams park and however many workgroups resume them, EVERY stream's accept / reject record equals the oracle's to the end, the poses agree with
    the oracle to 1e-9 and with the unsplit run to 1e-12, no stream reports a failed hand-over, and the replicas of a scene inside one batch are bit-equal."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import DenseTracker, FramePyramid
    ctx, stream = gpu_ctx
    cam = synth.CAM_DEFAULT
    NS, B = 8, 300
    sc = synth.Scene(2011)
    rng = np.random.default_rng(5)
    base = synth.trajectory(6)
    T_prev = [base[b % 6] for b in range(NS)]
    T_cur = [synth.pose_mul(synth.pose(synth.so3_exp([rng.normal(0, 5e-4), np.deg2rad(rng.uniform(0.05, 0.5)), rng.normal(0, 5e-4)]),
                                       [rng.normal(0, 0.003), rng.normal(0, 0.002), -rng.uniform(0.02, 0.08)]), T_prev[b]) for b in range(NS)]
    prev_f = [sc.render(cam, T_prev[b], seed=500 + b) for b in range(NS)]
    cur_f = [sc.render(cam, T_cur[b], seed=600 + b) for b in range(NS)]
    prev = FramePyramid(ctx, stream, cam, batch=B)
    cur = FramePyramid(ctx, stream, cam, batch=B)
    prev.upload(np.stack([prev_f[b % NS][0] for b in range(B)]), np.stack([prev_f[b % NS][1] for b in range(B)]))
    cur.upload(np.stack([cur_f[b % NS][0] for b in range(B)]), np.stack([cur_f[b % NS][1] for b in range(B)]))
    prev.preprocessing(); cur.preprocessing()
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    dtp = DenseTracker(ctx, prev)
    dtp.computeDensePointCloudCpu(I.reshape(12))
    dt = DenseTracker(ctx, cur)
    dt.ref_dense_points = dtp.ref_dense_points
    refs = {}
    for s_ in range(NS):
        clouds = [O.pointcloud_cpu(prev_f[s_][1], prev.cams[l], l, I) for l in range(3)]
        pyr_p, pyr_c = O.build_pyramid(prev_f[s_][0]), O.build_pyramid(cur_f[s_][0])
        fl = [O.convert_sobel(p) for p in pyr_c]
        refs[s_] = O.dense_tracking_cpu(clouds, pyr_p, [f[0] for f in fl], [f[1] for f in fl], [f[2] for f in fl], cur.cams, I, want_rec=True)
    lvl0_trials = [int(((r[2][:, 0] == 0) & (r[2][:, 1] < 2)).sum()) for r in refs.values()]
    out = {}
    try:
        for K in (0, 1, 3, 5, 8):
            ctx.set_option("trk_split", K)
            T, passes = dt.denseTrackingCpu(prev.pyr, I.reshape(12), from_u8=True)
            recs = dt.lm_records()
            assert (passes > 0).all(), f"K = {K}: dense_passes = -1 (a sibling workgroup of a parked stream never arrived)"
            assert ctx.get_stat("trk_exact_fallbacks") == 0
            for b in range(B):
                T_ref, passes_ref, rec_ref = refs[b % NS]
                if b < 2 * NS or b >= B - NS:
                    _check_cpu_sem_trajectory(recs[b], passes[b], rec_ref, passes_ref, f"K = {K}, stream {b}")
                np.testing.assert_allclose(T[b], T_ref, rtol=0, atol=1e-9)
                assert np.array_equal(T[b], T[b % NS]) and passes[b] == passes[b % NS] and recs[b].tobytes() == recs[b % NS].tobytes(), (K, b)      # replicas: the same bits
            out[K] = (T.copy(), passes.copy())
    finally:
        ctx.set_option("trk_split", 10)
    for K in (1, 3, 5, 8):
        assert np.array_equal(out[K][1], out[0][1])
        np.testing.assert_allclose(out[K][0], out[0][0], rtol=0, atol=1e-12)
    print(f"continuation launch: level-0 trials per scene {lvl0_trials}; K = 0 / 1 / 3 / 5 / 8: all records = the oracle's, poses of the split runs within "
          f"{max(float(np.abs(out[K][0] - out[0][0]).max()) for K in (1, 3, 5, 8)):.1e} of the unsplit run")


def test_dense_tracking_lm_trajectory_seq_chi2_to_the_end(gpu_ctx):
    """Option "trk_seq_chi2": the accept test runs on the reference's OWN sums -- `float chi2` accumulated sequentially in f32 over the samples in row-major
    order (dense_tracking.cpp:229-262,341-383).  Then no near-tie is left to summation order: EVERY stream's accept / reject record equals the oracle's to the
    end, chi2 of every trial BIT-equal (the per-sample terms are bit-equal, the sum is taken in the same order), final pose 1e-9."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import DenseTracker, FramePyramid
    ctx, stream = gpu_ctx
    cam = synth.CAM_DEFAULT
    B = 8
    sc = synth.Scene(2011)
    rng = np.random.default_rng(5)
    base = synth.trajectory(6)
    T_prev = [base[b % 6] for b in range(B)]
    T_cur = [synth.pose_mul(synth.pose(synth.so3_exp([rng.normal(0, 5e-4), np.deg2rad(rng.uniform(0.05, 0.5)), rng.normal(0, 5e-4)]),
                                       [rng.normal(0, 0.003), rng.normal(0, 0.002), -rng.uniform(0.02, 0.08)]), T_prev[b]) for b in range(B)]
    prev_f = [sc.render(cam, T_prev[b], seed=500 + b) for b in range(B)]
    cur_f = [sc.render(cam, T_cur[b], seed=600 + b) for b in range(B)]
    prev = FramePyramid(ctx, stream, cam, batch=B)
    cur = FramePyramid(ctx, stream, cam, batch=B)
    prev.upload(np.stack([f[0] for f in prev_f]), np.stack([f[1] for f in prev_f]))
    cur.upload(np.stack([f[0] for f in cur_f]), np.stack([f[1] for f in cur_f]))
    prev.preprocessing(); cur.preprocessing()
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    dtp = DenseTracker(ctx, prev)
    dtp.computeDensePointCloudCpu(I.reshape(12))
    dt = DenseTracker(ctx, cur)
    dt.ref_dense_points = dtp.ref_dense_points
    ctx.set_option("trk_seq_chi2", 1)
    try:
        for from_u8 in (True, False):
            T, passes = dt.denseTrackingCpu(prev.pyr, I.reshape(12), from_u8=from_u8)
            recs = dt.lm_records()
            for b in range(B):
                clouds = [O.pointcloud_cpu(prev_f[b][1], prev.cams[l], l, I) for l in range(3)]
                pyr_p, pyr_c = O.build_pyramid(prev_f[b][0]), O.build_pyramid(cur_f[b][0])
                fl = [O.convert_sobel(p) for p in pyr_c]
                T_ref, passes_ref, rec_ref = O.dense_tracking_cpu(clouds, pyr_p, [f[0] for f in fl], [f[1] for f in fl], [f[2] for f in fl], cur.cams, I,
                                                                  want_rec=True)
                ref = rec_ref.copy()
                dup = np.zeros(len(ref), bool)
                for k in range(1, len(ref)):
                    dup[k] = ref[k, 1] == 0 and ref[k - 1, 1] == 0 and ref[k, 0] == ref[k - 1, 0]      # the reference repeats a rejected trial before it stops
                ref = ref[~dup]
                rec = recs[b]
                assert len(rec) == len(ref) == passes[b], (b, len(rec), len(ref), passes[b])
                assert np.array_equal(rec["level"], ref[:, 0].astype(np.int32)) and np.array_equal(rec["accepted"], ref[:, 1].astype(np.int32)), f"stream {b}"
                assert np.array_equal(rec["chi2"].astype(np.float32), ref[:, 2].astype(np.float32)), f"stream {b}: chi2 of the trials"
                assert np.array_equal(rec["new_chi2"].astype(np.float32), ref[:, 3].astype(np.float32)), f"stream {b}: new_chi2 of the trials"
                np.testing.assert_allclose(T[b], T_ref, rtol=0, atol=1e-9)
    finally:
        ctx.set_option("trk_seq_chi2", 0)


def test_dense_full_resolution_variant(gpu_ctx, scene_frames):
    """GpuTracker::jacobianReduction / chi2 and computePointCloud (full-res f32 semantics of
    gpu/dense_tracking.cu): cloud bit-exact, sums within 1e-5 relative (f32 per-pixel math; the
    kernel accumulates in f64, the oracle too)."""
    import torch
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import GpuTracker
    ctx, stream = gpu_ctx
    cam, prev, cur, (img_p, disp_p, T_p), (img_c, disp_c, T_c) = _dense_setup(scene_frames, ctx, stream)
    l = 1
    w, h = cur.w[l], cur.h[l]
    c = cur.cams[l]
    Q = np.array([[1, 0, 0, -c.cx], [0, 1, 0, -c.cy], [0, 0, 0, c.f], [0, 0, 1.0 / c.b, 0]])
    TQ = Q.astype(np.float32)                      # T = identity
    gt = GpuTracker(ctx, stream, w, h)
    with torch.cuda.stream(stream):
        cloud = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    gt.computePointCloud(TQ.T.reshape(16), prev.disp[0], w, h, prev.stride[0], w, 1 << l, cloud)
    ctx.sync()
    cloud_ref = O.pointcloud_full(TQ.T.reshape(16), disp_p, w, h, 1 << l)
    assert np.array_equal(cloud.cpu().numpy(), cloud_ref)
    pyr_p, pyr_c = O.build_pyramid(img_p), O.build_pyramid(img_c)
    fp, _, _ = O.convert_sobel(pyr_p[l])
    fc, dx, dy = O.convert_sobel(pyr_c[l])
    T_true = synth.pose_mul(T_c, synth.pose_inv(T_p))
    T34 = T_true.astype(np.float32).T.reshape(12)   # GpuMatrix34: column-major 3x4
    gt.bindTexture(cur.f32[l][0], cur.dx[l][0], cur.dy[l][0], w, h, cur.stride[l])
    got = gt.jacobianReduction(prev.f32[l][0], cloud, T34, c.f, c.cx, c.cy, w, h, cur.stride[l], w)
    # oracle wants dense (stride = w) images
    ref = O.dense_pass_full(cloud_ref, fp, fc, dx, dy, np.float32(c.f), np.float32(c.cx), np.float32(c.cy), T34, True)
    assert got["n_valid"] == ref["n_valid"] and ref["n_valid"] > 1000
    np.testing.assert_allclose(got["H"], ref["H"], rtol=1e-5, atol=1e-5 * np.abs(ref["H"]).max())
    np.testing.assert_allclose(got["b"], ref["b"], rtol=1e-5, atol=1e-5 * np.abs(ref["b"]).max())
    chi2 = gt.chi2(prev.f32[l][0], cloud, T34, c.f, c.cx, c.cy, w, h, cur.stride[l], w)
    np.testing.assert_allclose(chi2, ref["chi2"], rtol=1e-6)
    # GpuTracker::residualImage (gpu/dense_tracking.cu:495-541): per-pixel f32 arithmetic only => bit-exact
    with torch.cuda.stream(stream):
        rimg = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    gt.residualImage(prev.f32[l][0], cloud, T34, c.f, c.cx, c.cy, w, h, cur.stride[l], w, rimg)
    rimg_ref = O.residual_image_full(cloud_ref, fp, fc, np.float32(c.f), np.float32(c.cx), np.float32(c.cy), T34)
    got_r = rimg.cpu().numpy()
    assert np.array_equal(got_r, rimg_ref)
    grey = (rimg_ref[..., 0] == rimg_ref[..., 1]) & (rimg_ref[..., 3] == 1)
    assert grey.sum() == ref["n_valid"] and (~grey).any()      # the rest is red (out of frame) or green (no depth)


def test_dense_residual_images(gpu_ctx, scene_frames):
    """DenseTracker::residual_img[level] (dense_tracking.cpp:52-54,279-329), the GUI side output of denseTrackingCpu.
    (1) one H,b pass at a given pose: bit-exact vs the oracle's pass, f32 and fused-u8 source alike;
    (2) after the device-resident tracker: the image at the pose the tracker recorded for the last H,b pass of each
        level is bit-exact vs the oracle at that pose, and close to what the oracle's own tracking run leaves behind
        (the two LM trajectories agree to 1e-4)."""
    import oracle as O
    from scavislam_amd.frontend import DenseTracker
    ctx, stream = gpu_ctx
    cam, prev, cur, (img_p, disp_p, T_p), (img_c, disp_c, T_c) = _dense_setup(scene_frames, ctx, stream)
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    dtp = DenseTracker(ctx, prev)
    dtp.computeDensePointCloudCpu(I.reshape(12))
    dt = DenseTracker(ctx, cur)
    dt.ref_dense_points = dtp.ref_dense_points
    for r in dt.residual_img:                                  # constructor state: (0,0,0,1)
        assert np.array_equal(r.cpu().numpy()[..., :3], np.zeros(r.shape[:-1] + (3,), np.float32)) and bool((r[..., 3] == 1).all())
    clouds = [O.pointcloud_cpu(disp_p, prev.cams[l], l, I) for l in range(3)]
    pyr_p, pyr_c = O.build_pyramid(img_p), O.build_pyramid(img_c)
    fl = [O.convert_sobel(p) for p in pyr_c]
    # (1) fixed pose
    dt.d_T_jac.copy_(__import__("torch").as_tensor(np.tile(I.reshape(12), (cur.batch, 3, 1))))
    for from_u8 in (False, True):
        got = dt.computeResidualImages(prev.pyr, from_u8=from_u8)
        for l in range(3):
            _, ref = O.dense_pass_cpu(clouds[l], pyr_p[l], fl[l][0], fl[l][1], fl[l][2], cur.cams[l], I, True, want_rimg=True)
            assert np.array_equal(got[l][0], ref), (l, from_u8)
            assert (ref[..., 0] == 1).any() or (ref[..., 1] == 1).any()
    # (2) after tracking
    T_gpu, _ = dt.denseTrackingCpu(prev.pyr, I.reshape(12))
    got = dt.computeResidualImages(prev.pyr)
    Tj = dt.d_T_jac.cpu().numpy()[0]
    T_ref, _, rimg_ref = O.dense_tracking_cpu(clouds, pyr_p, [f[0] for f in fl], [f[1] for f in fl], [f[2] for f in fl],
                                              cur.cams, I, want_rimg=True)
    for l in range(3):
        _, ref = O.dense_pass_cpu(clouds[l], pyr_p[l], fl[l][0], fl[l][1], fl[l][2], cur.cams[l], Tj[l].reshape(3, 4), True, want_rimg=True)
        assert np.array_equal(got[l][0], ref), l
        assert np.abs(got[l][0] - rimg_ref[l]).mean() < 2e-3, l
    # the level-0 image belongs to a pose at most one LM step away from the result
    assert np.abs(Tj[0].reshape(3, 4) - T_gpu[0]).max() < 1e-2


def test_dense_tracking_rgbd_config_with_invalid_depth(gpu_ctx):
    """RGB-D intrinsics (data/rgbd_example.cfg), disparity from u16 depth with the depthToDisp
    semantics (frame_grabber-impl.cpp:136-151: depth * 1/5000 m -> f / (depth * b), 0 where the sensor has no depth),
    10 % invalid pixels in blobs.  Cloud bit-exact (w = -1 entries included), pass sums and tracked pose as usual."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import DenseTracker
    ctx, stream = gpu_ctx
    cam = synth.CAM_RGBD
    sc = synth.Scene(2013)
    traj = synth.trajectory(3, step=0.03, yaw_deg=0.15)
    rng = np.random.default_rng(2013)
    frames = []
    for i in (0, 1):
        img, disp = sc.render(cam, traj[i], seed=40 + i)
        depth_u16 = np.where(disp > 0, np.rint(cam["f"] * cam["b"] / np.maximum(disp, 1e-6) * 5000.0), 0).astype(np.uint16)
        for _ in range(60):                                     # blobs of missing depth, about 10 % of the image
            x0, y0 = rng.integers(0, cam["w"] - 40), rng.integers(0, cam["h"] - 40)
            depth_u16[y0:y0 + rng.integers(8, 40), x0:x0 + rng.integers(8, 40)] = 0
        depth = depth_u16.astype(np.float32) * np.float32(1.0 / 5000.0)
        with np.errstate(divide="ignore"):
            d = np.where(depth_u16 > 0, np.float32(cam["f"]) / (depth * np.float32(cam["b"])) , np.float32(0)).astype(np.float32)
        frames.append((img, d))
    assert 0.05 < (frames[0][1] == 0).mean() < 0.25
    prev = _frame(ctx, stream, cam, [frames[0][0]], [frames[0][1]])
    cur = _frame(ctx, stream, cam, [frames[1][0]], [frames[1][1]])
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    dtp = DenseTracker(ctx, prev)
    dtp.computeDensePointCloudCpu(I.reshape(12))
    ctx.sync()
    clouds = [O.pointcloud_cpu(frames[0][1], prev.cams[l], l, I) for l in range(3)]
    for l in range(3):
        got = dtp.ref_dense_points[l][0].cpu().numpy()
        assert np.array_equal(got, clouds[l]) and (got[..., 3] == -1).any()
    dt = DenseTracker(ctx, cur)
    dt.ref_dense_points = dtp.ref_dense_points
    pyr_p, pyr_c = O.build_pyramid(frames[0][0]), O.build_pyramid(frames[1][0])
    fl = [O.convert_sobel(p) for p in pyr_c]
    for l in range(3):
        ref = O.dense_pass_cpu(clouds[l], pyr_p[l], fl[l][0], fl[l][1], fl[l][2], cur.cams[l], I, True)
        got = dt.pass_sums(l, prev.pyr, I.reshape(12), True)[0]
        assert got["n_valid"] == ref["n_valid"]
        np.testing.assert_allclose(got["H"], ref["H"], rtol=0, atol=1e-9 * np.abs(ref["H"]).max())
    T_gpu, passes = dt.denseTrackingCpu(prev.pyr, I.reshape(12), from_u8=True)
    T_ref, _ = O.dense_tracking_cpu(clouds, pyr_p, [f[0] for f in fl], [f[1] for f in fl], [f[2] for f in fl], cur.cams, I)
    T_true = synth.pose_mul(traj[1], synth.pose_inv(traj[0]))
    np.testing.assert_allclose(T_gpu[0], T_ref, rtol=0, atol=1e-4)
    assert np.abs(T_gpu[0][:, :3] - T_true[:, :3]).max() < 2e-3      # rotation recovered (the 3 cm translation is below what this scene constrains)


def test_dense_tracking_multi_workgroup_variant(gpu_ctx, scene_frames, monkeypatch):
    """Latency-mode tracker (<= 32 streams: 4 workgroups share every sweep of a stream, partial sums added in a fixed order
    behind a device-scope barrier) against the one-workgroup-per-stream kernel on a small batch of DIFFERENT streams:
    same passes, poses equal to 1e-9 (only the order of the f64 partial sums differs), per-level Jacobian poses too."""
    import torch
    from scavislam_amd import synth
    from scavislam_amd.frontend import DenseTracker, FramePyramid
    ctx, stream = gpu_ctx
    cam = scene_frames["cam"]
    sc = synth.Scene(2011)
    traj = synth.trajectory(8, step=0.04, yaw_deg=0.3)
    B = 5
    prev_f = [sc.render(cam, traj[i], seed=50 + i) for i in range(B)]
    cur_f = [sc.render(cam, traj[i + 1], seed=60 + i) for i in range(B)]
    prev = FramePyramid(ctx, stream, cam, batch=B)
    cur = FramePyramid(ctx, stream, cam, batch=B)
    prev.upload(np.stack([f[0] for f in prev_f]), np.stack([f[1] for f in prev_f]))
    cur.upload(np.stack([f[0] for f in cur_f]), np.stack([f[1] for f in cur_f]))
    prev.preprocessing(); cur.preprocessing()
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    dtp = DenseTracker(ctx, prev)
    dtp.computeDensePointCloudCpu(I.reshape(12))
    out = {}
    for nwg in ("1", "4"):
        ctx.set_option("trk_nwg", int(nwg))
        dt = DenseTracker(ctx, cur)
        dt.ref_dense_points = dtp.ref_dense_points
        T, passes = dt.denseTrackingCpu(prev.pyr, I.reshape(12), from_u8=True)
        out[nwg] = (T.copy(), passes.copy(), dt.d_T_jac.cpu().numpy().copy())
    # the 128-register build of the one-workgroup kernel (used when there are more streams than CUs): same arithmetic
    ctx.set_option("trk_nwg", 1)
    ctx.set_option("trk_regs", 2)
    dt = DenseTracker(ctx, cur)
    dt.ref_dense_points = dtp.ref_dense_points
    T2, passes2 = dt.denseTrackingCpu(prev.pyr, I.reshape(12), from_u8=True)
    assert np.array_equal(T2, out["1"][0]) and np.array_equal(passes2, out["1"][1])
    ctx.set_option("trk_regs", 0)
    ctx.set_option("trk_nwg", 0)
    (T1, p1, j1), (T4, p4, j4) = out["1"], out["4"]
    assert np.array_equal(p1, p4) and len(set(p1.tolist())) >= 1
    np.testing.assert_allclose(T4, T1, rtol=0, atol=1e-9)
    np.testing.assert_allclose(j4, j1, rtol=0, atol=1e-9)
    for b in range(B):                                       # every stream moved towards its own true motion
        T_true = synth.pose_mul(traj[b + 1], synth.pose_inv(traj[b]))
        assert np.abs(T4[b] - T_true).max() < 0.5 * np.abs(I - T_true).max()


def test_fastgrid_large_frame_two_sweep_compaction(gpu_ctx):
    """A 2560x1440 frame: level-0 cells are 853x480 pixels and hold ~20 000 candidates each (rounds 1-5: beyond the list sort's cap and the one-sweep
    compaction's LDS budget -- the two-sweep fallback; round 6: the cell's bitmap, 54 KB of LDS, has no cap); lists stay bit-exact (order included)."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import FastGrid
    ctx, stream = gpu_ctx
    w, h = 2560, 1440
    cam = dict(synth.CAM_DEFAULT, w=w, h=h, cx=w / 2, cy=h / 2)
    imgs = [synth.noise_image(w, h, 900 + i) for i in range(2)]
    fr = _frame(ctx, stream, cam, [imgs[0]], with_float=False)
    fg = FastGrid(ctx, fr, corner_cap=60000)
    grids = [O.fastgrid_for_level(fr.w[l], fr.h[l], l) for l in range(3)]
    for i in range(2):
        fr.upload(imgs[i][None])
        fr.preprocessing()
        fg.detectAdaptively(trials=6)
        pyr = O.build_pyramid(imgs[i])
        for l in range(3):
            xy_ref, cc_ref, et_ref = O.fastgrid_detect_adaptively(grids[l], pyr[l], 6)
            xy, cc, et, ts = fg.corners(0, l)
            assert np.array_equal(cc, cc_ref) and np.array_equal(et, et_ref), (i, l)
            assert np.array_equal(xy, xy_ref), (i, l)


def test_matcher_two_keyframes_two_streams(gpu_ctx, scene_frames):
    """Candidate points anchored in two different keyframes (keyframe_map lookup by anchor id, matcher.cpp:326-345), the
    active keyframe being the second one, for a batch of two current frames with different pose guesses: every record of
    both streams bit-exact against the oracle."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.frontend import FastGrid, FramePyramid, GuidedMatcher
    ctx, stream = gpu_ctx
    cam = scene_frames["cam"]
    sc = synth.Scene(2011)
    traj = synth.trajectory(8)
    k0, k1, c0, c1 = 0, 2, 5, 6
    img_k0, disp_k0 = sc.render(cam, traj[k0], seed=k0)
    img_k1, disp_k1 = sc.render(cam, traj[k1], seed=k1)
    curs = [sc.render(cam, traj[i], seed=i) for i in (c0, c1)]
    fr = _frame(ctx, stream, cam, [c[0] for c in curs], [c[1] for c in curs])
    fg = FastGrid(ctx, fr)
    for _ in range(2):
        fg.detectAdaptively(trials=6)
    kfs = []
    for img in (img_k0, img_k1):
        kf = FramePyramid(ctx, stream, cam, batch=1, with_float=False)
        kf.upload(img[None])
        kf.preprocessing()
        kfs.append(kf)
    rng = np.random.default_rng(21)
    pts = np.concatenate([synth.candidate_points(rng, cam, disp_k0, traj[k0], (200, 100, 40), kf_index=0),
                          synth.candidate_points(rng, cam, disp_k1, traj[k1], (200, 100, 40), kf_index=1)])
    pts["point_id"] = np.arange(len(pts))
    rng.shuffle(pts)
    T_act = traj[k1]                                                     # active keyframe = keyframe 1
    T_guess = []
    for i, ci in enumerate((c0, c1)):
        T = synth.pose_mul(traj[ci], synth.pose_inv(T_act))
        T[:, 3] += np.array([0.003, -0.002, 0.004]) * (i + 1)
        T_guess.append(T)
    gm = GuidedMatcher(ctx, fr, fg)
    res = gm.match([(kfs[0].pyr, 0, traj[k0].reshape(12)), (kfs[1].pyr, 0, traj[k1].reshape(12))],
                   np.stack([T.reshape(12) for T in T_guess]), T_act.reshape(12), pts)
    pyr_k = [O.build_pyramid(img_k0), O.build_pyramid(img_k1)]
    n_ok = 0
    for s in range(2):
        pyr_c = O.build_pyramid(curs[s][0])
        trees = []
        for l in range(3):
            xy, cc, et, ts = fg.corners(s, l)
            trees.append(O.quadtree_from_corners(xy, cc, pyr_c[l].shape[1], pyr_c[l].shape[0]))
        ref = O.match(pyr_k, [traj[k0].reshape(12), traj[k1].reshape(12)], T_guess[s], T_act, pyr_c, curs[s][1], trees, fr.cams, pts)
        for k in ("status", "znssd"):
            bad = np.nonzero(res[s][k] != ref[k])[0]
            assert bad.size == 0, (s, k, bad[:8], res[s][k][bad[:8]], ref[k][bad[:8]], pts["kf_index"][bad[:8]], pts["anchor_level"][bad[:8]],
                                   res[s]["u"][bad[:8]], ref["u"][bad[:8]], res[s]["v"][bad[:8]], ref["v"][bad[:8]])
        found = (ref["status"] == 0) | (ref["status"] == 6)
        assert np.array_equal(res[s]["u"][found], ref["u"][found]) and np.array_equal(res[s]["v"][found], ref["v"][found])
        ok = ref["status"] == 0
        assert np.array_equal(res[s]["obs"][ok], ref["obs"][ok])
        np.testing.assert_allclose(res[s]["xyz_actkey"][found], ref["xyz_actkey"][found], rtol=1e-12, atol=1e-12)
        for kfi in (0, 1):
            assert (ok & (pts["kf_index"] == kfi)).sum() > 20, (s, kfi)
        n_ok += ok.sum()
    assert n_ok > 200
