"""GPU parity of the block-matching disparity stage (StereoFrontend::calcDisparityCpu, stereo_frontend.cpp:620-653)
against the oracle restatement of cv::StereoBM -- an integer pipeline, so every stage must be bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(gpu_ctx, cam, lefts, rights, prm):
    from scavislam_amd.frontend import FramePyramid, StereoMatcher
    ctx, stream = gpu_ctx
    fr = FramePyramid(ctx, stream, cam, batch=len(lefts), with_float=False)
    fr.upload(np.stack(lefts))
    sm = StereoMatcher(ctx, fr, prm)
    sm.upload_right(np.stack(rights))
    sm.calcDisparityCpu()
    out = [sm.disparity_host(b) for b in range(len(lefts))]
    sm.close()
    return out


def _prm(validate=True, speckle=True):
    from scavislam_amd.ctypes_types import StereoParams
    p = StereoParams.reference()
    if not validate:
        p.disp12_max_diff = -1
    if not speckle:
        p.speckle_window = 0
    return p


@pytest.fixture(scope="module")
def stereo_pairs():
    from scavislam_amd import synth
    sc = synth.Scene(2011)
    traj = synth.trajectory(4)
    return [synth.render_stereo(sc, synth.CAM_DEFAULT, traj[i], seed=10 + i) for i in (0, 3)]


@pytest.mark.parametrize("validate,speckle", [(False, False), (True, False), (True, True)])
def test_stereo_bm_stages_640x480(gpu_ctx, stereo_pairs, validate, speckle):
    """raw block matching, + left-right check, + speckle filter: each prefix of the pipeline equals the oracle."""
    import oracle as O
    from scavislam_amd import synth
    prm = _prm(validate, speckle)
    got = _run(gpu_ctx, synth.CAM_DEFAULT, [p[0] for p in stereo_pairs], [p[1] for p in stereo_pairs], prm)
    for (l, r, truth), g in zip(stereo_pairs, got):
        ref = O.stereo_bm(l, r, prm)
        assert g.dtype == np.float32 and g.shape == ref.shape
        assert np.array_equal(g, ref), f"{(g != ref).sum()} px differ (validate={validate}, speckle={speckle})"
        if validate and speckle:
            ok = g >= 0
            assert ok.mean() > 0.5 and np.median(np.abs(g - truth)[ok]) < 0.25      # it is a usable disparity map


def test_stereo_bm_odd_size_and_borders(gpu_ctx):
    """odd height (last prefiltered row = cap quirk), width not a multiple of 64, disparities up to the search range."""
    import oracle as O
    from scavislam_amd import synth
    sc = synth.Scene(7)
    cam = dict(synth.CAM_DEFAULT)
    cam.update(w=203, h=97, cx=101.0, cy=48.0, f=160.0, b=0.3)
    l, r, _ = synth.render_stereo(sc, cam, synth.trajectory(2)[1], seed=3)
    got = _run(gpu_ctx, cam, [l], [r], _prm())[0]
    assert np.array_equal(got, O.stereo_bm(l, r))
    assert (got[:, :31] == -1).all()


def test_stereo_bm_degenerate_images(gpu_ctx):
    """flat images (texture test rejects everything), pure noise (uniqueness / LR check / speckles remove most),
    identical left and right (disparity 0 nearly everywhere it survives)."""
    import oracle as O
    from scavislam_amd import synth
    cam = dict(synth.CAM_DEFAULT)
    cam.update(w=160, h=120, cx=80.0, cy=60.0)
    rng = np.random.default_rng(5)
    flat = np.full((120, 160), 90, np.uint8)
    noise_l = rng.integers(0, 256, (120, 160)).astype(np.uint8)
    noise_r = rng.integers(0, 256, (120, 160)).astype(np.uint8)
    tex = synth.noise_image(160, 120, seed=9)
    lefts, rights = [flat, noise_l, tex], [flat, noise_r, tex]
    got = _run(gpu_ctx, cam, lefts, rights, _prm())
    for l, r, g in zip(lefts, rights, got):
        assert np.array_equal(g, O.stereo_bm(l, r))
    assert (got[0] == -1).all()
    keep = got[2] >= 0
    assert keep.any() and (got[2][keep] == 0).mean() > 0.9      # ties inside flat blobs go to the larger disparity


@pytest.mark.parametrize("cap", [5, 41, 42, 63])
def test_stereo_bm_prefilter_caps(gpu_ctx, cap):
    """preFilterCap decides which winner search runs: cap <= 41 bounds every SAD below 4096 (packed 16-bit keys), above it the 32-bit keys.
    A textured pair, a noise pair and a two-pixel stripe pattern against its one-pixel shift (every tap saturates: SADs of 49 * 2 cap =
    6174 at cap 63 on one disparity parity, 0 on the other -- long runs of equal minima, first one wins): bit-exact for all caps."""
    import oracle as O
    from scavislam_amd import synth
    cam = dict(synth.CAM_DEFAULT, w=320, h=240, cx=160.0, cy=120.0)
    sc = synth.Scene(7)
    l, r, _ = synth.render_stereo(sc, cam, synth.trajectory(2)[1], seed=3)
    rng = np.random.default_rng(cap)
    nl = rng.integers(0, 256, (240, 320)).astype(np.uint8)
    nr_ = np.roll(nl, -5, axis=1)
    stripes = np.tile(np.repeat(np.array([0, 255], np.uint8), 2), (240, 80))
    sl, sr = stripes, np.roll(stripes, -1, axis=1)
    prm = _prm()
    prm.prefilter_cap = cap
    got = _run(gpu_ctx, cam, [l, nl, sl], [r, nr_, sr], prm)
    for g, (a, b) in zip(got, [(l, r), (nl, nr_), (sl, sr)]):
        assert np.array_equal(g, O.stereo_bm(a, b, prm))
    prm2 = _prm(validate=False, speckle=False)      # the raw search output of the stripe pair
    prm2.prefilter_cap = cap
    prm2.uniqueness_ratio = 0
    prm2.texture_threshold = 0
    g = _run(gpu_ctx, cam, [sl], [sr], prm2)[0]
    ref = O.stereo_bm(sl, sr, prm2)
    assert np.array_equal(g, ref) and (ref >= 0).mean() > 0.5


def test_stereo_unsupported_parameters(gpu_ctx):
    import ctypes as C
    from scavislam_amd.ctypes_types import StereoParams
    ctx, _ = gpu_ctx
    p = StereoParams.reference()
    p.sad_window = 9
    h = C.c_void_p()
    rc = ctx.lib.svs_stereo_create(ctx.h, 640, 480, 1, C.byref(p), C.byref(h))
    assert rc == 5 and not h.value      # SVS_ERR_UNSUPPORTED, nothing allocated


def test_speckle_filter_strips_many_frames_and_widths(gpu_ctx):
    """The speckle filter works on strips of rows held in LDS and finishes the components that cross strip boundaries in three small
    kernels: four 640x480 pairs in one batch (several calls: the first version re-walked from the BIG sentinel after a lost CAS and only
    one of these pairs hit it), a width that is not a multiple of 4 with several strips, and images made of specks (noise pairs: nearly every
    component is small, many straddle a boundary)."""
    import oracle as O
    from scavislam_amd import synth
    sc = synth.Scene(2011)
    traj = synth.trajectory(5)
    pairs = [synth.render_stereo(sc, synth.CAM_DEFAULT, traj[i], seed=10 + i) for i in range(4)]
    for _ in range(2):
        got = _run(gpu_ctx, synth.CAM_DEFAULT, [p[0] for p in pairs], [p[1] for p in pairs], _prm())
        for (l, r, _t), g in zip(pairs, got):
            assert np.array_equal(g, O.stereo_bm(l, r))
    cam = dict(synth.CAM_DEFAULT)
    cam.update(w=322, h=250, cx=161.0, cy=125.0, f=250.0, b=0.3)
    l, r, _ = synth.render_stereo(sc, cam, traj[2], seed=5)
    rng = np.random.default_rng(11)
    nl = np.clip(l.astype(np.int32) + rng.integers(-40, 41, l.shape), 0, 255).astype(np.uint8)      # heavy noise: specks everywhere
    nr_ = np.clip(r.astype(np.int32) + rng.integers(-40, 41, r.shape), 0, 255).astype(np.uint8)
    got = _run(gpu_ctx, cam, [l, nl], [r, nr_], _prm())
    for (a, b_), g in zip([(l, r), (nl, nr_)], got):
        ref = O.stereo_bm(a, b_)
        assert np.array_equal(g, ref), f"{(g != ref).sum()} px differ"
    cam2 = dict(synth.CAM_DEFAULT)
    rng = np.random.default_rng(12)
    l2, r2, _ = pairs[1]
    nl2 = np.clip(l2.astype(np.int32) + rng.integers(-30, 31, l2.shape), 0, 255).astype(np.uint8)
    nr2 = np.clip(r2.astype(np.int32) + rng.integers(-30, 31, r2.shape), 0, 255).astype(np.uint8)
    got = _run(gpu_ctx, cam2, [nl2], [nr2], _prm())[0]
    ref = O.stereo_bm(nl2, nr2)
    assert np.array_equal(got, ref), f"{(got != ref).sum()} px differ"
    assert 0.02 < (ref >= 0).mean() < 0.98
