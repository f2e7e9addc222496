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
import json
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
# pose within 1e-6 on all but ONE frame and within 2e-6 on that one (what is left: calcFastMotionOnly's own last steps are decided by the rounding of ITS chi2 sums -- f64, ~1e-14 -- and move
# the pose by ~1e-8 ... 1e-7; a keyframe dropped at such a frame keeps the offset in its world pose, and every later frame that matches points anchored in other
# keyframes sees it).  Measured (profiles/r4_gpu_tests.txt, trk_seq_chi2 = 1): 200 / 199 of 200 frames within 1e-6, worst 1.1e-6.
def _check_strict(st, n_frames, what):
    assert st["other_points"] == 0 and st["frames_1e6"] >= n_frames - 1 and st["max_dT"] <= 2e-6, (what, st)


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


def test_batched_call_runs_eight_sequences_of_the_reference_loop(gpu_ctx):
    """svs_frontend_process_frames -- the entry point bench.py times -- over SEQUENCES: eight camera streams in ONE batch, each living through its own part of the
    there-and-back sequence (phase offsets of 16 frames, 40 frames each: different scenes, different keyframe drops at different steps), previous frames, keyframe
    pyramids, FAST thresholds and clouds of all eight persisting inside the one front-end object.

    How it is held against the reference: every stream's sub-sequence first runs through the reference's OWN loop with the one-call binding in place
    (libsvs_hipbranch_seq_onecall.so, one stream per call) and is compared with the reference's CPU build run here on the same frames (strict bars: decisions, ids,
    thresholds identical, pose 1e-6 on every frame, accepted points identical up to the few on the reprojection gate -- see the assertion).  What that loop handed to the library frame by frame -- kept keyframes, matchAndTrack's lists, poses -- is then
    replayed for all eight streams together through the batched call, and every stream must return the BITS the loop consumed: refined pose, every match record, the
    pass count, the persistent thresholds.  Since the loop's next inputs are a function of those outputs, the batch IS eight reference loops."""
    import torch
    if not (_have("libsvs_hipbranch_seq_onecall.so") and _have("libsvs_ref_seq.so")):
        pytest.skip("oracle/_ref/libsvs_hipbranch_seq_onecall.so / libsvs_ref_seq.so not present")
    import oracle as O
    from scavislam_amd import capi
    from scavislam_amd.frontend import StereoFrontend
    ctx, stream = gpu_ctx
    B, M, STEP = 8, 40, 16
    cam = S.cam_of("default")
    fl = list(S.frames("default", STEP * (B - 1) + M))
    recs, outs = [], []
    worst_other = 0
    for b in range(B):
        sub = fl[b * STEP:b * STEP + M]
        seq = O.RefSequence(_cams("default"), hip_branch=True, one_call=True)
        hip, rec = [], []
        for img, disp in sub:
            r = seq.step(img, disp)
            r["crc"] = S.frame_crc(img, disp)
            r["lines"] = [ln[np.lexsort((ln[:, 4], ln[:, 3], ln[:, 2], ln[:, 1], -ln[:, 0]))] if len(ln) else ln for ln in r["lines"]]
            if r["dropped"]:
                r["new_ids"], r["new_val"] = seq.new_points(r["actkey_id"])
            hip.append(r)
            rec.append(seq.onecall_record())
        seq.close()
        ref_seq = O.RefSequence(_cams("default"))
        ref = S.run(ref_seq, "default", M, frame_list=sub)
        ref_seq.close()
        st = S.compare(hip, ref, f"stream {b} (frames {b * STEP}..{b * STEP + M - 1}): one-call binding in the reference's loop vs the reference's CPU build")
        # pose within 1e-6 on EVERY frame; accepted points: identical but for the handful that sit on the reprojection gate when calcFastMotionOnly ends one step apart.
        # Its accept test is `chi2 - new_chi2 > 0` on two sums accumulated sequentially in one double IN LIST ORDER (pose_optimizer.h:236-269), and the list order follows
        # the heap (global.h:47-54 hashes shared pointers by address; stereo_frontend.cpp:337-342 walks an unordered_set of them): the reference's own CPU build, run on
        # these same eight sub-sequences with nothing changed but the addresses its objects land at, differs from itself by up to YARD points in a stream
        # (tests/golden/yardstick_heap_order.json, made by tools/yardstick_heap_order.py: 10 of 128 runs with other points, 2 runs failing even compare()'s hard part).
        # The HIP branch is held to that spread, not to more: at most the reference's own (measured here: 2 on one stream, 0 on the others).
        yard = json.load(open(os.path.join(GOLDEN, "yardstick_heap_order.json")))
        YARD = int(yard["max_other_points"])
        assert YARD >= 1 and yard["runs_with_other_points"] >= 1
        assert st["frames_1e6"] == M and st["max_dT"] <= 1e-6 and st["other_points"] <= YARD and st["frames_with_other_points"] <= 1, \
            (f"stream {b}: more accepted points differ from the reference's CPU build than the reference differs from ITSELF on another heap layout ({YARD})", st)
        worst_other = max(worst_other, st["other_points"])
        assert rec[0] is None and all(x is not None for x in rec[1:])
        recs.append(rec); outs.append(hip)
    # the replay: all eight through the batched entry point
    dev = torch.device("cuda", 0)

    def dev_frames(k):
        with torch.cuda.stream(stream):
            left = torch.as_tensor(np.stack([fl[b * STEP + k][0] for b in range(B)])).to(dev)
            disp = torch.as_tensor(np.stack([fl[b * STEP + k][1] for b in range(B)]).astype(np.float32)).to(dev)
        stream.synchronize()
        return dict(left=left, disp=disp)

    fe = StereoFrontend(ctx, cam, max_points=16384, max_keyframes=32, params=capi.FrontendParams.reference(), n_streams=B)
    fe.processFirstFrames(**dev_frames(0))
    n_rec = n_kept = n_recloud = 0
    for k in range(1, M):
        for b in range(B):
            rc = recs[b][k]
            if rc["kept_slot"] >= 0:
                fe.keepKeyframe(rc["kept_slot"], rc["kept_pose"], stream=b)
                n_kept += 1
            fe.setCandidateLists(rc["pts"], rc["group_end"], stream=b)
        fr = dev_frames(k)
        fe.processFrames(np.stack([recs[b][k]["T_guess"] for b in range(B)]), np.stack([recs[b][k]["T_act"] for b in range(B)]), **fr)
        for b in range(B):
            rc = recs[b][k]
            res, m, g = fe.results(b)
            assert bytes(res.T_cur_from_actkey) == bytes(rc["res"].T_cur_from_actkey), f"frame {k}, stream {b}: refined pose"
            assert res.dense_passes == rc["res"].dense_passes and res.n_matched == rc["res"].n_matched and res.tracking_ok == rc["res"].tracking_ok, (k, b)
            assert m.tobytes() == rc["matches"].tobytes(), f"frame {k}, stream {b}: match records"
            n_rec += len(m)
        # a stream whose keyframe logic changed the pose behind the call (new keyframe: identity; switch: the pose relative to the other keyframe) had its dense cloud made
        # again at that pose (stereo_frontend.cpp:298-302); for the others the same call re-makes the cloud they already have
        if any(recs[b][k]["recloud"] for b in range(B)):
            T_all = np.stack([recs[b][k]["recloud_pose"] if recs[b][k]["recloud"] else np.array(recs[b][k]["res"].T_cur_from_actkey) for b in range(B)])
            fe.recomputeCloud(T_all)
            n_recloud += sum(recs[b][k]["recloud"] for b in range(B))
        del fr
    # the persistent FAST thresholds after the last frame = what the loops recorded (cells of the three levels)
    for b in range(B):
        want = outs[b][M - 1]["fast_thr"]
        got = []
        for l in range(3):
            ncell = [9, 9, 4][l]
            got.append(fe.corners(b, l)[3][:ncell])
        assert np.array_equal(np.concatenate(got), want), (b, np.concatenate(got), want)
    fe.close()
    drops = [sum(r["dropped"] for r in o) for o in outs]
    print(f"batched one-call, {B} streams x {M} frames at phase offsets of {STEP}: every stream = the reference's loop (vs the CPU build: decisions / ids / thresholds identical, "
          f"pose 1e-6 on every frame, at most {worst_other} other accepted points in a stream); batched replay bit-equal: "
          f"{n_rec} match records, {n_kept} keyframes kept, {n_recloud} clouds re-made after keyframe decisions, keyframe drops per stream {drops}")
