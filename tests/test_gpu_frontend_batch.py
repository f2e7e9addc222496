"""The batched / asynchronous forms of the one-call front end (svs_frontend_create_batch, *_frames, submit / wait / prefetch) and its configuration
switches (use_n_levels_in_frontent = 2, fewer than 20 matches, unkept keyframe slots).  The reference point is the blocking one-stream call
svs_frontend_process_frame, which tests/test_gpu_ref_frame.py holds against the reference's own processFrame: every other way of driving the same
kernels must return the same bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

I34 = np.hstack([np.eye(3), np.zeros((3, 1))])


def _streams(n, block_matching=False):
    """n camera streams with their own scene position, keyframe, previous and current frame, candidate list and motion guess"""
    from scavislam_amd import synth
    cam = synth.CAM_NEWCOLLEGE
    sc = synth.Scene(2011)
    traj = synth.trajectory(8 + n)
    out = []
    for b in range(n):
        kf_i, prev_i, cur_i = b, b + 3, b + 4
        fr = {name: synth.render_stereo(sc, cam, traj[i], seed=10 * b + s) for name, i, s in (("kf", kf_i, 1), ("prev", prev_i, 2), ("cur", cur_i, 3))}
        rng = np.random.default_rng(7 + b)
        pts = synth.candidate_points(rng, cam, np.maximum(fr["kf"][2], 0), traj[kf_i], (400 - 60 * b, 200, 60 + 10 * b))
        T_guess = synth.pose_mul(synth.pose(synth.so3_exp(np.array([0.001, -0.002, 0.0005]) * (b + 1)), np.array([0.004, 0.0, -0.004])),
                                 synth.pose_mul(traj[cur_i], synth.pose_inv(traj[prev_i])))
        out.append(dict(fr=fr, pts=pts, n_new=150 + 20 * b, T_guess=T_guess, T_kf=traj[kf_i], T_act=traj[prev_i]))
    return cam, out


def _single(ctx, cam, s, params, prefetch=False, split=False):
    from scavislam_amd.frontend import StereoFrontend
    bm = bool(params.use_block_matching)
    kw = (lambda n: dict(right=s["fr"][n][1])) if bm else (lambda n: dict(disp=s["fr"][n][2]))
    fe = StereoFrontend(ctx, cam, max_points=1024, max_keyframes=3, params=params)
    fe.processFirstFrame(s["fr"]["kf"][0], **kw("kf"))
    fe.keepKeyframe(0, s["T_kf"])
    if prefetch:
        fe.prefetchFrame(s["fr"]["prev"][0], **kw("prev"))
        fe.processFirstFrame()
    else:
        fe.processFirstFrame(s["fr"]["prev"][0], **kw("prev"))
    fe.setCandidates(s["pts"], s["n_new"])
    if prefetch:
        fe.prefetchFrame(s["fr"]["cur"][0], **kw("cur"))
        out, m, g = fe.processFrame(None, s["T_guess"], s["T_act"])
    elif split:
        fe.submitFrame(s["fr"]["cur"][0], s["T_guess"], s["T_act"], **kw("cur"))
        out, m, g = fe.waitFrame()
    else:
        out, m, g = fe.processFrame(s["fr"]["cur"][0], s["T_guess"], s["T_act"], **kw("cur"))
    clouds = [fe.cloud_host(l) for l in range(3)]
    corners = [fe.corners(0, l)[0] for l in range(params.n_levels)]
    fe.close()
    return out, m, g, clouds, corners


def _same(a, b, what):
    oa, ma, ga, ca, ka = a
    ob, mb, gb, cb, kb = b
    assert np.array_equal(np.array(oa.T_cur_from_actkey), np.array(ob.T_cur_from_actkey)), what
    assert oa.dense_passes == ob.dense_passes and oa.n_matched == ob.n_matched and oa.tracking_ok == ob.tracking_ok, what
    assert ma.tobytes() == mb.tobytes() and ga.tobytes() == gb.tobytes(), what
    assert bytes(oa.point_stats) == bytes(ob.point_stats) and bytes(oa.pose_stats) == bytes(ob.pose_stats), what
    for l in range(3):
        assert np.array_equal(ca[l], cb[l]), (what, l)
    for x, y in zip(ka, kb):
        assert np.array_equal(x, y), what


@pytest.mark.parametrize("block_matching", [False, True])
def test_batch_of_streams_equals_one_stream_calls(gpu_ctx, block_matching):
    """three streams with different frames, keyframes, candidate lists (different lengths) and poses through svs_frontend_process_frames (frames in
    device memory, one launch per stage) == each stream through its own blocking one-stream call"""
    import torch
    from scavislam_amd import capi
    from scavislam_amd.frontend import StereoFrontend
    ctx, stream = gpu_ctx
    cam, S = _streams(3, block_matching)
    prm = capi.FrontendParams.reference(use_block_matching=block_matching)
    singles = [_single(ctx, cam, s, prm) for s in S]
    B, h, w = len(S), cam["h"], cam["w"]
    fe = StereoFrontend(ctx, cam, max_points=1024, max_keyframes=3, params=prm, n_streams=B)
    dev = torch.device("cuda", 0)

    def frames(name):
        with torch.cuda.stream(stream):
            left = torch.as_tensor(np.stack([s["fr"][name][0] for s in S])).to(dev)
            right = torch.as_tensor(np.stack([s["fr"][name][1] for s in S])).to(dev) if block_matching else None
            disp = None if block_matching else torch.as_tensor(np.stack([s["fr"][name][2] for s in S]).astype(np.float32)).to(dev)
        stream.synchronize()
        return dict(left=left, right=right, disp=disp)

    fe.processFirstFrames(**frames("kf"))
    for b, s in enumerate(S):
        fe.keepKeyframe(0, s["T_kf"], stream=b)
    fe.processFirstFrames(**frames("prev"))
    for b, s in enumerate(S):
        fe.setCandidates(s["pts"], s["n_new"], stream=b)
    cur = frames("cur")
    fe.processFrames(np.stack([s["T_guess"].reshape(12) for s in S]), np.stack([s["T_act"].reshape(12) for s in S]), **cur)
    T_all, ok_all = fe.poses()
    for b in range(B):
        out, m, g = fe.results(b)
        clouds = [fe.cloud_host(l, stream=b) for l in range(3)]
        corners = [fe.corners(b, l)[0] for l in range(3)]
        _same((out, m, g, clouds, corners), singles[b], f"stream {b}")
        assert np.array_equal(T_all[b].reshape(12), np.array(out.T_cur_from_actkey)) and ok_all[b] == 1
        assert out.n_points == len(S[b]["pts"]) and out.n_matched > 100
    # in place: the next frames written straight into the front end's own buffers (no copy kernel), same result as passing pointers
    fe2 = StereoFrontend(ctx, cam, max_points=1024, max_keyframes=3, params=prm, n_streams=B)

    def put(name):
        (pl, sl, bl), (pr, sr, br), (pd, sd, bd) = fe2.inputView()
        for b, s in enumerate(S):
            L, R, D = s["fr"][name]
            Lp = np.zeros((h, sl), np.uint8); Lp[:, :w] = L
            ctx.call("svs_memcpy_h2d", pl + b * bl, Lp.ctypes.data, Lp.nbytes)
            if block_matching:
                Rp = np.zeros((h, sr), np.uint8); Rp[:, :w] = R
                ctx.call("svs_memcpy_h2d", pr + b * br, Rp.ctypes.data, Rp.nbytes)
            else:
                Dp = np.zeros((h, sd), np.float32); Dp[:, :w] = D
                ctx.call("svs_memcpy_h2d", pd + 4 * b * bd, Dp.ctypes.data, Dp.nbytes)
        ctx.sync()

    put("kf"); fe2.processFirstFrames()
    for b, s in enumerate(S):
        fe2.keepKeyframe(0, s["T_kf"], stream=b)
    put("prev"); fe2.processFirstFrames()
    for b, s in enumerate(S):
        fe2.setCandidates(s["pts"], s["n_new"], stream=b)
    put("cur")
    fe2.processFrames(np.stack([s["T_guess"].reshape(12) for s in S]), np.stack([s["T_act"].reshape(12) for s in S]))
    for b in range(B):
        out, m, g = fe2.results(b)
        _same((out, m, g, [fe2.cloud_host(l, stream=b) for l in range(3)], [fe2.corners(b, l)[0] for l in range(3)]), singles[b], f"in place, stream {b}")
    fe.close(); fe2.close()


def test_throughput_batch_replicas_agree(gpu_ctx):
    """At the batch size of the bench (more streams than CUs: one tracker workgroup per stream in its four-waves-per-SIMD build, the side stream beside
    it, every stage launched once for all streams) the size-independent properties: streams that were given the same inputs return the same BITS
    wherever they sit in the batch (no cross-talk, no dependence on the position), and each agrees with the three-stream batch -- which
    tests above hold bit-equal to the one-stream call and tests/test_gpu_ref_frame.py to the reference -- up to the dense tracker's summation order
    (a different workgroup count per stream): pose to 1e-9, the matcher's records equal but for the one-in-a-thousand point whose warped patch feels the
    last bits of the pose."""
    import torch
    from scavislam_amd import capi
    from scavislam_amd.frontend import StereoFrontend
    ctx, stream = gpu_ctx
    cam, S3 = _streams(3)
    prm = capi.FrontendParams.reference()
    dev = torch.device("cuda", 0)

    def run(S):
        B = len(S)
        fe = StereoFrontend(ctx, cam, max_points=1024, max_keyframes=3, params=prm, n_streams=B)

        def frames(name):
            with torch.cuda.stream(stream):
                left = torch.as_tensor(np.stack([s["fr"][name][0] for s in S])).to(dev)
                disp = torch.as_tensor(np.stack([s["fr"][name][2] for s in S]).astype(np.float32)).to(dev)
            stream.synchronize()
            return dict(left=left, disp=disp)

        fe.processFirstFrames(**frames("kf"))
        for b, s in enumerate(S):
            fe.keepKeyframe(0, s["T_kf"], stream=b)
        fe.processFirstFrames(**frames("prev"))
        for b, s in enumerate(S):
            fe.setCandidates(s["pts"], s["n_new"], stream=b)
        fe.processFrames(np.stack([s["T_guess"].reshape(12) for s in S]), np.stack([s["T_act"].reshape(12) for s in S]), **frames("cur"))
        res = [fe.results(b) for b in range(B)]
        clouds = [[fe.cloud_host(l, stream=b) for l in range(3)] for b in (0, 1, 2, B - 3, B - 2, B - 1)]
        fe.close()
        return res, clouds

    small, _ = run(S3)
    B = 2 * 256 + 30                                  # above two streams per CU of an MI355X: also the one-stream schedule of the chain
    big, clouds = run([S3[b % 3] for b in range(B)])
    for b in range(B):
        out, m, g = big[b]
        o0, m0, g0 = big[b % 3]
        assert np.array_equal(np.array(out.T_cur_from_actkey), np.array(o0.T_cur_from_actkey)) and out.dense_passes == o0.dense_passes, b
        assert m.tobytes() == m0.tobytes() and g.tobytes() == g0.tobytes() and bytes(out.point_stats) == bytes(o0.point_stats), b
    for k in range(3):
        tail = B - 3 + k                             # a stream at the end of the batch; its inputs are those of stream tail % 3
        for l in range(3):
            assert np.array_equal(clouds[3 + k][l], clouds[tail % 3][l]), (k, l)
        out, m, g = big[k]
        os_, ms, gs = small[k]
        assert out.dense_passes == os_.dense_passes and abs(out.n_matched - os_.n_matched) <= 2 and out.tracking_ok == os_.tracking_ok == 1
        np.testing.assert_allclose(np.array(out.T_cur_from_actkey), np.array(os_.T_cur_from_actkey), rtol=0, atol=1e-9)
        # the matcher warps the key patch with the tracked pose and truncates to u8: a pose that differs in its last bits moves one patch pixel in about one
        # point per thousand (bit-exactness note), which shows in that point's score and, rarely, in its winner
        same = (m["status"] == ms["status"]) & (m["u"] == ms["u"]) & (m["v"] == ms["v"]) & (m["znssd"] == ms["znssd"])
        assert same.mean() > 0.99 and (m["status"] == ms["status"]).mean() > 0.998, (k, same.mean())
        assert (g["accepted"] == gs["accepted"]).mean() > 0.995


def test_tracker_grid_order_does_not_change_results(gpu_ctx):
    """Context option "trk_balance" (dense.hip): in a batch of two or more streams per CU the tracker's workgroups are launched in the order of the LM work
    their streams needed in the LAST frame (dealt round-robin to the XCDs) instead of stream order; 2 = the experimental variant that also gives the longest
    streams 2 .. 4 workgroups.  A stream's result must not depend on where its workgroup sits: three tracked frames (the order changes from the second on),
    0 vs 1 bit-equal in every output, 2 equal up to the summation order of the split streams (pose 1e-9, same LM pass counts, no failed stream)."""
    import torch
    from scavislam_amd import capi
    from scavislam_amd.frontend import StereoFrontend
    ctx, stream = gpu_ctx
    cam, S3 = _streams(3)
    prm = capi.FrontendParams.reference()
    dev = torch.device("cuda", 0)
    B = 2 * 256 + 30
    S = [S3[b % 3] for b in range(B)]

    def frames(name):
        with torch.cuda.stream(stream):
            left = torch.as_tensor(np.stack([s["fr"][name][0] for s in S])).to(dev)
            disp = torch.as_tensor(np.stack([s["fr"][name][2] for s in S]).astype(np.float32)).to(dev)
        stream.synchronize()
        return dict(left=left, disp=disp)

    F = {n: frames(n) for n in ("kf", "prev", "cur")}
    T_guess, T_act = np.stack([s["T_guess"].reshape(12) for s in S]), np.stack([s["T_act"].reshape(12) for s in S])

    def run(mode, pipeline=0, flat=1):
        ctx.set_option("trk_balance", mode)
        ctx.set_option("fe_pipeline", pipeline)
        ctx.set_option("trk_flat", flat)
        try:
            fe = StereoFrontend(ctx, cam, max_points=1024, max_keyframes=3, params=prm, n_streams=B)
            fe.processFirstFrames(**F["kf"])
            fe.keepKeyframes(0, np.stack([s["T_kf"].reshape(12) for s in S]))
            fe.processFirstFrames(**F["prev"])
            for b, s in enumerate(S):
                fe.setCandidates(s["pts"], s["n_new"], stream=b)
            outs = []
            for name in ("cur", "prev", "cur"):          # prev -> cur, cur -> prev, prev -> cur: three tracked frames with different LM pass counts per stream
                fe.processFrames(T_guess, T_act, **F[name])
                outs.append([fe.results(b) for b in (0, 1, 2, 257, B - 2, B - 1)] + [fe.poses()])
            fe.close()
            return outs
        finally:
            ctx.set_option("trk_balance", 1)
            ctx.set_option("fe_pipeline", 1)
            ctx.set_option("trk_flat", 1)

    base, order, split = run(0), run(1, 1), run(2)
    # "trk_flat" (round 6): big batches run the flat state-machine kernel (the sweep inlined, the LM state in LDS: no callee-saved traffic around 18 calls per frame);
    # 0 = the round-5 kernel (the sweep as a call).  Same sweep, sums, solve and decisions: every output bit-equal, in stream order and in the balanced order
    for mode in (0, 1):
        old = run(mode, 0, flat=0)
        new = base if mode == 0 else order
        for k in range(3):
            for (o0, m0, g0), (o1, m1, g1) in zip(old[k][:-1], new[k][:-1]):
                assert np.array_equal(np.array(o0.T_cur_from_actkey), np.array(o1.T_cur_from_actkey)) and o0.dense_passes == o1.dense_passes >= 0, ("trk_flat", mode, k)
                assert m0.tobytes() == m1.tobytes() and g0.tobytes() == g1.tobytes() and bytes(o0.point_stats) == bytes(o1.point_stats), ("trk_flat", mode, k)
            assert np.array_equal(old[k][-1][0], new[k][-1][0]) and np.array_equal(old[k][-1][1], new[k][-1][1]), ("trk_flat", mode, k)
    # the cross-frame pipeline ("fe_pipeline": the pyramid of frame N+1 on the side stream beside frame N's pose refinement / gate / cloud; the three frames above
    # are enqueued back to back only up to the blocking result reads -- here they are enqueued without a read in between, the bench's pattern)
    ready = torch.cuda.Event()
    with torch.cuda.stream(stream):
        ready.record(stream)                             # the frames in F are complete (uploaded and synchronised above): the event the pipelined schedule asks for

    def run_async(pipeline, produced_on_the_stream=False):
        ctx.set_option("fe_pipeline", pipeline)
        try:
            fe = StereoFrontend(ctx, cam, max_points=1024, max_keyframes=3, params=prm, n_streams=B)
            fe.processFirstFrames(**F["kf"])
            fe.keepKeyframes(0, np.stack([s["T_kf"].reshape(12) for s in S]))
            fe.processFirstFrames(**F["prev"])
            for b, s in enumerate(S):
                fe.setCandidates(s["pts"], s["n_new"], stream=b)
            if produced_on_the_stream:
                # ONE pair of buffers, rewritten by a copy kernel on the context's stream right in front of every call, no ready event: only the
                # stream's order says when a frame is complete, so the library must read it in that order (no side-stream pyramid ahead of the copy)
                with torch.cuda.stream(stream):
                    buf_l, buf_d = torch.empty_like(F["cur"]["left"]), torch.empty_like(F["cur"]["disp"])
            for name in ("cur", "prev", "cur", "prev", "cur"):
                if produced_on_the_stream:
                    with torch.cuda.stream(stream):
                        buf_l.copy_(F[name]["left"]); buf_d.copy_(F[name]["disp"])
                    fe.processFrames(T_guess, T_act, left=buf_l, disp=buf_d)
                else:
                    fe.processFrames(T_guess, T_act, ready_event=ready if pipeline else None, **F[name])
            out = [fe.results(b) for b in (0, 1, 2, 257, B - 1)] + [fe.poses(), [fe.corners(b, l) for b in (0, 1, B - 1) for l in range(3)]]
            fe.close()
            return out
        finally:
            ctx.set_option("fe_pipeline", 1)
    a0, a1, a2 = run_async(0), run_async(1), run_async(1, produced_on_the_stream=True)
    for (o0, m0, g0), (o2, m2, g2) in zip(a0[:5], a2[:5]):
        assert np.array_equal(np.array(o0.T_cur_from_actkey), np.array(o2.T_cur_from_actkey)) and o0.dense_passes == o2.dense_passes >= 0, "frames produced on the stream"
        assert m0.tobytes() == m2.tobytes() and g0.tobytes() == g2.tobytes()
    for (o0, m0, g0), (o1, m1, g1) in zip(a0[:5], a1[:5]):
        assert np.array_equal(np.array(o0.T_cur_from_actkey), np.array(o1.T_cur_from_actkey)) and o0.dense_passes == o1.dense_passes >= 0
        assert m0.tobytes() == m1.tobytes() and g0.tobytes() == g1.tobytes() and bytes(o0.point_stats) == bytes(o1.point_stats)
    assert np.array_equal(a0[5][0], a1[5][0]) and np.array_equal(a0[5][1], a1[5][1])
    for c0, c1 in zip(a0[6], a1[6]):
        assert all(np.array_equal(x, y) for x, y in zip(c0, c1))          # corners, cells, thresholds of the last frame
    for k in range(3):
        for (o0, m0, g0), (o1, m1, g1), (o2, m2, g2) in zip(base[k][:-1], order[k][:-1], split[k][:-1]):
            assert np.array_equal(np.array(o0.T_cur_from_actkey), np.array(o1.T_cur_from_actkey)) and o0.dense_passes == o1.dense_passes >= 0, k
            assert m0.tobytes() == m1.tobytes() and g0.tobytes() == g1.tobytes() and bytes(o0.point_stats) == bytes(o1.point_stats), k
            assert o2.dense_passes == o0.dense_passes and o2.tracking_ok == o0.tracking_ok, k
        T0, ok0 = base[k][-1]
        T1, ok1 = order[k][-1]
        T2, ok2 = split[k][-1]
        assert np.array_equal(T0, T1) and np.array_equal(ok0, ok1), k
        np.testing.assert_allclose(T2, T0, rtol=0, atol=1e-9)
        assert np.array_equal(ok2, ok0), k


def test_prefetch_and_split_call_equal_blocking_call(gpu_ctx):
    """svs_frontend_prefetch_frame (upload on the copy stream) + process_frame(NULL), and submit_frame / wait_frame, return what the blocking call returns;
    a three-frame sequence with the next frame prefetched while the current one is in flight keeps doing so."""
    from scavislam_amd import capi, synth
    from scavislam_amd.frontend import StereoFrontend
    ctx, stream = gpu_ctx
    cam, S = _streams(1)
    prm = capi.FrontendParams.reference()
    ref = _single(ctx, cam, S[0], prm)
    _same(_single(ctx, cam, S[0], prm, prefetch=True), ref, "prefetched")
    _same(_single(ctx, cam, S[0], prm, split=True), ref, "submit / wait")
    # sequence: frame k+1 is prefetched between submit(k) and wait(k)
    sc = synth.Scene(2011)
    traj = synth.trajectory(8)
    frames = [sc.render(cam, traj[i], seed=i) for i in range(5)]
    rng = np.random.default_rng(3)
    pts = synth.candidate_points(rng, cam, np.maximum(frames[0][1], 0), traj[0], (300, 150, 50))
    results = {}
    for mode in ("blocking", "overlapped"):
        fe = StereoFrontend(ctx, cam, max_points=1024, max_keyframes=2, params=prm)
        fe.processFirstFrame(frames[0][0], disp=frames[0][1])
        fe.keepKeyframe(0, traj[0])
        fe.setCandidates(pts, 100)
        T, seq = I34.copy(), []
        if mode == "overlapped":
            fe.prefetchFrame(frames[1][0], disp=frames[1][1])
        for k in range(1, 5):
            if mode == "blocking":
                out, m, g = fe.processFrame(frames[k][0], T, traj[0], disp=frames[k][1])
            else:
                fe.submitFrame(None, T, traj[0])
                if k + 1 < 5:
                    fe.prefetchFrame(frames[k + 1][0], disp=frames[k + 1][1])
                out, m, g = fe.waitFrame()
            T = np.array(out.T_cur_from_actkey).reshape(3, 4)
            seq.append((T.copy(), m.tobytes(), out.dense_passes))
        results[mode] = seq
        fe.close()
    for a, b in zip(results["blocking"], results["overlapped"]):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2]
    assert np.abs(results["blocking"][-1][0] - I34).max() > 1e-3


@pytest.mark.parametrize("block_matching", [False, True])
def test_side_stream_schedule_equals_one_stream(gpu_ctx, block_matching):
    """By default the chain enqueues FAST (and block matching) on a side stream beside the dense tracker (option "fe_overlap"); the schedule must not
    change a bit of any output, over several frames in a row (the side stream's work of frame N + 1 must wait for frame N's matcher, which reads the
    same corner bitmaps)."""
    from scavislam_amd import capi
    ctx, stream = gpu_ctx
    cam, S = _streams(1, block_matching)
    prm = capi.FrontendParams.reference(use_block_matching=block_matching)
    runs = {}
    for mode in (1, 0):
        ctx.set_option("fe_overlap", mode)
        try:
            runs[mode] = [_single(ctx, cam, S[0], prm), _single(ctx, cam, S[0], prm, prefetch=True)]
        finally:
            ctx.set_option("fe_overlap", 1)
    for a, b in zip(runs[1], runs[0]):
        _same(a, b, "side stream")
    assert runs[1][0][0].n_matched > 100


def test_two_front_end_levels(gpu_ctx):
    """use_n_levels_in_frontent = 2 (the reference's code default, stereo_frontend.cpp:68): FAST and the matcher run on levels 0 and 1 only.  Corner
    lists of those levels equal the oracle's; candidates of levels 0 / 1 get the results of the three-level run (levels are independent), a level-2
    candidate (the reference never creates one then) comes back unmatched."""
    import oracle as O
    from scavislam_amd import capi
    ctx, stream = gpu_ctx
    cam, S = _streams(1)
    s = S[0]
    three = _single(ctx, cam, s, capi.FrontendParams.reference(n_levels=3))
    two = _single(ctx, cam, s, capi.FrontendParams.reference(n_levels=2))
    lv = s["pts"]["anchor_level"]
    m3, m2 = three[1], two[1]
    assert m2[lv < 2].tobytes() == m3[lv < 2].tobytes() and (lv == 2).sum() > 10
    assert (m2["status"][lv == 2] != 0).all()
    pyr = O.build_pyramid(s["fr"]["cur"][0])
    grids = [O.fastgrid_for_level(pyr[l].shape[1], pyr[l].shape[0], l) for l in range(2)]
    for name in ("kf", "prev"):                                  # the front end's FastGrid saw two first frames (5 trials) before this one
        p = O.build_pyramid(s["fr"][name][0])
        for l in range(2):
            O.fastgrid_detect_adaptively(grids[l], p[l], 5)
    for l in range(2):
        xy, cc, et = O.fastgrid_detect_adaptively(grids[l], pyr[l], 6)
        assert np.array_equal(two[4][l], xy), l
        assert np.array_equal(three[4][l], xy), l
    assert two[0].n_matched < three[0].n_matched and two[0].tracking_ok == 1


def test_too_few_matches_and_unkept_keyframes(gpu_ctx):
    """matchAndTrack returns false below 20 observations, before calcFastMotionOnly (stereo_frontend.cpp:1053-1056): tracking_ok = 0 and the pose is the
    dense tracker's.  A candidate naming a keyframe slot that was never kept is refused (it would send the matcher after a null pyramid)."""
    from scavislam_amd import capi
    from scavislam_amd.frontend import StereoFrontend
    ctx, stream = gpu_ctx
    cam, S = _streams(1)
    s = S[0]
    full = _single(ctx, cam, s, capi.FrontendParams.reference())
    k = int(np.searchsorted(np.cumsum(full[1]["status"] == 0), 12)) + 1     # the first records holding 12 matches (a record's result does not depend on the others)
    few = dict(s, pts=s["pts"][:k], n_new=min(10, k))
    out, m, g, clouds, corners = _single(ctx, cam, few, capi.FrontendParams.reference())
    n_ok = int((m["status"] == 0).sum())
    assert 0 < n_ok < 20 and out.tracking_ok == 0 and out.n_matched == n_ok and out.pose_stats.status == 3
    # the pose is what the dense tracker left: same tracker input as the full run, whose refined pose differs from it
    out0, m0, g0, _, _ = _single(ctx, cam, dict(s, pts=s["pts"][:0], n_new=0), capi.FrontendParams.reference())
    assert np.array_equal(np.array(out.T_cur_from_actkey), np.array(out0.T_cur_from_actkey))
    assert not np.array_equal(np.array(out.T_cur_from_actkey), np.array(full[0].T_cur_from_actkey))
    assert out0.n_points == 0 and out0.tracking_ok == 0
    fe = StereoFrontend(ctx, cam, max_points=64, max_keyframes=3)
    fe.processFirstFrame(s["fr"]["kf"][0], disp=s["fr"]["kf"][2])
    fe.keepKeyframe(0, s["T_kf"])
    bad = s["pts"][:8].copy()
    bad["kf_index"][3] = 2                                        # slot 2 exists but holds nothing
    with pytest.raises(capi.SvsError):
        fe.setCandidates(bad, 4)
    bad["kf_index"][3] = -1                                       # "anchor frame not in keyframe_map" is a legal candidate (matcher.cpp:336-339)
    fe.setCandidates(bad, 4)
    fe.close()
