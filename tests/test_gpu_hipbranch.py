"""The HIP branch COMPILED IN PLACE (missing 2; north_star: "drops into stereo_slam unchanged").

oracle/_ref/libsvs_hipbranch_frame.so is the reference's own StereoFrontend::processFrame translation unit -- processFrame, computeFastCorners, matchAndTrack,
processMatchedPoints (stereo_frontend.cpp:183-306,656-679,832-1065), GuidedMatcher::match (matcher.cpp), their containers (hash maps of keyframes and
vertices, lists of shared CandidatePoints, TrackData, QuadTrees) -- compiled with SCAVISLAM_HIP_SUPPORT defined and an `#if defined(SCAVISLAM_HIP_SUPPORT)`
branch inserted AT THE REFERENCE'S OWN SWITCH POINTS (the three SCAVISLAM_CUDA_SUPPORT switches of processFrame, the head of computeFastCorners and of
GuidedMatcher::match, the calcFastMotionOnly call).  The branch bodies are INTEGRATION.md's glue over include/scavislam_hip.hpp (oracle/ref_shim/
hipbranch_glue.inc) and end in scavislam_amd/libscavislam_hip.so, i.e. on the GPU.  The recipe is oracle/Makefile; the library is built where
/root/reference exists and travels prebuilt.

It is compared with the SAME translation unit compiled without the define (libsvs_ref_frame.so = the reference's CPU build): what the reference draws and
hands on per frame must be the same -- accepted points and both line ends per level bit-equal, refined pose 1e-6, track length, clouds."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _have(name):
    import oracle as O
    return os.path.exists(os.path.join(os.path.dirname(O.__file__), "_ref", name))


@pytest.mark.parametrize("camname", ["newcollege", "default"])
def test_reference_process_frame_with_hip_branch_in_place(gpu_ctx, camname):
    if not (_have("libsvs_hipbranch_frame.so") and _have("libsvs_ref_frame.so")):
        pytest.skip("oracle/_ref/libsvs_hipbranch_frame.so / libsvs_ref_frame.so not present (built by oracle/Makefile where /root/reference exists)")
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.ctypes_types import level_cams
    cam = synth.CAM_DEFAULT if camname == "default" else synth.CAM_NEWCOLLEGE
    cams = level_cams(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
    sc = synth.Scene(2011)
    traj = synth.trajectory(8)
    k0, k1, p, c = 0, 2, 4, 5
    img_k0, disp_k0 = sc.render(cam, traj[k0], seed=k0)
    img_k1, disp_k1 = sc.render(cam, traj[k1], seed=k1)
    img_p, disp_p = sc.render(cam, traj[p], seed=p)
    img_c, disp_c = sc.render(cam, traj[c], seed=c)
    disp_c = disp_c.copy(); disp_c[::9, ::4] = 0.0; disp_c[200:260, 100:300] = -1.0
    rng = np.random.default_rng(31)
    pts = np.concatenate([synth.candidate_points(rng, cam, disp_k0, traj[k0], (500, 250, 90), kf_index=0),
                          synth.candidate_points(rng, cam, disp_k1, traj[k1], (500, 250, 90), kf_index=1)])
    rng.shuffle(pts)
    pts["point_id"] = np.arange(len(pts))
    pts[0]["kf_index"] = -1                                   # anchor keyframe not in the vertex map
    pts[1]["anchor_obs_pyr"][:2] = (2.0, 2.0)
    pts[2]["xyz_anchor"] *= 0.05
    T_act = traj[k1]
    T_prev_from_act = synth.pose_mul(traj[p], synth.pose_inv(T_act))
    n = len(pts)
    rngl = np.random.default_rng(3)
    list_of = np.where(rngl.random(n) < 0.12, 1, np.where(rngl.random(n) < 0.1, 0, -1)).astype(np.int32)
    list_of[pts["kf_index"] < 0] = -1
    pyr_k = [O.build_pyramid(img_k0), O.build_pyramid(img_k1)]
    pyr_p, pyr_c = O.build_pyramid(img_p), O.build_pyramid(img_c)
    clouds_prev = [O.pointcloud_cpu(disp_p, cams[l], l, T_prev_from_act) for l in range(3)]
    fl = [O.convert_sobel(q) for q in pyr_c]
    args = (pyr_k, [traj[k0].reshape(12), traj[k1].reshape(12)], 1, [(0, 37)], cams, pts, list_of, T_prev_from_act, clouds_prev, pyr_p, pyr_c,
            [f[0] for f in fl], [f[1] for f in fl], [f[2] for f in fl], disp_c)
    ref = O.ref_process_frame(*args)                          # the reference's CPU build
    hip = O.ref_process_frame(*args, hip_branch=True)         # the same code, HIP branch in place
    assert ref["ok"] and hip["ok"] and not hip["is_frame_dropped"]
    n_lines = 0
    for l in range(3):
        assert hip["lines"][l].shape == ref["lines"][l].shape, (l, hip["lines"][l].shape, ref["lines"][l].shape)
        assert np.array_equal(hip["lines"][l], ref["lines"][l]), f"draw lines of level {l}"
        n_lines += len(ref["lines"][l])
    assert n_lines > 40 and (np.concatenate(ref["lines"])[:, 0] == 1).sum() > 3
    dT = np.abs(hip["T"] - ref["T"]).max()
    assert dT < 1e-6, dT
    assert abs(hip["av_track_length"] - ref["av_track_length"]) <= 1e-9 * max(1.0, ref["av_track_length"])
    same_rimg = []
    for l in range(3):
        a, b = hip["clouds"][l], ref["clouds"][l]
        assert np.array_equal(a[..., 3], b[..., 3])
        np.testing.assert_allclose(a[..., :3], b[..., :3], rtol=1e-5, atol=1e-5)
        # residual images (GUI output of the dense tracker): same samples valid; grey values agree where the LM trajectories agree
        ra, rb = hip["rimg"][l], ref["rimg"][l]
        assert np.array_equal(ra[..., 3], rb[..., 3])
        same_rimg.append(float(np.mean(np.abs(ra - rb).max(-1) < 2e-3)))
    assert min(same_rimg) > 0.98, same_rimg
    print(f"{camname}: HIP branch in place vs the reference's CPU build: pose deviation {dT:.2e}, {n_lines} draw lines identical, residual images agree on {same_rimg}")


def test_reference_process_frame_cuda_build_with_hip_branch_in_place(gpu_ctx):
    """The reference's CUDA build (SCAVISLAM_CUDA_SUPPORT) with the HIP branch in place AT THE CUDA SWITCHES (missing 6): libsvs_hipbranch_frame_cuda.so
    is the translation unit of libsvs_ref_frame_cuda.so compiled with SCAVISLAM_HIP_SUPPORT as well -- declarations, members and the matcher radius (4) are the CUDA
    build's, `tracker_.denseTrackingGpu` / the disparity upload / `computeDensePointCloudGpu` (stereo_frontend.cpp:192-196,213-215,298-302) and the CUDA branch of
    preprocessing run in libscavislam_hip.so.  Against the same unit running the reference's own kernels through the CUDA emulator: draw lists identical, pose within
    1e-7 (measured 3.3e-9: the reference's kernels sum in f32, the HIP tracker in f64), clouds 4e-6 absolute (measured: one f32 ulp)."""
    if not (_have("libsvs_hipbranch_frame_cuda.so") and _have("libsvs_ref_frame_cuda.so")):
        pytest.skip("oracle/_ref/libsvs_hipbranch_frame_cuda.so / libsvs_ref_frame_cuda.so not present")
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.ctypes_types import level_cams
    from test_gpu_ref_frame import I34, _cuda_case
    case, pts, list_of, disp_c = _cuda_case()
    camd = case["cam"]
    cams = level_cams(camd["f"], camd["cx"], camd["cy"], camd["b"], camd["w"], camd["h"])
    fp, _, _ = O.preprocess_gpu_sem(case["img_prev"])
    fc, dx, dy = O.preprocess_gpu_sem(case["img_cur"])
    cloud_prev = O.ref_pointcloud_gpu(case["disp_prev"], synth.level_cams(camd), I34)
    pyr_k, pyr_c = O.build_pyramid(case["img_prev"]), O.build_pyramid(case["img_cur"])
    args = ([pyr_k], [I34.reshape(12)], 0, [], cams, pts, list_of, I34, cloud_prev, fp, pyr_c, fc, dx, dy, disp_c)
    ref = O.ref_process_frame(*args, cuda_build=True)
    hip = O.ref_process_frame(*args, cuda_build=True, hip_branch=True)
    assert ref["ok"] and hip["ok"]
    n_lines = 0
    for l in range(3):
        assert hip["lines"][l].shape == ref["lines"][l].shape, (l, hip["lines"][l].shape, ref["lines"][l].shape)
        assert np.array_equal(hip["lines"][l], ref["lines"][l]), f"draw lines of level {l}"
        n_lines += len(ref["lines"][l])
    assert n_lines > 40
    dT = np.abs(hip["T"] - ref["T"]).max()
    assert dT < 1e-7, dT          # measured 3.3e-9
    assert abs(hip["av_track_length"] - ref["av_track_length"]) <= 1e-9 * max(1.0, ref["av_track_length"])
    for l in range(3):
        a, b = hip["clouds"][l], ref["clouds"][l]
        assert np.array_equal(a[..., 3], b[..., 3])
        valid = a[..., 3] > 0
        da = np.abs(a[..., :3] - b[..., :3])[valid]
        print(f"  CUDA build, HIP branch in place, cloud: max abs deviation {da.max():.2e}, max relative {(da / np.maximum(np.abs(b[..., :3][valid]), 1e-3)).max():.2e}")
        np.testing.assert_allclose(a[..., :3], b[..., :3], rtol=0, atol=4e-6)      # measured 1.9e-6 = one f32 ulp at 16..32 m
    print(f"CUDA build with the HIP branch in place: pose deviation {dT:.2e}, {n_lines} draw lines identical")


def _graph_tables(prob, n_outer, rng):
    """a double window in the shape of the reference's tables (ids from one counter, observations of frames outside the window in the vis_sets,
    marginalised pose-pose edges with an OUTER end + co-visibility edges without a constraint), from the flat arrays of synth.ba_window"""
    import oracle as O
    P, Lm = len(prob["poses"]), len(prob["psi"])
    pose_ids = 100 + 3 * np.arange(P)
    point_ids = 5000 + 7 * np.arange(Lm)
    wtype = (np.arange(P) >= P - n_outer).astype(np.int32)
    e = prob["edges"]
    psi = prob["psi"]
    xyz = np.stack([psi[:, 0] / psi[:, 2], psi[:, 1] / psi[:, 2], 1.0 / psi[:, 2]], 1)
    anchor_of = np.zeros(Lm, np.int64)
    anchor_of[e["point"]] = e["anchor"]
    level = np.round(np.log(1.0 / e["info"][:, 0]) / np.log(4.0)).astype(np.int32)
    extra_pt = rng.choice(Lm, 40, replace=False)
    obs_point = np.concatenate([point_ids[e["point"]], point_ids[extra_pt]])
    obs_pose = np.concatenate([pose_ids[e["pose"]], np.where(np.arange(40) % 2 == 0, 7, 9001)])      # two frames that are not in the window
    obs_level = np.concatenate([level, np.zeros(40, np.int32)])
    obs_center = np.concatenate([e["obs"], rng.uniform(0, 400, (40, 3))])
    pe_ids, pe_marg, pe_T12, pe_L12, pe_L21 = [], [], [], [], []
    for c in prob["cons"]:
        pe_ids.append((pose_ids[c["pose1"]], pose_ids[c["pose2"]])); pe_marg.append(1)
        pe_T12.append(O.se3_inv(c["T_21"]).reshape(12)); pe_L21.append(c["info"]); pe_L12.append(0.5 * c["info"] + np.eye(6).reshape(36))
    for i in (0, 3):
        pe_ids.append((pose_ids[i + 1], pose_ids[i])); pe_marg.append(0)
        pe_T12.append(np.eye(3, 4).reshape(12)); pe_L12.append(np.zeros(36)); pe_L21.append(np.zeros(36))
    return (pose_ids, wtype, prob["poses"], point_ids, pose_ids[anchor_of], xyz, obs_point, obs_pose, obs_level, obs_center, pe_ids, pe_marg, pe_T12, pe_L12, pe_L21,
            prob["cam"])


def _flat_from_recorded_graph(r):
    """the flat arrays of the HIP back end from what the reference's copyDataToG2o handed to (a recording) g2o: vertices in the order of addVertex,
    edges in the order of addEdge -- every marginalised pose-pose edge is there twice, once per direction, as the reference adds them"""
    from scavislam_amd.ctypes_types import BA_CONSTRAINT_DTYPE, BA_EDGE_DTYPE
    v, est, ed, dd = r["vertices"], r["estimates"], r["edges"], r["edge_data"]
    is_pose = v[:, 0] == 0
    pose_ids, point_ids = v[is_pose, 1], v[~is_pose, 1]
    pidx = {int(i): k for k, i in enumerate(pose_ids)}
    lidx = {int(i): k for k, i in enumerate(point_ids)}
    poses, psi = est[is_pose].copy(), est[~is_pose][:, :3].copy()
    proj = ed[:, 0] == 0
    edges = np.zeros(int(proj.sum()), BA_EDGE_DTYPE)
    edges["obs"] = dd[proj][:, :3]
    edges["info"] = dd[proj][:, [12, 16, 20]]
    edges["point"] = [lidx[int(i)] for i in ed[proj, 1]]
    edges["pose"] = [pidx[int(i)] for i in ed[proj, 2]]
    edges["anchor"] = [pidx[int(i)] for i in ed[proj, 3]]
    cons = np.zeros(int((~proj).sum()), BA_CONSTRAINT_DTYPE)
    cons["T_21"] = dd[~proj][:, :12]
    cons["info"] = dd[~proj][:, 12:48]
    cons["pose1"] = [pidx[int(i)] for i in ed[~proj, 1]]
    cons["pose2"] = [pidx[int(i)] for i in ed[~proj, 2]]
    return pose_ids, point_ids, poses, psi, edges, cons


@pytest.mark.parametrize("P,L,n_outer", [(12, 300, 3), (30, 2500, 5)])
def test_slamgraph_optimize_hip_branch_and_recorded_graph(gpu_ctx, P, L, n_outer):
    """Back end (slam_graph.cpp:312-355,907-1080).  (1) The reference's copyDataToG2o, compiled as it is, hands its graph to a recording g2o
    (libsvs_ref_slamgraph.so); THAT graph -- not a hand-made problem -- is given to svs_ba_set_problem / svs_ba_optimize and to the oracle: same LM
    trajectory, state update within 1e-6 relative.  (2) SlamGraph::optimize compiled with the SCAVISLAM_HIP_SUPPORT branch in place
    (libsvs_hipbranch_slamgraph.so: tables -> scavislam_hip::SlamGraphBA::optimizeWindow -> GPU -> tables): the poses and points it leaves in the
    reference's tables equal the oracle's result on the recorded graph the same way."""
    have_ref = _have("libsvs_hipbranch_slamgraph.so") and _have("libsvs_ref_slamgraph.so")
    fixture = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_slamgraph_recorded.npz")
    if not have_ref and P != 12:
        pytest.skip("oracle/_ref libraries not present (built by oracle/Makefile where /root/reference exists); the 12-keyframe case runs from tests/golden")
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams, Cam
    ctx, stream = gpu_ctx
    prob = synth.ba_window(P=P, L=L, seed=5, n_outer=n_outer)
    if have_ref:
        tables = _graph_tables(prob, n_outer, np.random.default_rng(1))
        rec = O.ref_slamgraph_optimize(*tables, 2, True, 3.0, 0.0)                    # Backend's own call: OptParams(2, true, 3) (backend.cpp:187)
        if P == 12:                                                                   # the committed fixture IS this recording (tests/golden/make_golden_slamgraph.py)
            fx = np.load(fixture)
            assert all(np.array_equal(fx[k], rec[k]) for k in ("vertices", "estimates", "edges", "edge_data"))
    else:                                                                             # without oracle/_ref: part (1) on the graph the reference recorded when the fixture was made
        rec = dict(np.load(fixture))
    pose_ids, point_ids, poses0, psi0, edges, cons = _flat_from_recorded_graph(rec)
    assert len(cons) == 2 * len(prob["cons"]) and len(edges) == len(prob["edges"])
    c = prob["cam"]
    camc = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])
    prm = BaParams.reference_defaults()
    poses_ref, psi_ref, st_ref = O.ba_optimize(poses0, psi0, edges, cons, camc, prm)
    upd_p, upd_l = np.abs(poses_ref - poses0).max(), np.abs(psi_ref - psi0).max()
    assert upd_p > 1e-4 and st_ref.accepted >= 1
    # (1) the recorded graph through the Python mirror of the C ABI
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(poses0, psi0, edges, cons, camc, prm)
    st = opt.optimize()
    poses_hip, psi_hip = opt.restoreDataFromG2o()
    opt.close()
    assert (st.trials, st.accepted, st.terminated) == (st_ref.trials, st_ref.accepted, st_ref.terminated)
    assert np.abs(poses_hip - poses_ref).max() <= 1e-6 * upd_p
    # landmarks: the 1e-6 bar plus what the back-substitution makes of the pose update's own 1e-9 on a weak-parallax landmark (tests/test_gpu_ba.py, same bar)
    import np_model as M
    amp = M.landmark_amplification(poses0, psi0, edges, (c["f"], c["cx"], c["cy"], c["b"]), st_ref.lambda_final)
    err_pose = np.abs(poses_hip - poses_ref).max()

    def landmarks_ok(psi_got, psi_exp, a):
        err_l = np.abs(psi_got - psi_exp).max(1)
        assert (err_l <= 1e-6 * upd_l + 10.0 * a * max(err_pose, 1e-12)).all(), float((err_l - 10.0 * a * err_pose).max() / upd_l)
        assert int((err_l > 1e-6 * upd_l).sum()) <= 8
    landmarks_ok(psi_hip, psi_ref, amp)
    if not have_ref:
        print(f"P={P}: recorded graph from the fixture ({len(edges)} observation edges, {len(cons)} pose-pose edges) -> HIP vs oracle: poses {np.abs(poses_hip - poses_ref).max():.1e}")
        return
    # (2) the reference's optimize() with the HIP branch in place, on the same tables
    hip = O.ref_slamgraph_optimize(*tables, 2, True, 3.0, 0.0, hip_branch=True)
    assert (hip["stats"]["trials"], hip["stats"]["accepted"]) == (st_ref.trials, st_ref.accepted)
    # tables come back in the order they were given: poses by id, points by id; the oracle's result in the recorded (= addVertex) order
    order_p = np.argsort(pose_ids); order_l = np.argsort(point_ids)
    exp_poses = poses_ref[order_p][np.argsort(np.argsort(tables[0]))]
    assert np.abs(hip["poses_out"] - exp_poses).max() <= 1e-6 * upd_p
    psi_sorted = psi_ref[order_l][np.argsort(np.argsort(tables[3]))]
    exp_xyz = np.stack([psi_sorted[:, 0] / psi_sorted[:, 2], psi_sorted[:, 1] / psi_sorted[:, 2], 1.0 / psi_sorted[:, 2]], 1)
    xyz0 = np.asarray(tables[5])
    got = hip["points_out"]
    landmarks_ok(np.stack([got[:, 0] / got[:, 2], got[:, 1] / got[:, 2], 1.0 / got[:, 2]], 1), psi_sorted, amp[order_l][np.argsort(np.argsort(tables[3]))])
    assert np.abs(got - exp_xyz).max() <= 1e-5 * np.abs(exp_xyz - xyz0).max()
    print(f"P={P}: recorded graph ({len(edges)} observation edges, {len(cons)} pose-pose edges) -> HIP vs oracle: poses {np.abs(poses_hip - poses_ref).max():.1e} "
          f"(update {upd_p:.1e}); HIP branch in place: poses {np.abs(hip['poses_out'] - exp_poses).max():.1e}")
