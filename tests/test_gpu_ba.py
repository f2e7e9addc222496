"""GPU parity of the double-window BA (SlamGraph::optimize) against the CPU oracle, via the C ABI.

Bar (BASELINE.json north_star): pose-graph state update within 1e-6 relative of the CPU path.
f64 everywhere; differences come only from summation order (atomics / wave reductions).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cam(c):
    from scavislam_amd.ctypes_types import Cam
    return Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])


def _rel_update_err(new, ref, start):
    upd = np.abs(ref - start).max()
    return np.abs(new - ref).max() / max(upd, 1e-300)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("P,L,n_outer", [(15, 3000, 3), (6, 200, 0)])
def test_reduced_system_matches_oracle(gpu_ctx, P, L, n_outer, mode):
    """One Schur step: the packed reduced camera system (H_schur + lambda I, b_schur) and chi2 built by
    the landmark kernel equal the oracle's dense construction to 1e-10 of the matrix scale, in both
    self-edge modes (G2O_LITERAL / EXACT)."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.ba_window(P, L, seed=2012, n_outer=n_outer)
    cam = _cam(prob["cam"])
    prm = BaParams.reference_defaults()
    prm.self_edge_mode = mode
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    for lam in (50.0, 0.7):
        H, b, chi2 = opt.reduced_system(lam)
        H_ref, b_ref = O.ba_reduced_system(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm, lam)
        chi2_ref = O.ba_chi2(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        assert np.allclose(H, H.T)
        np.testing.assert_allclose(H, H_ref, rtol=0, atol=1e-10 * np.abs(H_ref).max())
        np.testing.assert_allclose(b, b_ref, rtol=0, atol=1e-10 * np.abs(b_ref).max())
        np.testing.assert_allclose(chi2, chi2_ref, rtol=1e-12)
    # the two modes must differ exactly by the spurious +M on anchor diagonals (B-7)
    opt.close()


@pytest.mark.parametrize("P,L,n_outer,seed", [(15, 3000, 3, 2012), (8, 300, 2, 5), (4, 50, 0, 6)])
def test_optimize_matches_oracle(gpu_ctx, P, L, n_outer, seed):
    """SlamGraph::optimize(OptParams(2,true,3)): same LM trajectory (trials/accepts/termination) and
    final poses + points within 1e-6 relative of the applied update."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.ba_window(P, L, seed=seed, n_outer=n_outer)
    cam = _cam(prob["cam"])
    prm = BaParams.reference_defaults()
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    st = opt.optimize()
    poses, psi = opt.restoreDataFromG2o()
    poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    assert (st.iterations, st.trials, st.accepted, st.terminated) == \
        (st_ref.iterations, st_ref.trials, st_ref.accepted, st_ref.terminated)
    np.testing.assert_allclose(st.chi2_init, st_ref.chi2_init, rtol=1e-12)
    np.testing.assert_allclose(st.chi2_final, st_ref.chi2_final, rtol=1e-9)
    np.testing.assert_allclose(st.lambda_final, st_ref.lambda_final, rtol=1e-6)
    assert st_ref.accepted >= 1 and st_ref.chi2_final < st_ref.chi2_init
    assert _rel_update_err(poses, poses_ref, prob["poses"]) < 1e-6
    assert _rel_update_err(psi, psi_ref, prob["psi"]) < 1e-6
    opt.close()


def test_optimize_rejected_steps_and_termination(gpu_ctx):
    """A window where the first trials are rejected (lambda_init tiny, poses badly perturbed): exercises
    the reject / lambda*nu path and the Terminate-after-max-trials path; must follow the oracle."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.ba_window(8, 300, seed=77, n_outer=0, pose_sigma_t=0.4, pose_sigma_r_deg=8.0, outlier_frac=0.2)
    cam = _cam(prob["cam"])
    seen_reject = False
    for lam0, robust in ((1e-9, 0), (1e-6, 1), (50.0, 1)):
        prm = BaParams.reference_defaults()
        prm.lambda_init, prm.use_robust, prm.num_iters = lam0, robust, 3
        opt = SlamGraphOptimizer(ctx, stream)
        opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        st = opt.optimize()
        poses, psi = opt.restoreDataFromG2o()
        poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        assert (st.iterations, st.trials, st.accepted, st.terminated) == \
            (st_ref.iterations, st_ref.trials, st_ref.accepted, st_ref.terminated)
        seen_reject |= st_ref.trials > st_ref.accepted
        if st_ref.accepted:
            assert _rel_update_err(poses, poses_ref, prob["poses"]) < 1e-6
            assert _rel_update_err(psi, psi_ref, prob["psi"]) < 1e-6
        else:
            assert np.array_equal(poses, prob["poses"])
        opt.close()
    assert seen_reject, "test is meant to exercise rejected LM trials"


def test_optimize_full_size_properties(gpu_ctx):
    """BASELINE size (50 KF / 20k landmarks, ~100k edges): too slow to diff against a Python model,
    so check size-independent properties: (1) the solution of the reduced system satisfies the FULL
    normal equations through the oracle's chi2 (cost decreases exactly as reported), (2) sharding the
    landmarks over 2, 4 and 8 pseudo-ranks (8 = BASELINE configs[3]'s own shape: 313 chunks of 64 landmarks dealt
    to 8 shards) and summing the partial reduced systems reproduces the single-GPU system to 1e-10 (linearity of
    the Schur reduction over landmarks)."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer, shard_problem
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.ba_window(50, 20000, seed=2012)
    cam = _cam(prob["cam"])
    prm = BaParams.reference_defaults()
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    H, b, chi2 = opt.reduced_system(50.0)
    n = H.shape[0]
    for world in (2, 4, 8):
        Hs, bs, cs = np.zeros_like(H), np.zeros_like(b), 0.0
        for r in range(world):
            sh = shard_problem(prob, r, world)
            o = SlamGraphOptimizer(ctx, stream)
            o.copyDataToG2o(sh["poses"], sh["psi"], sh["edges"], sh["cons"], cam, prm, add_pose_terms=sh["add_pose_terms"])
            Hr, br, cr = o.reduced_system(50.0)
            Hs += Hr - 50.0 * np.eye(n)      # expand adds lambda I per call
            bs += br
            cs += cr
            o.close()
        Hs += 50.0 * np.eye(n)
        np.testing.assert_allclose(Hs, H, rtol=0, atol=1e-10 * np.abs(H).max())
        np.testing.assert_allclose(bs, b, rtol=0, atol=1e-10 * np.abs(b).max())
        np.testing.assert_allclose(cs, chi2, rtol=1e-12)
    st = opt.optimize()
    poses, psi = opt.restoreDataFromG2o()
    chi2_after = O.ba_chi2(poses, psi, prob["edges"], prob["cons"], cam, prm)
    np.testing.assert_allclose(chi2_after, st.chi2_final, rtol=1e-9)
    assert st.chi2_final < 0.5 * st.chi2_init and st.accepted >= 1
    # and against the oracle's full optimize (a few hundred ms on the CPU)
    poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    assert _rel_update_err(poses, poses_ref, prob["poses"]) < 1e-6
    assert _rel_update_err(psi, psi_ref, prob["psi"]) < 1e-6
    opt.close()


@pytest.mark.parametrize("world,n_kf,n_pts,seed", [(2, 10, 800, 4), (8, 50, 20000, 2012)])
def test_sharded_optimize_single_process(gpu_ctx, world, n_kf, n_pts, seed):
    """The all-reduce hook: run `world` landmark shards on ONE GPU with a callback that sums the
    shards' buffers, and compare with the unsharded optimize (the N>1 control flow without RCCL;
    the real multi-process path is covered by tests/test_dist_gloo.py on CPU).  world = 8 at 50 KF / 20k
    landmarks is BASELINE configs[3] as named: eight shards of the inner window, every pseudo-rank's update
    held to 1e-6 of the one-GPU update."""
    import ctypes as C
    import threading
    import torch
    from scavislam_amd import capi, synth
    from scavislam_amd.backend import SlamGraphOptimizer, shard_problem, _as_tensor
    from scavislam_amd.ctypes_types import BaParams
    ctx0, stream0 = gpu_ctx
    prob = synth.ba_window(n_kf, n_pts, seed=seed)
    cam = _cam(prob["cam"])
    prm = BaParams.reference_defaults()
    base = SlamGraphOptimizer(ctx0, stream0)
    base.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    st0 = base.optimize()
    poses0, psi0 = base.restoreDataFromG2o()
    ctxs = [capi.torch_context(0) for _ in range(world)]
    barrier = threading.Barrier(world, timeout=120)
    slots = [None] * world
    results = [None] * world

    def run(rank):
        ctx, stream = ctxs[rank]
        sh = shard_problem(prob, rank, world)
        opt = SlamGraphOptimizer(ctx, stream)
        opt.copyDataToG2o(sh["poses"], sh["psi"], sh["edges"], sh["cons"], cam, prm, add_pose_terms=sh["add_pose_terms"])

        def allreduce(d_buf, count, _u):
            ctx.sync()
            slots[rank] = _as_tensor(d_buf, count, 0)
            barrier.wait()
            total = slots[0].clone()
            for r in range(1, world):
                total += slots[r]
            torch.cuda.synchronize()
            barrier.wait()
            slots[rank].copy_(total)
            torch.cuda.synchronize()
            barrier.wait()
            return 0
        cb = capi.ALLREDUCE_FN(allreduce)
        st = opt.optimize(cb)
        results[rank] = (st, *opt.restoreDataFromG2o(), sh["owner"])
        opt.close()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    for r in range(world):
        st, poses, psi, owner = results[r]
        assert (st.trials, st.accepted) == (st0.trials, st0.accepted)
        assert _rel_update_err(poses, poses0, prob["poses"]) < 1e-6
        mine = owner == r
        assert _rel_update_err(psi[mine], psi0[mine], prob["psi"][mine]) < 1e-6
    for c, _ in ctxs:
        c.close()


def test_speculative_lm_equals_host_driven_lm(gpu_ctx, monkeypatch):
    """The device-side accept/reject path (all iterations enqueued at once, ba_lm_kernel decides) and the
    host-driven loop (option no_speculation: one synchronisation per trial) are the same arithmetic in
    the same order up to the order of the f64 atomics (not reproducible run to run either): identical LM
    trajectory, states and statistics equal to ~1e-8, also when trials are rejected mid-way."""
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.ba_window(8, 300, seed=77, n_outer=0, pose_sigma_t=0.4, pose_sigma_r_deg=8.0, outlier_frac=0.2)
    cam = _cam(prob["cam"])
    for lam0, iters in ((1e-9, 3), (1e-3, 6), (50.0, 4)):
        out = []
        for no_spec in (False, True):
            prm = BaParams.reference_defaults()
            prm.lambda_init, prm.num_iters = lam0, iters
            opt = SlamGraphOptimizer(ctx, stream)
            opt.set_option("no_speculation", int(no_spec))
            opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
            st = opt.optimize()
            out.append((st, *opt.restoreDataFromG2o()))
            opt.close()
        (a, pa, sa), (b, pb, sb) = out
        assert (a.iterations, a.trials, a.accepted, a.terminated) == (b.iterations, b.trials, b.accepted, b.terminated)
        for x, y in ((a.chi2_init, b.chi2_init), (a.chi2_final, b.chi2_final), (a.lambda_final, b.lambda_final)):
            assert abs(x - y) <= 1e-7 * abs(y)
        if a.accepted:
            assert _rel_update_err(pa, pb, prob["poses"]) < 1e-7 and _rel_update_err(sa, sb, prob["psi"]) < 1e-7
        else:
            assert np.array_equal(pa, pb) and np.array_equal(sa, sb)


def _with_loop_closure(prob, i, j, rng):
    """Adds a relative-pose constraint between the far-apart poses i < j (a loop-closure style edge, slam_graph.cpp:937-981
    builds the same kind for outer-window poses): the block envelope of the reduced system then spans j - i + 1 rows."""
    from scavislam_amd import synth
    from scavislam_amd.ctypes_types import BA_CONSTRAINT_DTYPE
    gt = prob["poses_gt"].reshape(-1, 3, 4)
    T_ji = synth.pose_mul(gt[j], synth.pose_inv(gt[i]))
    dn = np.concatenate([rng.normal(0, 0.005, 3), rng.normal(0, 0.002, 3)])
    T_ji = synth.pose_mul(synth.pose(synth.so3_exp(dn[3:]), dn[:3]), T_ji)
    Lam = np.eye(6) * 30.0
    Lam[:3, :3] *= (350 * np.linalg.norm(T_ji[:, 3]) / 8.0) ** 2
    Lam[3:, 3:] *= 100.0 ** 2
    c = np.zeros(1, BA_CONSTRAINT_DTYPE)
    c[0]["T_21"], c[0]["info"], c[0]["pose1"], c[0]["pose2"] = T_ji.reshape(12), Lam.reshape(36), i, j
    return dict(prob, cons=np.concatenate([prob["cons"], c]))


@pytest.mark.parametrize("case", ["fused_two_fronts", "fused_one_front", "lds_forced", "global_forced", "lds_wide_envelope",
                                  "global_wide_envelope", "grid_wide_envelope", "grid_p45_wide_envelope", "grid_rows_wide_envelope", "ordered_loop_closure"])
def test_every_solve_variant_matches_oracle(gpu_ctx, monkeypatch, case):
    """The reduced camera system is solved by one of five kernels depending on the width of its block envelope: the fused
    register-resident elimination (<= 10 block rows; two fronts when the window is long enough), the LDS-window pipeline
    (<= 28 and it fits LDS), the multi-workgroup tile-resident blocked Cholesky (wide envelopes; round 5) and the multi-workgroup
    Cholesky by block rows it replaced (option "no_tile_solve"), the one-workgroup global-memory blocked Cholesky (narrow envelopes
    with the LDS variants switched off, or on request).  Each must give the oracle's update to 1e-6."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    rng = np.random.default_rng(3)
    P = 45 if case == "grid_p45_wide_envelope" else 40      # (45 poses: the last 24 x 24 tile row of the tile-resident solve is padded by three identity blocks)
    prob = synth.ba_window(P, 4000, seed=11, n_outer=2)
    options = {"fused_one_front": "one_front", "lds_forced": "no_fused_solve", "global_forced": "no_lds_solve", "global_wide_envelope": "no_grid_solve",
               "grid_rows_wide_envelope": "no_tile_solve"}
    if case == "lds_wide_envelope":
        prob = _with_loop_closure(prob, 5, 27, rng)          # envelope of 23 block rows: too wide for the LDS window at P = 40 (194 KB), one workgroup
    elif case in ("global_wide_envelope", "grid_wide_envelope", "grid_p45_wide_envelope", "grid_rows_wide_envelope", "ordered_loop_closure"):
        prob = _with_loop_closure(prob, 0, P - 1, rng)       # full envelope: 40 block rows (one workgroup / spread over several)
    cam = _cam(prob["cam"])
    prm = BaParams.reference_defaults()
    opt = SlamGraphOptimizer(ctx, stream)
    if case in options:
        opt.set_option(options[case], 1)
    if case.endswith("wide_envelope"):
        opt.set_option("no_order", 1)                        # these cases are about the kernels for wide envelopes: keep the caller's pose order
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    expect = {"fused_two_fronts": "two fronts", "fused_one_front": "one front", "lds_forced": "LDS-window", "global_forced": "global-memory",
              "lds_wide_envelope": "global-memory", "global_wide_envelope": "global-memory", "grid_wide_envelope": "multi-workgroup tile-resident",
              "grid_p45_wide_envelope": "multi-workgroup tile-resident", "grid_rows_wide_envelope": "multi-workgroup blocked",
              "ordered_loop_closure": "LDS-window"}[case]      # (the same loop closure in the fill-reducing order: 40 -> 15 block rows, back in LDS)
    assert expect in opt.info()["solve_kernel"], opt.info()
    if case == "ordered_loop_closure":
        assert opt.info()["pose_order"] == "reverse Cuthill-McKee" and opt.info()["envelope_rows"] <= 20, opt.info()
    st = opt.optimize()
    poses, psi = opt.restoreDataFromG2o()
    poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    assert (st.iterations, st.trials, st.accepted, st.terminated) == (st_ref.iterations, st_ref.trials, st_ref.accepted, st_ref.terminated)
    assert st_ref.accepted >= 1
    np.testing.assert_allclose(st.chi2_final, st_ref.chi2_final, rtol=1e-8)      # wide envelopes: longer elimination chains
    assert _rel_update_err(poses, poses_ref, prob["poses"]) < 1e-6               # the parity bar (measured: 2e-9 .. 2e-8)
    # landmarks: the same 1e-6 bar, plus what the back-substitution x_l = D^-1 (b_l - sum W_i^T x_i) makes of the pose update's own
    # error: |dx_l| <= a_l |dx_p| with a_l = |D^-1| |W| (np_model.landmark_amplification; numeric Jacobians at the start state).  The
    # pose update differs from the oracle's by ~1e-9 relative (order of the f64 atomics); a weak-parallax landmark (tiny smallest
    # eigenvalue of H_ll) multiplies that by several hundred.  So every landmark must satisfy
    #     |psi - psi_ref|_l <= 1e-6 max|update| + 10 a_l max|poses - poses_ref|
    # and the test reports for how many the second term was needed at all.
    import np_model as M
    c = prob["cam"]
    amp = M.landmark_amplification(prob["poses"], prob["psi"], prob["edges"], (c["f"], c["cx"], c["cy"], c["b"]), st_ref.lambda_final)
    upd = np.abs(psi_ref - prob["psi"]).max()
    err_pose = np.abs(poses - poses_ref).max()
    err_l = np.abs(psi - psi_ref).max(1)
    assert (err_l <= 1e-6 * upd + 10.0 * amp * err_pose).all(), float((err_l - 10.0 * amp * err_pose).max() / upd)
    n_over = int((err_l > 1e-6 * upd).sum())
    print(f"{case}: {n_over} of {int((amp > 0).sum())} landmarks above the plain 1e-6 bar (all explained by pose error x amplification)")
    assert n_over <= 2      # measured: 0 of 4 000 in nine variants, 0 .. 1 in the tenth (the order of the f64 atomics differs from run to run); was <= 8
    opt.close()


def test_landmarks_with_64_observations(gpu_ctx):
    """Landmarks seen by 64 keyframes (one landmark fills a whole wave chunk, 32 circulant pair rounds) next to ordinary ones;
    anchors in the middle of their runs as well as at the start.  The reduced system spans 64 block rows, so this also runs
    the global-memory solve on a real (not forced) wide envelope.  A 65th observation switches that landmark to the
    one-workgroup-per-landmark kernel (ba_wide_landmark_kernel)."""
    import oracle as O
    from scavislam_amd import capi, synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BA_EDGE_DTYPE, BaParams
    ctx, stream = gpu_ctx
    P = 70
    prob = synth.ba_window(P, 600, seed=31, n_outer=0)
    c = prob["cam"]
    gt = prob["poses_gt"].reshape(-1, 3, 4)
    rng = np.random.default_rng(9)
    e = prob["edges"]
    extra = []
    for l, (first, anchor) in zip((5, 77, 300), ((0, 0), (3, 40), (6, 69))):        # runs [first, first + 64), anchor inside
        e = e[e["point"] != l]
        pg = prob["psi_gt"][l]
        xa = np.array([pg[0] / pg[2], pg[1] / pg[2], 1.0 / pg[2]])
        Tw = synth.pose_inv(gt[anchor])
        xw = Tw[:, :3] @ xa + Tw[:, 3]
        for i in range(first, first + 64):
            y = gt[i][:, :3] @ xw + gt[i][:, 3]
            obs = np.array([c["f"] * y[0] / y[2] + c["cx"], c["f"] * y[1] / y[2] + c["cy"], c["f"] * (y[0] - c["b"]) / y[2] + c["cx"]])
            rec = np.zeros(1, BA_EDGE_DTYPE)
            rec["obs"], rec["info"], rec["point"], rec["pose"], rec["anchor"] = obs + rng.normal(0, 0.5, 3), (1.0, 1.0, 0.333 ** 2), l, i, anchor
            extra.append(rec)
    edges = np.concatenate([e] + extra)
    rng.shuffle(edges)                                                               # the reference iterates hash sets: any input order
    assert np.bincount(edges["point"]).max() == 64
    cam = _cam(c)
    prm = BaParams.reference_defaults()
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(prob["poses"], prob["psi"], edges, prob["cons"], cam, prm)
    H, b, chi2 = opt.reduced_system(50.0)
    H_ref, b_ref = O.ba_reduced_system(prob["poses"], prob["psi"], edges, prob["cons"], cam, prm, 50.0)
    np.testing.assert_allclose(H, H_ref, rtol=0, atol=1e-10 * np.abs(H_ref).max())
    np.testing.assert_allclose(b, b_ref, rtol=0, atol=1e-10 * np.abs(b_ref).max())
    st = opt.optimize()
    poses, psi = opt.restoreDataFromG2o()
    poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], edges, prob["cons"], cam, prm)
    assert (st.iterations, st.trials, st.accepted, st.terminated) == (st_ref.iterations, st_ref.trials, st_ref.accepted, st_ref.terminated)
    assert _rel_update_err(poses, poses_ref, prob["poses"]) < 1e-6
    assert _rel_update_err(psi, psi_ref, prob["psi"]) < 1e-6
    # a 65th observation moves landmark 5 from the wave-chunk kernel to the wide-landmark kernel: still the oracle's result
    one_more = edges[edges["point"] == 5][:1].copy()
    one_more["pose"] = 64
    y = gt[64][:, :3] @ (synth.pose_inv(gt[int(one_more["anchor"][0])])[:, :3] @ np.array([prob["psi_gt"][5][0], prob["psi_gt"][5][1], 1.0]) / prob["psi_gt"][5][2]
                          + synth.pose_inv(gt[int(one_more["anchor"][0])])[:, 3]) + gt[64][:, 3]
    one_more["obs"] = [c["f"] * y[0] / y[2] + c["cx"], c["f"] * y[1] / y[2] + c["cy"], c["f"] * (y[0] - c["b"]) / y[2] + c["cx"]]
    edges65 = np.concatenate([edges, one_more])
    opt.copyDataToG2o(prob["poses"], prob["psi"], edges65, prob["cons"], cam, prm)
    st = opt.optimize()
    poses, psi = opt.restoreDataFromG2o()
    poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], edges65, prob["cons"], cam, prm)
    assert (st.trials, st.accepted) == (st_ref.trials, st_ref.accepted)
    assert _rel_update_err(poses, poses_ref, prob["poses"]) < 1e-6 and _rel_update_err(psi, psi_ref, prob["psi"]) < 1e-6
    opt.close()


@pytest.mark.parametrize("tiles", [True, False])
def test_double_window_230_poses_wide_landmarks(gpu_ctx, tiles):
    """The reference's real window shape (data/newcollege.cfg:21-22, backend.cpp:141-144): 30 inner + 200 outer poses, all free
    (slam_graph.cpp:932), ~400 pose-pose constraints incl. two loop closures (slam_graph.cpp:937-981), active points observed from
    every window pose that sees them -- among them landmarks with 100 and 180 observations (no cap in slam_graph.cpp:1001-1027).
    Reduced system <= 1e-10 and optimize <= 1e-6 vs the oracle; the wide envelope (224 block rows) lands on the multi-workgroup solves: the tile-resident blocked Cholesky and, on request, the one by block rows."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.double_window(n_inner=30, n_outer=200, L=6000, seed=21, n_long=(100, 180, 70), n_loops=2)
    assert len(prob["poses"]) == 230 and 350 <= len(prob["cons"]) <= 450
    counts = np.bincount(prob["edges"]["point"], minlength=len(prob["psi"]))
    assert counts.max() == 180 and (counts > 64).sum() == 3
    cam = _cam(prob["cam"])
    prm = BaParams.reference_defaults()
    opt = SlamGraphOptimizer(ctx, stream)
    if not tiles:
        opt.set_option("no_tile_solve", 1)                   # the multi-workgroup Cholesky by block rows (rounds 3-4) instead of the tile-resident one
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    assert ("tile-resident" in opt.info()["solve_kernel"]) == tiles, opt.info()
    H, b, chi2 = opt.reduced_system(50.0)
    H_ref, b_ref = O.ba_reduced_system(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm, 50.0)
    np.testing.assert_allclose(H, H_ref, rtol=0, atol=1e-10 * np.abs(H_ref).max())
    np.testing.assert_allclose(b, b_ref, rtol=0, atol=1e-10 * np.abs(b_ref).max())
    st = opt.optimize()
    poses, psi = opt.restoreDataFromG2o()
    poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    assert (st.iterations, st.trials, st.accepted, st.terminated) == (st_ref.iterations, st_ref.trials, st_ref.accepted, st_ref.terminated)
    assert st_ref.accepted >= 1
    assert _rel_update_err(poses, poses_ref, prob["poses"]) < 1e-6
    assert _rel_update_err(psi, psi_ref, prob["psi"]) < 1e-6
    opt.close()


def test_loop_closure_window_solved_in_fill_reducing_order(gpu_ctx):
    """VERDICT round 3, missing 4: the reference's solver orders the pose blocks before it factorises (slam_graph.cpp:1063-1074 -> LinearSolverCSparse); in the
    caller's order two loop-closure constraints on a chain of 230 keyframes make the FILLED envelope span the whole chain (224 block rows).  The library now takes
    a reverse Cuthill-McKee order of the pose block graph when it narrows the envelope by a quarter or more, solves a permuted copy of the reduced system and
    scatters x / the trial poses back.  Held: the order is a permutation with envelope <= 48, the result equals the oracle's (1e-6 of the update, same LM
    trajectory) and the result in the caller's order ("no_order", the multi-workgroup Cholesky on the 224-row envelope) to 1e-9."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.double_window(n_inner=30, n_outer=200, L=6000, seed=21, n_long=(), n_loops=2)
    cam = _cam(prob["cam"])
    prm = BaParams.reference_defaults()
    poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    out = {}
    for no_order in (0, 1):
        opt = SlamGraphOptimizer(ctx, stream)
        opt.set_option("no_order", no_order)
        opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        info = opt.info()
        if not no_order:
            perm = opt.pose_order()
            assert sorted(perm.tolist()) == list(range(230)) and not np.array_equal(perm, np.arange(230))
            assert info["pose_order"] == "reverse Cuthill-McKee" and info["envelope_rows"] <= 48 and info["envelope_rows_callers_order"] >= 200, info
        else:
            assert info["pose_order"] == "caller's" and info["envelope_rows"] >= 200 and "multi-workgroup" in info["solve_kernel"], info
        st = opt.optimize()
        poses, psi = opt.restoreDataFromG2o()
        assert (st.iterations, st.trials, st.accepted, st.terminated) == (st_ref.iterations, st_ref.trials, st_ref.accepted, st_ref.terminated)
        assert _rel_update_err(poses, poses_ref, prob["poses"]) < 1e-6 and _rel_update_err(psi, psi_ref, prob["psi"]) < 1e-6
        out[no_order] = (poses, psi, info)
        opt.close()
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-9 * np.abs(poses_ref - prob["poses"]).max()
    print("loop-closure window:", out[0][2], "| caller's order:", out[1][2]["solve_kernel"], out[1][2]["envelope_rows"])


def test_many_chunks_two_workgroups_per_cu(gpu_ctx):
    """More wave chunks than 8 x the CUs: the Schur pass runs four waves per workgroup, two workgroups per CU, with the SMALL LDS pool
    (one window of at most 16 poses).  (a) a plain 100-keyframe / 40 000-landmark inner window; (b) the 230-pose double window with
    30 000 landmarks, whose long tracks and loop closures span more poses than the window holds (clamped window, the rest by global
    atomics) and whose 100- and 180-observation landmarks take the wide-landmark kernel."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prm = BaParams.reference_defaults()
    cases = [synth.ba_window(100, 40000, seed=31), synth.double_window(n_inner=30, n_outer=200, L=30000, seed=5, n_long=(100, 180), n_loops=2)]
    for prob in cases:
        cam = _cam(prob["cam"])
        opt = SlamGraphOptimizer(ctx, stream)
        opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        assert opt.info()["wave_chunks"] > 8 * 256
        H, b, chi2 = opt.reduced_system(50.0)
        H_ref, b_ref = O.ba_reduced_system(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm, 50.0)
        np.testing.assert_allclose(H, H_ref, rtol=0, atol=1e-10 * np.abs(H_ref).max())
        np.testing.assert_allclose(b, b_ref, rtol=0, atol=1e-10 * np.abs(b_ref).max())
        st = opt.optimize()
        poses, psi = opt.restoreDataFromG2o()
        poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        assert (st.iterations, st.trials, st.accepted, st.terminated) == (st_ref.iterations, st_ref.trials, st_ref.accepted, st_ref.terminated)
        assert _rel_update_err(poses, poses_ref, prob["poses"]) < 1e-6
        opt.close()


def test_largest_supported_window_256_poses(gpu_ctx):
    """P = 256 keyframes (the documented maximum of the single-workgroup solves): long two-front elimination
    (front 1 takes ~120 block rows), everything else as usual; 257 poses are refused with SVS_ERR_UNSUPPORTED."""
    import oracle as O
    from scavislam_amd import capi, synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.ba_window(256, 6000, seed=77, n_outer=3)
    cam = _cam(prob["cam"])
    prm = BaParams.reference_defaults()
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    st = opt.optimize()
    poses, psi = opt.restoreDataFromG2o()
    poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    assert (st.iterations, st.trials, st.accepted, st.terminated) == (st_ref.iterations, st_ref.trials, st_ref.accepted, st_ref.terminated)
    assert st_ref.accepted >= 1
    assert _rel_update_err(poses, poses_ref, prob["poses"]) < 1e-6
    assert _rel_update_err(psi, psi_ref, prob["psi"]) < 1e-6
    with pytest.raises(capi.SvsError, match="status 5"):
        opt.copyDataToG2o(np.tile(prob["poses"][:1], (257, 1)), prob["psi"], prob["edges"], prob["cons"], cam, prm)
    opt.close()


def test_kernel_timing_brackets_are_opt_in(gpu_ctx):
    """svs_ba_set_timing: the hipEvent brackets around the three big kernels are off by default (kernel_times() = 0) and
    report plausible per-trial durations when switched on; results do not depend on them."""
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.ba_window(15, 3000, seed=2012)
    cam = _cam(prob["cam"])
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, BaParams.reference_defaults())
    st0 = opt.optimize()
    kt = opt.kernel_times()
    assert kt["reduce_ms"] == 0 and kt["solve_ms"] == 0 and kt["backsub_ms"] == 0
    p0, s0 = opt.restoreDataFromG2o()
    opt.reset_state(prob["poses"], prob["psi"])
    opt.set_timing(True)
    st1 = opt.optimize()
    kt = opt.kernel_times()
    assert kt["n_trials"] == st1.trials == st0.trials
    for k in ("reduce_ms", "solve_ms", "backsub_ms"):
        assert 0.001 < kt[k] / kt["n_trials"] < 5.0, (k, kt)
    p1, s1 = opt.restoreDataFromG2o()
    assert _rel_update_err(p1, p0, prob["poses"]) < 1e-9
    opt.set_timing(False)
    opt.close()


def test_library_communicator_world_1(gpu_ctx):
    """The library-owned RCCL communicator (svs_comm_*, comm.hip): created from a unique id, attached with svs_ba_set_comm, used by
    svs_ba_optimize for every exchange of the sharded path.  One rank = the collectives are identities, so the result must equal the
    plain single-GPU optimize bit-close, and the communicator must have carried 1 pattern + 2 x (system + scalars) all-reduces."""
    import torch
    from scavislam_amd import synth
    from scavislam_amd.backend import Communicator, SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.ba_window(10, 800, seed=4)
    cam = _cam(prob["cam"])
    prm = BaParams.reference_defaults()
    base = SlamGraphOptimizer(ctx, stream)
    base.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    st0 = base.optimize()
    poses0, psi0 = base.restoreDataFromG2o()
    comm = Communicator(ctx, 0, 1)
    # the collective itself: in place, on the ctx stream
    with torch.cuda.stream(stream):
        t = torch.arange(1000, dtype=torch.float64, device="cuda")
    comm.allreduce(t.data_ptr(), t.numel())
    ctx.sync()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64))
    n0 = comm.stats()["n_calls"]
    opt = SlamGraphOptimizer(ctx, stream)
    opt.set_comm(comm)
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    st = opt.optimize()
    poses, psi = opt.restoreDataFromG2o()
    s = comm.stats()
    assert (s["rank"], s["world"]) == (0, 1)
    assert s["n_calls"] - n0 == 1 + 2 * st.trials
    assert (st.trials, st.accepted) == (st0.trials, st0.accepted)
    assert _rel_update_err(poses, poses0, prob["poses"]) < 1e-7 and _rel_update_err(psi, psi0, prob["psi"]) < 1e-7
    opt.close(); base.close(); comm.close()


def test_failed_set_problem_invalidates_the_handle(gpu_ctx):
    """A svs_ba_set_problem that fails half-way (here: an out-of-range edge index after a valid smaller window) must leave the handle
    unusable -- optimize / get_state / reduced_system refuse -- until a later call completes; no stale sizes next to old buffers."""
    from scavislam_amd import capi, synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    small = synth.ba_window(8, 200, seed=3)
    big = synth.ba_window(60, 900, seed=5)
    cam = _cam(small["cam"])
    prm = BaParams.reference_defaults()
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(small["poses"], small["psi"], small["edges"], small["cons"], cam, prm)
    opt.optimize()
    bad = big["edges"].copy()
    bad["pose"][len(bad) // 2] = 60                    # out of range
    with pytest.raises(capi.SvsError):
        opt.copyDataToG2o(big["poses"], big["psi"], bad, big["cons"], cam, prm)
    for call in (opt.optimize, opt.restoreDataFromG2o, lambda: opt.reduced_system(50.0)):
        with pytest.raises(capi.SvsError):
            call()
    opt.copyDataToG2o(big["poses"], big["psi"], big["edges"], big["cons"], cam, prm)      # a completed call revives it
    st = opt.optimize()
    assert st.trials >= 1
    opt.close()


def test_optimize_batch_of_windows(gpu_ctx):
    """svs_ba_optimize_batch: six different windows (sizes 6..40 keyframes, one of them with rejected first trials), each on its own
    context, enqueued together: every window ends exactly where its stand-alone optimize ends (same statistics, state to 1e-9)."""
    from scavislam_amd import capi, synth
    from scavislam_amd.backend import SlamGraphOptimizer, optimize_batch
    from scavislam_amd.ctypes_types import BaParams
    ctx0, stream0 = gpu_ctx
    specs = [(6, 200, 1, {}), (15, 3000, 2, {}), (40, 4000, 3, {}), (8, 300, 77, dict(pose_sigma_t=0.4, pose_sigma_r_deg=8.0, outlier_frac=0.2)),
             (25, 2500, 5, {}), (12, 900, 6, {})]
    probs = [synth.ba_window(P, L, seed=sd, n_outer=2 if P > 8 else 0, **kw) for P, L, sd, kw in specs]
    prms = []
    for i in range(len(probs)):
        prm = BaParams.reference_defaults()
        if i == 3:
            prm.lambda_init, prm.use_robust, prm.num_iters = 1e-9, 0, 3      # rejected trials: the host-driven remainder runs inside the batch
        prms.append(prm)
    ref = []
    for prob, prm in zip(probs, prms):
        o = SlamGraphOptimizer(ctx0, stream0)
        o.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], _cam(prob["cam"]), prm)
        st = o.optimize()
        ref.append((st, *o.restoreDataFromG2o()))
        o.close()
    ctxs = [capi.torch_context(0) for _ in probs]
    opts = []
    for (c, s), prob, prm in zip(ctxs, probs, prms):
        o = SlamGraphOptimizer(c, s)
        o.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], _cam(prob["cam"]), prm)
        opts.append(o)
    sts = optimize_batch(opts)
    saw_reject = False
    for o, st, (st0, p0, s0), prob in zip(opts, sts, ref, probs):
        assert (st.iterations, st.trials, st.accepted, st.terminated) == (st0.iterations, st0.trials, st0.accepted, st0.terminated)
        saw_reject |= st.trials > st.accepted
        poses, psi = o.restoreDataFromG2o()
        if st0.accepted:
            assert _rel_update_err(poses, p0, prob["poses"]) < 1e-7 and _rel_update_err(psi, s0, prob["psi"]) < 1e-7
        o.close()
    assert saw_reject
    for c, _ in ctxs:
        c.close()


def test_persistent_window_equals_rebuilt_window(gpu_ctx):
    """svs_ba_window_update: a window of 30 keyframes slides over a 48-keyframe sequence, one keyframe per step.
    Every step hands over ONLY the newest keyframe's observations (+ ids, current values, constraints); the edge list is assembled on
    the device from the stored observations.  Each step's optimize must equal -- statistics exactly, state to 1e-9 of the update --
    the optimize of the same window handed over completely through svs_ba_set_problem (copyDataToG2o's form)."""
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BA_CONSTRAINT_DTYPE, BaParams
    ctx, stream = gpu_ctx
    Pall, W = 48, 30
    uni = synth.ba_window(Pall, 4000, seed=91, n_outer=0)
    cam = _cam(uni["cam"])
    prm = BaParams.reference_defaults()
    pose_id = 1000 + 3 * np.arange(Pall)                          # the graph's own ids: sparse, not window indices
    point_id = 5 + 7 * np.arange(len(uni["psi"]))
    edges = uni["edges"]
    anchor_of = np.full(len(uni["psi"]), -1)
    anchor_of[edges["point"]] = edges["anchor"]
    poses = uni["poses"].copy()
    psi = uni["psi"].copy()
    inc = SlamGraphOptimizer(ctx, stream)
    full = SlamGraphOptimizer(ctx, stream)
    gt = uni["poses_gt"].reshape(-1, 3, 4)
    given = np.zeros(len(edges), bool)
    for step, last in enumerate(range(W - 1, Pall)):
        win = np.arange(last - W + 1, last + 1)                   # keyframes of the window, oldest first
        in_win = np.zeros(Pall, bool); in_win[win] = True
        e_ok = in_win[edges["pose"]] & in_win[edges["anchor"]]
        active = np.unique(edges["point"][e_ok])                  # points seen from the window whose anchor is in it
        new = e_ok & ~given if step else e_ok.copy()
        if step == 0:                                             # hand over a few observations that are NOT (yet) in the window, too
            new |= (edges["pose"] == last + 1)
        else:
            new = (edges["pose"] == last + 1) & ~given if last + 1 < Pall else np.zeros(len(edges), bool)
            new |= e_ok & ~given
        given |= new
        new_obs = edges[new].copy()
        new_obs["point"] = point_id[edges["point"][new]]
        new_obs["pose"] = pose_id[edges["pose"][new]]
        new_obs["anchor"] = -123                                   # ignored
        # two relative-pose constraints between window poses (by frame id)
        cons = np.zeros(2, BA_CONSTRAINT_DTYPE)
        for k, (i, j) in enumerate(((win[0], win[1]), (win[2], win[4]))):
            cons[k]["T_21"] = synth.pose_mul(gt[j], synth.pose_inv(gt[i])).reshape(12)
            cons[k]["info"] = (np.eye(6) * 200.0).reshape(36)
            cons[k]["pose1"], cons[k]["pose2"] = pose_id[i], pose_id[j]
        inc.windowUpdate(pose_id[win], poses[win], point_id[active], psi[active], pose_id[anchor_of[active]], new_obs, cons, cam, prm)
        st = inc.optimize()
        p_inc, s_inc = inc.restoreDataFromG2o()
        # the same window, complete, by index
        remap_pose = np.full(Pall, -1); remap_pose[win] = np.arange(W)
        remap_point = np.full(len(uni["psi"]), -1); remap_point[active] = np.arange(len(active))
        ef = edges[e_ok].copy()
        ef["point"], ef["pose"], ef["anchor"] = remap_point[edges["point"][e_ok]], remap_pose[edges["pose"][e_ok]], remap_pose[edges["anchor"][e_ok]]
        cf = cons.copy()
        cf["pose1"], cf["pose2"] = [remap_pose[(c["pose1"] - 1000) // 3] for c in cons], [remap_pose[(c["pose2"] - 1000) // 3] for c in cons]
        full.copyDataToG2o(poses[win], psi[active], ef, cf, cam, prm)
        st_f = full.optimize()
        p_f, s_f = full.restoreDataFromG2o()
        assert (st.iterations, st.trials, st.accepted, st.terminated) == (st_f.iterations, st_f.trials, st_f.accepted, st_f.terminated), step
        np.testing.assert_allclose(st.chi2_final, st_f.chi2_final, rtol=1e-9, err_msg=f"step {step}")
        assert _rel_update_err(p_inc, p_f, poses[win]) < 1e-8 and _rel_update_err(s_inc, s_f, psi[active]) < 1e-8, step
        assert inc.info()["wave_chunks"] >= 1
        poses[win] = p_inc                                         # the graph takes the result (restoreDataFromG2o) and moves on
        psi[active] = s_inc
    inc.close(); full.close()


def test_persistent_window_rollback_and_forget(gpu_ctx):
    """(1) a svs_ba_window_update that fails after its observations were appended (here: a constraint on a pose that is not in the
    window) must leave the store as it found it -- the retry with the same observations succeeds and equals the rebuilt window instead of failing
    on duplicates; (2) svs_ba_window_forget_keyframes drops the observations of keyframes that left the map for good: the store shrinks, later
    windows are unaffected."""
    from scavislam_amd import capi, synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BA_CONSTRAINT_DTYPE, BaParams
    ctx, stream = gpu_ctx
    Pall, W = 24, 12
    uni = synth.ba_window(Pall, 1500, seed=17, n_outer=0)
    cam = _cam(uni["cam"])
    prm = BaParams.reference_defaults()
    pose_id = 40 + 2 * np.arange(Pall)
    point_id = 3 + 5 * np.arange(len(uni["psi"]))
    edges = uni["edges"]
    anchor_of = np.full(len(uni["psi"]), -1)
    anchor_of[edges["point"]] = edges["anchor"]
    inc, full = SlamGraphOptimizer(ctx, stream), SlamGraphOptimizer(ctx, stream)
    no_cons = np.zeros(0, BA_CONSTRAINT_DTYPE)

    def window(last):
        win = np.arange(last - W + 1, last + 1)
        in_win = np.zeros(Pall, bool); in_win[win] = True
        e_ok = in_win[edges["pose"]] & in_win[edges["anchor"]]
        return win, e_ok, np.unique(edges["point"][e_ok])

    def by_id(mask):
        o = edges[mask].copy()
        o["point"], o["pose"], o["anchor"] = point_id[edges["point"][mask]], pose_id[edges["pose"][mask]], -1
        return o

    def rebuilt(win, e_ok, active):
        rp = np.full(Pall, -1); rp[win] = np.arange(W)
        rl = np.full(len(uni["psi"]), -1); rl[active] = np.arange(len(active))
        ef = edges[e_ok].copy()
        ef["point"], ef["pose"], ef["anchor"] = rl[edges["point"][e_ok]], rp[edges["pose"][e_ok]], rp[edges["anchor"][e_ok]]
        full.copyDataToG2o(uni["poses"][win], uni["psi"][active], ef, no_cons, cam, prm)
        st = full.optimize()
        return st, full.restoreDataFromG2o()

    given = np.zeros(len(edges), bool)
    win, e_ok, active = window(W - 1)
    new = e_ok.copy()
    bad = np.zeros(1, BA_CONSTRAINT_DTYPE)
    bad[0]["T_21"] = np.eye(3, 4).reshape(12); bad[0]["info"] = np.eye(6).reshape(36)
    bad[0]["pose1"], bad[0]["pose2"] = pose_id[win[0]], pose_id[Pall - 1]          # the second pose is not in the window
    with pytest.raises(capi.SvsError):
        inc.windowUpdate(pose_id[win], uni["poses"][win], point_id[active], uni["psi"][active], pose_id[anchor_of[active]], by_id(new), bad, cam, prm)
    inc.windowUpdate(pose_id[win], uni["poses"][win], point_id[active], uni["psi"][active], pose_id[anchor_of[active]], by_id(new), no_cons, cam, prm)   # the retry
    given |= new
    st = inc.optimize()
    p_inc, s_inc = inc.restoreDataFromG2o()
    st_f, (p_f, s_f) = rebuilt(win, e_ok, active)
    assert (st.trials, st.accepted) == (st_f.trials, st_f.accepted)
    assert _rel_update_err(p_inc, p_f, uni["poses"][win]) < 1e-8 and _rel_update_err(s_inc, s_f, uni["psi"][active]) < 1e-8
    # slide by six keyframes, forget the six that left, slide on: every window equals the rebuilt one
    for last in range(W, Pall):
        win, e_ok, active = window(last)
        new = e_ok & ~given
        given |= new
        if last == W + 5:
            inc.windowForgetKeyframes(pose_id[:6])
        inc.windowUpdate(pose_id[win], uni["poses"][win], point_id[active], uni["psi"][active], pose_id[anchor_of[active]], by_id(new), no_cons, cam, prm)
        st = inc.optimize()
        p_inc, s_inc = inc.restoreDataFromG2o()
        st_f, (p_f, s_f) = rebuilt(win, e_ok, active)
        assert (st.trials, st.accepted) == (st_f.trials, st_f.accepted), last
        # (poses agree to ~1e-13; a weak-parallax landmark amplifies that in its own back-substitution: the bar on psi is the suite's 1e-6)
        assert _rel_update_err(p_inc, p_f, uni["poses"][win]) < 1e-8 and _rel_update_err(s_inc, s_f, uni["psi"][active]) < 1e-6, last
    inc.close(); full.close()


@pytest.mark.parametrize("P,L", [(15, 3000), (50, 20000)])
def test_set_problem_device_route_equals_host_route(gpu_ctx, P, L):
    """svs_ba_set_problem marshals on the device since round 3 (the persistent window's kernels sort, chunk and slot the edges); the host marshalling of
    rounds 1-2 stays behind the "host_marshal" switch (and for landmark shards).  Both must lead to the same problem: same solve kernel, same reduced
    system to 1e-12, same LM record, state within 1e-9 of the update; and both within 1e-6 of the oracle."""
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prob = synth.ba_window(P, L, seed=2012)
    cam = _cam(prob["cam"])
    prm = BaParams.reference_defaults()
    rng = np.random.default_rng(1)
    edges = prob["edges"][rng.permutation(len(prob["edges"]))]            # any order
    out = {}
    for route in ("device", "host"):
        o = SlamGraphOptimizer(ctx, stream)
        o.set_option("host_marshal", 1 if route == "host" else 2)
        o.copyDataToG2o(prob["poses"], prob["psi"], edges, prob["cons"], cam, prm)
        H, b, chi2 = o.reduced_system(50.0)
        st = o.optimize()
        poses, psi = o.restoreDataFromG2o()
        out[route] = (H, b, chi2, st, poses, psi, o.info())
        o.close()
    (Hd, bd, cd, sd, pd, ld, idv), (Hh, bh, ch, sh, ph, lh, ih) = out["device"], out["host"]
    assert idv["solve_kernel"] == ih["solve_kernel"] and idv["envelope_rows"] == ih["envelope_rows"]
    scale = np.abs(Hh).max()
    assert np.abs(Hd - Hh).max() <= 1e-12 * scale and np.abs(bd - bh).max() <= 1e-12 * np.abs(bh).max() and abs(cd - ch) <= 1e-12 * ch
    assert (sd.iterations, sd.trials, sd.accepted, sd.terminated) == (sh.iterations, sh.trials, sh.accepted, sh.terminated)
    assert _rel_update_err(pd, ph, prob["poses"]) < 1e-9 and _rel_update_err(ld, lh, prob["psi"]) < 1e-9
    if P <= 15:
        poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], edges, prob["cons"], cam, prm)
        assert _rel_update_err(pd, poses_ref, prob["poses"]) < 1e-6 and _rel_update_err(ld, psi_ref, prob["psi"]) < 1e-6


def test_optimize_replays_a_recorded_graph(gpu_ctx):
    """the all-accepted optimize of a resident window is recorded once as a HIP graph and replayed (svs_ba_graph_stats); same LM trajectory
    and the same state (to the order of the f64 atomics) as the kernel-by-kernel path ("no_graph"), over repeated calls, after a reset of the state, after a rejected trial
    (the host-driven remainder takes over behind the replay) and after the problem is replaced by one of another layout (new recording)."""
    from scavislam_amd import synth
    from scavislam_amd.backend import SlamGraphOptimizer
    from scavislam_amd.ctypes_types import BaParams
    ctx, stream = gpu_ctx
    prm = BaParams.reference_defaults()
    outs = {}
    for mode in ("graph", "no_graph"):
        opt = SlamGraphOptimizer(ctx, stream)
        if mode == "no_graph":
            opt.set_option("no_graph", 1)
        rows = []
        for (P, L, seed) in ((50, 20000, 2012), (50, 20000, 2012), (15, 3000, 7), (15, 3000, 7)):
            prob = synth.ba_window(P, L, seed=seed)
            cam = _cam(prob["cam"])
            if not rows or rows[-1][0] != (P, L, seed):
                opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
            else:
                opt.reset_state(prob["poses"], prob["psi"])
            st = opt.optimize()
            st2 = opt.optimize()                     # a second optimize from the state the first one left (the other state buffer is current if one trial was accepted)
            poses, psi = opt.restoreDataFromG2o()
            rows.append(((P, L, seed), (st.trials, st.accepted, st.iterations, st.terminated, st2.trials, st2.accepted), st.chi2_final, st2.chi2_final, poses, psi))
        outs[mode] = (rows, opt.graph_stats())
        opt.close()
    (rg, (launches, captures)), (rn, (l0, c0)) = outs["graph"], outs["no_graph"]
    assert (l0, c0) == (0, 0)
    # a layout is recorded when it is seen the second time (a window whose layout changes with every call never pays a recording): of the 8 optimizes the first of each
    # (layout, current state buffer) runs kernel by kernel
    assert 2 <= launches <= 6 and 2 <= captures <= 4 and captures <= launches, (launches, captures)
    for a, b in zip(rg, rn):
        assert a[1] == b[1], (a[0], a[1], b[1])
        np.testing.assert_allclose([a[2], a[3]], [b[2], b[3]], rtol=1e-9)
        # poses to the order of the f64 atomics; landmarks: that pose noise times the back-substitution's amplification for weak-parallax points (several hundred, see
        # test_every_solve_variant_matches_oracle) over two optimizes -- measured 1e-7 of psi ~ 1
        assert np.abs(a[4] - b[4]).max() <= 1e-9 * max(1.0, np.abs(b[4]).max()) and np.abs(a[5] - b[5]).max() <= 1e-6 * max(1.0, np.abs(b[5]).max())
    print(f"graph replay: {launches} of 8 optimizes replayed from {captures} recordings; LM trajectories and states equal to the kernel-by-kernel path")
