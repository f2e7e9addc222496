"""Pins the oracle's restatement of the CUDA-build dense tracker against THE REFERENCE ITSELF.

oracle/_ref/libsvs_ref_gpu.so is compiled (oracle/Makefile) from /root/reference/scavislam/gpu/dense_tracking.{cuh,cu} --
the only reference sources without third-party dependencies -- through a host emulation of the CUDA constructs they use
(oracle/ref_shim/svs_cuda_emul.h: blocks of fiber threads, __syncthreads, lockstep warpReduce, linear-filtered textures).
Everything compared here is the reference's own code: frameJacobian, matTimesVec, cameraProject (.cu:24-80),
GpuSymMatrix6::addOuter / copyTo, GpuVector6::scaledAdd (.cuh:40-217), pointcloud_kernel (.cu:82-148),
jacobianReduction_kernel / chi2_kernel with their f32 reduction trees and host sums (.cu:172-493) and
residualImage_kernel (.cu:495-569).  The restatement's SVS_SUM_F32_TREE mode must equal it BIT FOR BIT; the f64 mode (what the
HIP path is compared with) shares the per-pixel code and differs only in the accumulator.
"""
import numpy as np
import pytest

import oracle as O
from scavislam_amd import synth

CAM_SMALL = dict(f=591.524 / 2, cx=159.5, cy=119.5, b=0.07468, w=320, h=240)


def colmajor34(T):
    T = np.asarray(T, np.float64).reshape(3, 4)
    return np.array([T[r, c] for c in range(4) for r in range(3)])


@pytest.fixture(scope="module")
def ref():
    return O.ref_gpu()


@pytest.fixture(scope="module")
def case():
    c = synth.dense_full_case(cam=CAM_SMALL, seed=2013, step=0.02, yaw_deg=0.2)
    cams = synth.level_cams(c["cam"])
    fp, _, _ = O.preprocess_gpu_sem(c["img_prev"])
    fc, dx, dy = O.preprocess_gpu_sem(c["img_cur"])
    cloud = [synth.cloud_full_level(c["disp_prev"], c["cam"], l) for l in range(3)]
    return dict(c, cams=cams, fp=fp, fc=fc, dx=dx, dy=dy, cloud=cloud)


def test_per_pixel_helpers_bit_equal(ref):
    """matTimesVec / cameraProject / frameJacobian: random points through the reference-compiled inlines and through the
    restatement (a 1-pixel image whose single block sum IS the pixel's contribution: adding zeros is exact)."""
    rng = np.random.default_rng(5)
    n_checked = 0
    for _ in range(300):
        w, h = 8, 8
        f, cx, cy = 30.0 + rng.uniform(0, 5), 3.5 + rng.uniform(-.5, .5), 3.5 + rng.uniform(-.5, .5)
        z = rng.uniform(1.5, 6)
        p = np.array([(rng.uniform(0.5, 6.5) - cx) / f * z, (rng.uniform(0.5, 6.5) - cy) / f * z, z, 1.0], np.float32)
        T = synth.pose(synth.so3_exp(rng.normal(0, 0.01, 3)), rng.normal(0, 0.02, 3))
        Tcm = colmajor34(T)
        q = ref.mat34_times_vec(Tcm, p)
        uv = ref.camera_project(f, cx, cy, q)
        cur, dx, dy, prev = [rng.random((h, w)).astype(np.float32) for _ in range(4)]
        cloud = np.zeros((h, w, 4), np.float32); cloud[..., 3] = -1
        cloud[2, 3] = p
        s = O.dense_pass_full_ex(cloud, prev, cur, dx, dy, np.float32(f), np.float32(cx), np.float32(cy), Tcm.astype(np.float32), 1, O.SUM_F32_TREE)
        inside = uv[0] >= 1 and uv[1] >= 1 and uv[0] <= w - 2 and uv[1] <= h - 2
        assert int(s["n_valid"]) == int(inside)
        if not inside:
            continue
        # the reference's own pieces, chained by hand: texture fetch at uv + 0.5 (emulated unit), frameJacobian, accumulate
        tr = ref.Tracker(ref, w, h)
        tr.bind(cur, dx, dy)
        H, b = tr.jacobian_reduction(prev, cloud, Tcm, f, cx, cy)
        c2 = tr.chi2(prev, cloud, Tcm, f, cx, cy)
        tr.close()
        assert np.array_equal(H, s["H"].astype(np.float32)) and np.array_equal(b, s["b"].astype(np.float32))
        assert np.float32(c2) == np.float32(s["chi2"])
        # the restatement's f64 mode runs the same per-pixel code: a single term is exact in either accumulator
        s64 = O.dense_pass_full_ex(cloud, prev, cur, dx, dy, np.float32(f), np.float32(cx), np.float32(cy), Tcm.astype(np.float32), 1, O.SUM_F64)
        assert np.array_equal(s64["H"].astype(np.float32), H)
        n_checked += 1
    assert n_checked > 100


def test_frame_jacobian_and_packing_bit_equal(ref):
    """frameJacobian (.cu:65-80) against the formula as restated (incl. the `1./p.z` double promotions), and the packed
    upper-by-column order of GpuSymMatrix6 (addOuter .cuh:163-203, copyTo .cuh:118-132) against svs_dense_sums.H."""
    rng = np.random.default_rng(6)
    for _ in range(2000):
        p = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.5, 20), 1], np.float32)
        f, gx, gy = np.float32(rng.uniform(50, 600)), np.float32(rng.normal(0, .2)), np.float32(rng.normal(0, .2))
        J = ref.frame_jacobian(p, f, gx, gy)
        x, y, z = p[:3]
        zsq = np.float32(z * z)
        dx, dy = np.float32(gx * f), np.float32(gy * f)
        f32 = np.float32
        J0 = f32(-np.float64(dx) * (1. / np.float64(z)))
        J1 = f32(-np.float64(dy) * 1. / np.float64(z))
        J2 = f32(f32(f32(dx * x) / zsq) + f32(f32(dy * y) / zsq))
        J3 = f32(f32(f32(dx * f32(x * y)) / zsq) + f32(dy * f32(f32(1) + f32(f32(y * y) / zsq))))
        J4 = f32(f32(-dx * f32(f32(1) + f32(f32(x * x) / zsq))) - f32(f32(dy * f32(x * y)) / zsq))
        J5 = f32(f32(f32(dx * y) / z) - f32(f32(dy * x) / z))
        assert np.array_equal(J, np.array([J0, J1, J2, J3, J4, J5], np.float32))
    v = rng.normal(0, 1, 6).astype(np.float32)
    H, b = ref.accumulate(np.zeros(21, np.float32), np.zeros(6, np.float32), v, np.float32(0.25))
    k = 0
    for c in range(6):
        for r in range(c + 1):
            assert H[k] == np.float32(v[c] * v[r])      # packed upper triangle, column by column: (0,0),(0,1),(1,1),(0,2)...
            k += 1
    assert np.array_equal(b, (v * np.float32(0.25)).astype(np.float32))
    full = ref.sym_copy_to(H)
    assert np.array_equal(full, np.outer(v, v).astype(np.float32).astype(np.float64)) and np.array_equal(full, full.T)


@pytest.mark.parametrize("level", [0, 1, 2])
def test_whole_image_passes_bit_equal(ref, case, level):
    """jacobianReduction + chi2 + residualImage on every pyramid level of a rendered frame pair: H (21), b (6), chi2 and the
    residual image equal the reference-compiled kernels bit for bit, at the true pose and at a perturbed one."""
    c = case["cams"][level]
    w, h = c["w"], c["h"]
    tr = ref.Tracker(ref, w, h)
    tr.bind(case["fc"][level], case["dx"][level], case["dy"][level])
    for T in (case["T_true"], np.hstack([np.eye(3), np.zeros((3, 1))]),
              synth.pose_mul(synth.pose(synth.so3_exp([0.004, -0.003, 0.002]), [0.01, -0.004, 0.02]), case["T_true"])):
        Tcm = colmajor34(T)
        H, b = tr.jacobian_reduction(case["fp"][level], case["cloud"][level], Tcm, c["f"], c["cx"], c["cy"])
        c2 = tr.chi2(case["fp"][level], case["cloud"][level], Tcm, c["f"], c["cx"], c["cy"])
        args = (case["cloud"][level], case["fp"][level], case["fc"][level], case["dx"][level], case["dy"][level],
                np.float32(c["f"]), np.float32(c["cx"]), np.float32(c["cy"]), Tcm.astype(np.float32))
        s = O.dense_pass_full_ex(*args, 1, O.SUM_F32_TREE)
        assert s["n_valid"] > 0.5 * w * h
        assert np.array_equal(H, s["H"].astype(np.float32)), np.abs(H - s["H"]).max()
        assert np.array_equal(b, s["b"].astype(np.float32))
        s0 = O.dense_pass_full_ex(*args, 0, O.SUM_F32_TREE)
        assert np.float32(c2) == np.float32(s0["chi2"]) == np.float32(s["chi2"])
        # f64 accumulation of the same per-pixel terms stays within f32 summation noise of the reference's result
        s64 = O.dense_pass_full_ex(*args, 1, O.SUM_F64)
        assert s64["n_valid"] == s["n_valid"]
        np.testing.assert_allclose(s64["H"], H, rtol=0, atol=2e-5 * np.abs(H).max())
        np.testing.assert_allclose(s64["chi2"], c2, rtol=2e-5)
        rimg = tr.residual_image(case["fp"][level], case["cloud"][level], Tcm, c["f"], c["cx"], c["cy"])
        rimg_o = O.residual_image_full(case["cloud"][level], case["fp"][level], case["fc"][level], np.float32(c["f"]), np.float32(c["cx"]),
                                       np.float32(c["cy"]), Tcm.astype(np.float32))
        assert np.array_equal(rimg, rimg_o)
    tr.close()


@pytest.mark.parametrize("factor", [1, 2, 4])
def test_point_cloud_bit_equal(ref, case, factor):
    """computePointCloud / pointcloud_kernel (.cu:82-148) incl. its level > 0 indexing (row not scaled, d * factor)."""
    cam = case["cam"]
    l = {1: 0, 2: 1, 4: 2}[factor]
    c = case["cams"][l]
    Q = np.array([[1, 0, 0, -c["cx"]], [0, 1, 0, -c["cy"]], [0, 0, 0, c["f"]], [0, 0, 1.0 / c["b"], 0]])
    Tinv = np.vstack([synth.pose_inv(case["T_true"]), [0, 0, 0, 1]])
    TQ = Tinv @ Q
    TQcm = TQ.T.reshape(16)
    got = ref.compute_point_cloud(TQcm, case["disp_prev"], c["w"], c["h"], factor)
    mine = O.pointcloud_full(TQcm.astype(np.float32), case["disp_prev"], c["w"], c["h"], factor)
    assert np.array_equal(got, mine)
    assert (got[..., 3] == -1).any() and (got[..., 3] == 1).any()


def test_partial_blocks_bit_equal(ref):
    """image sizes that are not multiples of the 8x8 block: only the lanes inside the image take part in the tree"""
    rng = np.random.default_rng(9)
    w, h = 21, 13
    cur, dx, dy, prev = [rng.random((h, w)).astype(np.float32) for _ in range(4)]
    z = rng.uniform(2, 4, (h, w)).astype(np.float32)
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    f, cx, cy = 25.0, 10.0, 6.0
    cloud = np.stack([(u - cx) / f * z, (v - cy) / f * z, z, np.ones_like(z)], -1).astype(np.float32)
    # the reference reads the cloud of out-of-image lanes before its bounds test (.cu:191 vs :193): give it padded storage
    pad_c = np.zeros((16, 24, 4), np.float32); pad_c[:h, :w] = cloud
    pad = [np.zeros((16, 24), np.float32) for _ in range(4)]
    for a, b_ in zip(pad, (cur, dx, dy, prev)):
        a[:h, :w] = b_
    T = colmajor34(synth.pose(np.eye(3), [0.01, 0.0, 0.01]))
    L = ref.L
    import ctypes as C
    t = L.svsref_tracker_create(w, h)
    L.svsref_tracker_bind_texture(t, pad[0].ctypes.data, pad[1].ctypes.data, pad[2].ctypes.data, w, h, 24)
    H = np.zeros(21, np.float32); b = np.zeros(6, np.float32)
    L.svsref_tracker_jacobian_reduction(t, pad[3].ctypes.data, pad_c.ctypes.data, T.ctypes.data, f, cx, cy, w, h, 24, 24, H.ctypes.data, b.ctypes.data)
    c2 = L.svsref_tracker_chi2(t, pad[3].ctypes.data, pad_c.ctypes.data, T.ctypes.data, f, cx, cy, w, h, 24, 24)
    L.svsref_tracker_destroy(t)
    s = O.dense_pass_full_ex(cloud, prev, cur, dx, dy, np.float32(f), np.float32(cx), np.float32(cy), T.astype(np.float32), 1, O.SUM_F32_TREE)
    assert s["n_valid"] > 50
    assert np.array_equal(H, s["H"].astype(np.float32)) and np.array_equal(b, s["b"].astype(np.float32))
    assert np.float32(c2) == np.float32(s["chi2"])


def _ref_driven_lm(ref, case, T0):
    """DenseTracker::denseTrackingGpu (dense_tracking.cpp:60-193) written a second time, in NumPy, around the
    reference-compiled GpuTracker passes (np.linalg.solve for ldlt, scipy expm for SE3::exp)."""
    from scipy.linalg import expm
    T = np.asarray(T0, np.float64).reshape(3, 4).copy()
    rec = []
    for l in (2, 1, 0):
        c = case["cams"][l]
        tr = ref.Tracker(ref, c["w"], c["h"])
        tr.bind(case["fc"][l], case["dx"][l], case["dy"][l])
        args = (case["fp"][l], case["cloud"][l])
        K = (c["f"], c["cx"], c["cy"])
        chi2 = np.float32(tr.chi2(*args, colmajor34(T), *K))
        rec.append((l, 2, chi2, chi2))
        nu, mu, stop, trial = 2.0, float(np.float32(0.01)), False, 0
        for _ in range(15):
            while True:
                H21, b6 = tr.jacobian_reduction(*args, colmajor34(T), *K)
                H = ref.sym_copy_to(H21)
                H = H + np.diag(mu * np.diag(H))
                b = b6.astype(np.float64)
                x = np.linalg.solve(H, -b)
                tw = np.zeros((4, 4))
                wx = x[3:]
                tw[:3, :3] = [[0, -wx[2], wx[1]], [wx[2], 0, -wx[0]], [-wx[1], wx[0], 0]]
                tw[:3, 3] = x[:3]
                Tn = (expm(tw) @ np.vstack([T, [0, 0, 0, 1]]))[:3]
                new_chi2 = np.float32(tr.chi2(*args, colmajor34(Tn), *K))
                rho = float(np.float32(chi2 - new_chi2))
                rec.append((l, 1 if rho > 0 else 0, chi2, new_chi2))
                if rho > 0:
                    T, chi2 = Tn, new_chi2
                    stop = np.abs(b).max() <= 1e-10
                    mu *= max(1. / 3., 1 - (2 * rho - 1) ** 3)
                    nu, trial = 2.0, 0
                else:
                    mu *= nu
                    nu *= 2
                    trial += 1
                    if trial == 2:
                        stop = True
                if rho > 0 or stop:
                    break
            if stop:
                break
        tr.close()
    return T, np.array(rec, np.float64)


def test_lm_loop_matches_reference_driven_loop(ref, case):
    """The restated denseTrackingGpu loop (f32-tree sums = the reference's arithmetic) against the same loop driven around the
    reference-compiled passes: identical accept/reject sequence, chi2 of every trial equal to f32 rounding of the pose, final pose
    equal to 1e-9.  Also: it converges to the true motion, and the f64-sum mode follows the same trajectory."""
    I = np.hstack([np.eye(3), np.zeros((3, 1))])
    T_ref, rec_ref = _ref_driven_lm(ref, case, I)
    T_o, passes, rec_o, Tj = O.dense_tracking_gpu(case["cloud"], case["fp"], case["fc"], case["dx"], case["dy"],
                                                  [c["f"] for c in case["cams"]], [c["cx"] for c in case["cams"]],
                                                  [c["cy"] for c in case["cams"]], I, O.SUM_F32_TREE)
    assert rec_o.shape == rec_ref.shape
    assert np.array_equal(rec_o[:, :2], rec_ref[:, :2])                   # level + accept/reject sequence
    np.testing.assert_allclose(rec_o[:, 2:], rec_ref[:, 2:], rtol=1e-5)
    np.testing.assert_allclose(T_o, T_ref, rtol=0, atol=1e-9)
    n_trials = int((rec_o[:, 1] < 2).sum())
    assert passes == 3 * 2 + 2 * n_trials                                 # chi2 + residualImage per level, 2 passes per trial
    assert (rec_o[:, 1] == 1).sum() >= 3 and np.abs(T_o - case["T_true"]).max() < 0.5 * np.abs(I - case["T_true"]).max()
    T64, _, rec64, _ = O.dense_tracking_gpu(case["cloud"], case["fp"], case["fc"], case["dx"], case["dy"],
                                            [c["f"] for c in case["cams"]], [c["cx"] for c in case["cams"]],
                                            [c["cy"] for c in case["cams"]], I, O.SUM_F64)
    near_tie = np.abs(rec_o[:, 2] - rec_o[:, 3]) < 1e-5 * np.abs(rec_o[:, 2])
    if not (near_tie & (rec_o[:, 1] < 2)).any():
        assert np.array_equal(rec64[:, :2], rec_o[:, :2])
        np.testing.assert_allclose(T64, T_o, rtol=0, atol=1e-5)


def test_texture_unit_quantisation_is_small(ref, case):
    """NVIDIA's texture unit keeps the bilinear weights in 1.8 fixed point; the emulator can switch that on.  It is device
    behaviour, not reference arithmetic, and is NOT reproduced by oracle or product: this test only bounds what it would do."""
    l = 1
    c = case["cams"][l]
    tr = ref.Tracker(ref, c["w"], c["h"])
    tr.bind(case["fc"][l], case["dx"][l], case["dy"][l])
    Tcm = colmajor34(case["T_true"])
    H0, b0 = tr.jacobian_reduction(case["fp"][l], case["cloud"][l], Tcm, c["f"], c["cx"], c["cy"])
    ref.set_tex_frac_bits(8)
    try:
        H1, b1 = tr.jacobian_reduction(case["fp"][l], case["cloud"][l], Tcm, c["f"], c["cx"], c["cy"])
    finally:
        ref.set_tex_frac_bits(-1)
    tr.close()
    rel = np.abs(H1 - H0).max() / np.abs(H0).max()
    assert 0 < rel < 2e-2
