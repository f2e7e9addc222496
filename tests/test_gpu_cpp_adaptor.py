"""The C++ adaptor classes of include/scavislam_hip.hpp (reference-named call surfaces over the C ABI),
compiled with g++ and run on the GPU: FastGrid::detectAdaptively + cell_grid2d, SlamGraphBA::optimize, StereoBM and
BA_SE3_XYZ_STEREO::calcFastMotionOnly must agree with the CPU oracle exactly like the Python path does."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_adaptor_matches_oracle(tmp_path):
    import oracle as O
    from scavislam_amd import synth
    from scavislam_amd.ctypes_types import BaParams, Cam
    exe = tmp_path / "adaptor_smoke"
    libdir = os.path.join(ROOT, "scavislam_amd")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adaptor_smoke.cpp"),
                           "-o", str(exe), "-L", libdir, "-lscavislam_hip", f"-Wl,-rpath,{libdir}"])
    w, h = 640, 480
    img = synth.noise_image(w, h, 321)
    with open(tmp_path / "img.bin", "wb") as f:
        f.write(np.array([w, h], np.int32).tobytes())
        f.write(img.tobytes())
    prob = synth.ba_window(9, 500, seed=8, n_outer=2)
    c = prob["cam"]
    with open(tmp_path / "ba.bin", "wb") as f:
        f.write(np.array([len(prob["poses"]), len(prob["psi"]), len(prob["edges"]), len(prob["cons"])], np.int32).tobytes())
        f.write(np.array([c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"]], np.float64).tobytes())
        f.write(prob["poses"].astype(np.float64).tobytes())
        f.write(prob["psi"].astype(np.float64).tobytes())
        f.write(prob["edges"].tobytes())
        f.write(prob["cons"].tobytes())
    # right image = left shifted by 6 px (a constant-disparity pair) for StereoBM; synthetic track for calcFastMotionOnly
    right = np.roll(img, -6, axis=1)
    (tmp_path / "right.bin").write_bytes(right.tobytes())
    from scavislam_amd.ctypes_types import MATCH_RESULT_DTYPE
    rng = np.random.default_rng(2)
    T_true = synth.pose(synth.so3_exp(np.array([0.01, 0.02, -0.01])), np.array([0.02, -0.03, 0.06]))
    n_tr = 150
    xyz = np.stack([rng.uniform(-3, 3, n_tr), rng.uniform(-1.5, 1.5, n_tr), rng.uniform(2.5, 12, n_tr)], 1)
    pc = xyz @ T_true[:, :3].T + T_true[:, 3]
    track = np.zeros(n_tr, MATCH_RESULT_DTYPE)
    track["obs"] = np.stack([pc[:, 0] / pc[:, 2] * c["f"] + c["cx"], pc[:, 1] / pc[:, 2] * c["f"] + c["cy"],
                             (pc[:, 0] - c["b"]) / pc[:, 2] * c["f"] + c["cx"]], 1) + rng.normal(0, 0.3, (n_tr, 3))
    track["xyz_actkey"] = xyz
    track["status"][::9] = 5
    with open(tmp_path / "track.bin", "wb") as f:
        f.write(np.array([n_tr], np.int32).tobytes())
        f.write(np.eye(3, 4).astype(np.float64).tobytes())
        f.write(track.tobytes())
    out = subprocess.check_output([str(exe), str(tmp_path / "img.bin"), str(tmp_path / "ba.bin"), str(tmp_path / "right.bin"),
                                   str(tmp_path / "track.bin")]).decode().splitlines()
    # FAST
    pyr = O.build_pyramid(img)
    grids = [O.fastgrid_for_level(pyr[l].shape[1], pyr[l].shape[0], l) for l in range(3)]
    for it in range(3):
        ref = [O.fastgrid_detect_adaptively(grids[l], pyr[l], 6) for l in range(3)]
    for l in range(3):
        tok = out[l].split()
        assert tok[0] == "CORNERS" and int(tok[1]) == l
        xy = ref[l][0]
        s = 0
        for x, y in xy.tolist():
            s = (s * 1000003 + (x * 4096 + y)) % (1 << 64)
        assert int(tok[2]) == len(xy) and int(tok[3]) == s, f"level {l} corner list differs"
        nc = grids[l].gx * grids[l].gy
        assert [int(t) for t in tok[4:4 + nc]] == list(grids[l].thr[:nc])
    # BA
    cam = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])
    prm = BaParams.reference_defaults()
    poses_ref, psi_ref, st = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    ba = [l for l in out if l.startswith("BA ")][0].split()
    assert ba[0] == "BA" and (int(ba[1]), int(ba[2]), int(ba[3])) == (st.iterations, st.trials, st.accepted)
    np.testing.assert_allclose(float(ba[5]), st.chi2_final, rtol=1e-9)
    P = np.array([float(l.split()[1]) for l in out if l.startswith("P ")]).reshape(-1, 12)
    S = np.array([float(l.split()[1]) for l in out if l.startswith("S ")]).reshape(-1, 3)
    assert np.abs(P - poses_ref).max() < 1e-6 * np.abs(poses_ref - prob["poses"]).max()
    assert np.abs(S - psi_ref).max() < 1e-6 * np.abs(psi_ref - prob["psi"]).max()
    # SlamGraphBA::optimizeWindow / optimizeSlidingWindow (graph ids, xyz_anchor, (center, level) observations in shuffled order, strays from a frame
    # outside the window): the same LM record and the same result as the index-based call
    baw = [l.split() for l in out if l.startswith("BAW ")]
    assert len(baw) == 2
    for tok in baw:
        assert (int(tok[2]), int(tok[3]), int(tok[4])) == (st.iterations, st.trials, st.accepted), tok
        assert float(tok[5]) < 1e-9 * np.abs(poses_ref - prob["poses"]).max() and float(tok[6]) < 1e-7 * np.abs(psi_ref - prob["psi"]).max(), tok
    # StereoBM
    dl = [l for l in out if l.startswith("DISP ")][0].split()
    dref = O.stereo_bm(img, right)
    assert int(dl[1]) == int((dref >= 0).sum()) and float(dl[2]) == float(dref[dref >= 0].astype(np.float64).sum())
    # calcFastMotionOnly
    ml = [l for l in out if l.startswith("MOTION ")][0].split()
    Tr, sr = O.motion_only(track, cam, np.eye(3, 4))
    assert int(ml[1]) == sr.num_obs
    np.testing.assert_allclose([float(ml[2]), float(ml[3])], [sr.initial_chi2, sr.chi2], rtol=1e-9)
    np.testing.assert_allclose(np.array([float(t) for t in ml[4:16]]).reshape(3, 4), Tr, rtol=0, atol=1e-9)
    # processMatchedPoints
    from scavislam_amd.ctypes_types import CANDIDATE_DTYPE
    gl = [l for l in out if l.startswith("GATE ")][0].split()
    pts = np.zeros(len(track), CANDIDATE_DTYPE)
    pts["anchor_level"][::3] = 1
    T_cpp = np.array([float(t) for t in ml[4:16]])
    g_ref, s_ref = O.process_matched_points(track, pts, len(track) // 3, cam, T_cpp, 2.0)
    assert [int(x) for x in gl[1:5]] == [s_ref["num_obs"], s_ref["num_track_points"], g_ref["accepted"].sum(), g_ref["is_new"].sum()]
    np.testing.assert_allclose(float(gl[5]), s_ref["sum_track_length"], rtol=1e-12)
    # StereoFrontend (one call per frame) through the C++ adaptor = the same call through the Python mirror, bit for bit
    from scavislam_amd import capi
    from scavislam_amd.frontend import StereoFrontend
    fl = [l for l in out if l.startswith("FRAME ")][0].split()
    ctx, _ = capi.torch_context(0)
    camd = dict(f=c["f"], cx=c["cx"], cy=c["cy"], b=c["b"], w=w, h=h)
    nxt = np.concatenate([img[:, 2:], np.repeat(img[:, -1:], 2, axis=1)], axis=1)
    disp = np.full((h, w), 8.0, np.float32)
    fe = StereoFrontend(ctx, camd, max_points=64, max_keyframes=2)
    I = np.eye(3, 4)
    fe.processFirstFrame(img, disp=disp)
    fe.keepKeyframe(0, I)
    fe.setCandidates(np.zeros(0, CANDIDATE_DTYPE), 0)
    res, _, _ = fe.processFrame(nxt, I, I, disp=disp)
    assert int(fl[1]) == res.dense_passes and res.dense_passes >= 3 and int(fl[2]) == 0
    assert np.array_equal(np.array([float(t) for t in fl[3:15]]), np.array(res.T_cur_from_actkey))
    assert np.abs(np.array(res.T_cur_from_actkey).reshape(3, 4) - I).max() > 1e-4          # it did track the shift
    fe.close()
    ctx.close()
