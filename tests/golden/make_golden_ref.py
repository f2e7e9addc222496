#!/usr/bin/env python3
"""Generates tests/golden/ref_*.npz FROM THE REFERENCE ITSELF: the outputs come from the libraries under oracle/_ref, which oracle/Makefile
compiles from the reference's own sources where they lie under /root/reference.  Each file holds the inputs and the
reference's outputs of one function group on a small seeded case; tests/test_oracle_cpu.py::test_reference_generated_goldens requires the C
oracle to reproduce them bit for bit -- also where neither /root/reference nor oracle/_ref exist.  Run from the repo root (needs
/root/reference):  python tests/golden/make_golden_ref.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402
from scavislam_amd import synth  # noqa: E402
from scavislam_amd.ctypes_types import CANDIDATE_DTYPE, MATCH_RESULT_DTYPE, Cam, PoseOptParams, level_cams  # noqa: E402

I34 = np.hstack([np.eye(3), np.zeros((3, 1))])


def cam_arr(c):
    return np.array([c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"]], np.float64)


# ---- GuidedMatcher<StereoCamera>::match on a 256 x 192 scene, two keyframes ----------------------------------------------------------
cam = dict(f=300.0, cx=128.0, cy=96.0, b=0.1, w=256, h=192)
cams = level_cams(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
sc = synth.Scene(77)
traj = synth.trajectory(6)
img_k0, disp_k0 = sc.render(cam, traj[0], seed=1)
img_k1, disp_k1 = sc.render(cam, traj[2], seed=2)
img_c, disp_c = sc.render(cam, traj[4], seed=3)
disp_c = disp_c.copy(); disp_c[::7, ::3] = 0.0
pyr_k = [O.build_pyramid(img_k0), O.build_pyramid(img_k1)]
pyr_c = O.build_pyramid(img_c)
corners = []
for l in range(3):
    g = O.fastgrid_for_level(pyr_c[l].shape[1], pyr_c[l].shape[0], l)
    for _ in range(3):
        xy, cc, et = O.fastgrid_detect_adaptively(g, pyr_c[l], 6)
    corners.append(xy.astype(np.int32))
rng = np.random.default_rng(5)
pts = np.concatenate([synth.candidate_points(rng, cam, disp_k0, traj[0], (150, 70, 25), kf_index=0),
                      synth.candidate_points(rng, cam, disp_k1, traj[2], (150, 70, 25), kf_index=1)])
rng.shuffle(pts)
pts["point_id"] = np.arange(len(pts))
pts[0]["kf_index"] = -1
pts[1]["anchor_obs_pyr"][:2] = (2.0, 2.0)
pts[2]["xyz_anchor"] *= 0.05
T_act = traj[2]
T_guess = synth.pose_mul(traj[4], synth.pose_inv(T_act))
T_guess[:, 3] += np.array([0.003, -0.002, 0.004])
kf_poses = np.stack([traj[0].reshape(12), traj[2].reshape(12)])
idx, obs, xyz = O.ref_match(pyr_k, kf_poses, T_guess, 1, pyr_c, disp_c, corners, cams, pts, 8, 22, 10)
assert len(idx) > 100
out = dict(cam=cam_arr(cam), kf_poses=kf_poses, T_guess=T_guess, T_act=T_act, disp_c=disp_c.astype(np.float32), pts=pts.view(np.uint8),
           ref_idx=idx, ref_obs=obs, ref_xyz=xyz)
for k in range(2):
    for l in range(3):
        out[f"kf{k}_l{l}"] = pyr_k[k][l]
for l in range(3):
    out[f"cur_l{l}"] = pyr_c[l]
    out[f"corners_l{l}"] = corners[l]
np.savez_compressed(os.path.join(HERE, "ref_matcher.npz"), **out)

# ---- the whole StereoFrontend::processFrame on the same scene (inputs shared with ref_matcher.npz; + the previous frame) -----------------------
img_prev, disp_prev = sc.render(cam, traj[3], seed=4)
pyr_prev = O.build_pyramid(img_prev)
T_prev_from_act = synth.pose_mul(traj[3], synth.pose_inv(T_act))
clouds_prev = [O.pointcloud_cpu(disp_prev, cams[l], l, T_prev_from_act) for l in range(3)]
flc = [O.convert_sobel(p) for p in pyr_c]
rngf = np.random.default_rng(3)
list_of = np.where(rngf.random(len(pts)) < 0.12, 1, np.where(rngf.random(len(pts)) < 0.1, 0, -1)).astype(np.int32)
list_of[pts["kf_index"] < 0] = -1
fr = O.ref_process_frame(pyr_k, kf_poses, 1, [(0, 37)], cams, pts, list_of, T_prev_from_act, clouds_prev, pyr_prev, pyr_c, [f[0] for f in flc], [f[1] for f in flc],
                         [f[2] for f in flc], disp_c)
assert fr["ok"] and sum(len(x) for x in fr["lines"]) > 50
out = dict(T0=T_prev_from_act, list_of=list_of, T=fr["T"], av_track_length=np.array([fr["av_track_length"]]))
for l in range(3):
    out[f"prev_l{l}"] = pyr_prev[l]
    out[f"cloud_prev_l{l}"] = clouds_prev[l]
    out[f"rimg_l{l}"] = fr["rimg"][l]
    out[f"lines_l{l}"] = fr["lines"][l]
    out[f"cloud_l{l}"] = fr["clouds"][l]
np.savez_compressed(os.path.join(HERE, "ref_frame.npz"), **out)

# ---- the reference's CUDA build of processFrame (denseTrackingGpu on the emulated kernels, matcher radius 4, computeDensePointCloudGpu) on a 320 x 240 frame:
# inputs come from the seeded renderer (tests/test_gpu_ref_frame.py::_cuda_case builds the same case), only the reference's outputs are stored
cam_small = dict(f=591.524 / 2, cx=159.5, cy=119.5, b=0.07468, w=320, h=240)
cc = synth.dense_full_case(cam=cam_small, seed=2013, step=0.02, yaw_deg=0.2)
camsc = level_cams(cam_small["f"], cam_small["cx"], cam_small["cy"], cam_small["b"], cam_small["w"], cam_small["h"])
scc = synth.Scene(2013)
trajc = synth.trajectory(5, step=0.02, yaw_deg=0.2)
img_cc, disp_cc = scc.render(cc["cam"], trajc[4], seed=2014)
assert np.array_equal(img_cc, cc["img_cur"])
rngc = np.random.default_rng(9)
ptsc = synth.candidate_points(rngc, cc["cam"], np.maximum(cc["disp_prev"], 0), I34, (260, 120, 40))
rngc.shuffle(ptsc)
ptsc["point_id"] = np.arange(len(ptsc))
list_ofc = np.where(rngc.random(len(ptsc)) < 0.2, 0, -1).astype(np.int32)
fpc, _, _ = O.preprocess_gpu_sem(cc["img_prev"])
fcc, dxc, dyc = O.preprocess_gpu_sem(cc["img_cur"])
cloud_prevc = O.ref_pointcloud_gpu(cc["disp_prev"], synth.level_cams(cc["cam"]), I34)
frc = O.ref_process_frame([O.build_pyramid(cc["img_prev"])], [I34.reshape(12)], 0, [], camsc, ptsc, list_ofc, I34, cloud_prevc, fpc, O.build_pyramid(cc["img_cur"]), fcc, dxc, dyc,
                          disp_cc, cuda_build=True)
assert frc["ok"] and sum(len(x) for x in frc["lines"]) > 40
out = dict(list_of=list_ofc, img_cur_checksum=np.array([int(cc["img_cur"].astype(np.int64).sum())]), T=frc["T"], av_track_length=np.array([frc["av_track_length"]]))
for l in range(3):
    out[f"lines_l{l}"] = frc["lines"][l]
    out[f"cloud_l{l}"] = frc["clouds"][l][::4] if l == 0 else frc["clouds"][l]
np.savez_compressed(os.path.join(HERE, "ref_frame_cuda.npz"), **out)

# ---- calcFastMotionOnly and processMatchedPoints on the matcher's own output ---------------------------------------------------------
camc = Cam(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
res = np.zeros(len(pts), MATCH_RESULT_DTYPE)
res["status"] = 5
res["status"][idx] = 0
res["obs"][idx] = obs
res["xyz_actkey"][idx] = xyz
out = dict(cam=cam_arr(cam), res=res.view(np.uint8), pts=pts.view(np.uint8), T0=T_guess)
for name, prm in (("robust", None), ("plain", PoseOptParams(0, 15, 2.0, -1.0, 1e-5)), ("fixed_mu", PoseOptParams(1, 4, 1.0, 1e-3, 1e-5))):
    T, st = O.ref_motion_only(res, camc, T_guess, prm)
    out[f"mo_{name}_T"] = T
    out[f"mo_{name}_stats"] = np.array([st.initial_chi2, st.chi2, st.max_err, st.num_obs, st.status])
T_mo = out["mo_robust_T"]
for mre in (2.0, 0.75):
    g, s, tree = O.ref_process_matched_points(res, pts, 150, camc, T_mo, mre)
    out[f"gate_{mre}_gated"] = g.view(np.uint8)
    out[f"gate_{mre}_stats"] = np.concatenate([s["num_points_grid2x2"], s["num_points_grid3x3"], s["num_matched_points"], [s["num_track_points"], s["num_obs"]]])
    out[f"gate_{mre}_av_track_length"] = np.array([s["sum_track_length"]])
    out[f"gate_{mre}_tree"] = tree
np.savez_compressed(os.path.join(HERE, "ref_motion_gate.npz"), **out)

# ---- DenseTracker, CPU branch (quarter grid) on a 128 x 96 frame pair -----------------------------------------------------------------
camd = dict(f=150.0, cx=64.0, cy=48.0, b=0.1, w=128, h=96)
camsd = level_cams(camd["f"], camd["cx"], camd["cy"], camd["b"], camd["w"], camd["h"])
scd = synth.Scene(78)
img_p, disp_p = scd.render(camd, traj[1], seed=11)
img_q, _ = scd.render(camd, traj[2], seed=12)
disp_p = disp_p.copy(); disp_p[20:40, 30:70] = 0.0
pyr_p, pyr_q = O.build_pyramid(img_p), O.build_pyramid(img_q)
out = dict(cam=cam_arr(camd), disp_p=disp_p.astype(np.float32))
for l in range(3):
    out[f"prev_l{l}"] = pyr_p[l]
    out[f"cur_l{l}"] = pyr_q[l]
T_off = synth.pose(synth.so3_exp(np.array([0.004, -0.006, 0.003])), np.array([0.03, -0.01, 0.05]))
clouds = O.ref_pointcloud_cpu(disp_p, camsd, I34)
fl = [O.convert_sobel(p) for p in pyr_q]
for l in range(3):
    out[f"cloud_l{l}"] = clouds[l]
for name, T0 in (("identity", I34), ("offset", T_off)):
    T, rimg = O.ref_dense_tracking_cpu(clouds, pyr_p, [f[0] for f in fl], [f[1] for f in fl], [f[2] for f in fl], camsd, T0)
    out[f"T0_{name}"] = np.asarray(T0)
    out[f"T_{name}"] = T
    for l in range(3):
        out[f"rimg_{name}_l{l}"] = rimg[l]
np.savez_compressed(os.path.join(HERE, "ref_dense_cpu.npz"), **out)

# ---- DenseTracker, CUDA branch: denseTrackingGpu host loop + emulated kernels on a 96 x 72 frame pair --------------------------------
camg = dict(f=591.524 / 2 * 0.3, cx=47.5, cy=35.5, b=0.07468, w=96, h=72)
c = synth.dense_full_case(cam=camg, seed=2013, step=0.02, yaw_deg=0.2)
camsg = synth.level_cams(c["cam"])
fp, _, _ = O.preprocess_gpu_sem(c["img_prev"])
fc, dx, dy = O.preprocess_gpu_sem(c["img_cur"])
cloud = [synth.cloud_full_level(c["disp_prev"], c["cam"], l) for l in range(3)]
T, rimg = O.ref_dense_tracking_gpu(cloud, fp, fc, dx, dy, camsg, I34)
out = dict(cam=cam_arr(camg), img_prev=c["img_prev"], img_cur=c["img_cur"], T=T)
for l in range(3):
    out[f"cloud_l{l}"] = cloud[l]
    out[f"rimg_l{l}"] = rimg[l]
np.savez_compressed(os.path.join(HERE, "ref_dense_gpu.npz"), **out)

# ---- g2o edge types: errors and Jacobians of 60 anchored stereo edges and 30 pose-pose edges ----------------------------------------------
L = O.ref_edges_lib()
ptr = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
rng = np.random.default_rng(17)
E = dict(cam4=[], psi=[], T_obs=[], T_anc=[], obs=[], err=[], Jp=[], Jo=[], Ja=[])
for k in range(60):
    cam4 = np.array([300 + 500 * rng.random(), 320 + 20 * rng.normal(), 240 + 20 * rng.normal(), 0.05 + 0.3 * rng.random()])
    T_anc = synth.pose(synth.so3_exp(rng.normal(0, 0.3, 3)), rng.normal(0, 2.0, 3)).reshape(12)
    T_obs = T_anc.copy() if k % 7 == 0 else synth.pose_mul(synth.pose(synth.so3_exp(rng.normal(0, 0.05, 3)), rng.normal(0, 0.4, 3)), T_anc.reshape(3, 4)).reshape(12)
    psi = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.4, 0.4), 1.0 / np.exp(rng.uniform(np.log(0.3), np.log(80.0)))])
    obs_ = np.array([rng.uniform(0, 640), rng.uniform(0, 480), rng.uniform(0, 640)])
    err, Jp, Jo, Ja = np.zeros(3), np.zeros(9), np.zeros(18), np.zeros(18)
    L.svs_refedge_psi2uvu(ptr(psi), ptr(T_obs), ptr(T_anc), ptr(obs_), ptr(cam4), ptr(err), ptr(Jp), ptr(Jo), ptr(Ja))
    for key, v in zip(E, (cam4, psi, T_obs, T_anc, obs_, err, Jp, Jo, Ja)):
        E[key].append(v)
S = dict(T21=[], T1=[], T2=[], err=[])
for k in range(30):
    T1 = synth.pose(synth.so3_exp(rng.normal(0, 0.4, 3)), rng.normal(0, 2.0, 3)).reshape(12)
    T2 = synth.pose(synth.so3_exp(rng.normal(0, 0.4, 3)), rng.normal(0, 2.0, 3)).reshape(12)
    T21 = synth.pose_mul(synth.pose(synth.so3_exp(rng.normal(0, 0.02, 3)), rng.normal(0, 0.05, 3)), synth.pose_mul(T2.reshape(3, 4), synth.pose_inv(T1.reshape(3, 4)))).reshape(12)
    err, J1, J2 = np.zeros(6), np.zeros(36), np.zeros(36)
    L.svs_refedge_se3(ptr(T21), ptr(T1), ptr(T2), ptr(err), ptr(J1), ptr(J2))
    for key, v in zip(S, (T21, T1, T2, err)):
        S[key].append(v)
np.savez_compressed(os.path.join(HERE, "ref_edges.npz"), **{"psi2uvu_" + k: np.array(v) for k, v in E.items()}, **{"se3_" + k: np.array(v) for k, v in S.items()})
print("reference-generated goldens written:", [f for f in sorted(os.listdir(HERE)) if f.startswith("ref_")])
