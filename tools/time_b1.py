#!/usr/bin/env python3
"""Latency-mode sweep (one camera stream): workgroups per stream of the quarter-grid tracker, per-stage times of the one-call front end."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scavislam_amd import capi, synth
from scavislam_amd.frontend import StereoFrontend

cam = synth.CAM_DEFAULT
sc = synth.Scene(2011)
traj = synth.trajectory(8)
A = sc.render(cam, traj[4], seed=1)
T_B = synth.pose_mul(synth.pose(synth.so3_exp(np.array([0.0003, 0.004, -0.0002])), np.array([0.002, -0.001, -0.05])), traj[4])
B = sc.render(cam, T_B, seed=2)
T_AB = synth.pose_mul(T_B, synth.pose_inv(traj[4]))
I = np.hstack([np.eye(3), np.zeros((3, 1))])
out = {}
for nwg in [int(a) for a in sys.argv[1:]] or [0]:
    ctx, stream = capi.torch_context(0)
    if nwg:
        ctx.set_option("trk_nwg", nwg)
    fe = StereoFrontend(ctx, cam, max_points=2048, max_keyframes=1)
    fe.processFirstFrame(A[0], disp=A[1])
    fe.keepKeyframe(0, traj[4])
    corners = [fe.corners(0, l)[0] for l in range(3)]
    from scavislam_amd.ctypes_types import CANDIDATE_DTYPE
    rows = []
    for l in range(3):
        xy = corners[l].astype(np.int64); u0, v0 = xy[:, 0] << l, xy[:, 1] << l
        d = A[1][v0, u0]; k = d > 0.5
        r = np.zeros(int(k.sum()), CANDIDATE_DTYPE); s_ = float(1 << l)
        z = cam["f"] * cam["b"] / d[k]
        r["xyz_anchor"] = np.stack([(u0[k] - cam["cx"]) / cam["f"] * z, (v0[k] - cam["cy"]) / cam["f"] * z, z], 1)
        r["anchor_obs_pyr"] = np.stack([u0[k] / s_, v0[k] / s_, (u0[k] - d[k]) / s_], 1); r["anchor_level"] = l
        rows.append(r)
    pts = np.concatenate(rows)[:2000]; pts["kf_index"] = 0
    fe.setCandidates(pts, len(pts) // 2)
    seq = [(B, I, T_AB), (A, T_AB, I)]
    ms = []
    for it in range(24):
        (img, disp), guess, true = seq[it & 1]
        t0 = time.perf_counter()
        res, m, g = fe.processFrame(img, guess, traj[4], disp=disp)
        ms.append((time.perf_counter() - t0) * 1e3)
    if os.environ.get("SVS_B1_TRACE"):      # kernel timelines (tools/timeline.py over a rocprofv3 trace): the trace ends with plain host-IO frames
        print("host_io_ms", round(float(np.median(ms[4:])), 4))
        fe.close(); ctx.close()
        continue
    # the same with the frame produced in the staging buffers (no host-side copy)
    ms0 = []
    for it in range(24):
        (img, disp), guess, true = seq[it & 1]
        vl, vr, vd = fe.stagingView()
        np.copyto(vl, img); np.copyto(vd, disp)
        t0 = time.perf_counter()
        res, m, g = fe.processFrame(vl, guess, traj[4], disp=vd)
        ms0.append((time.perf_counter() - t0) * 1e3)
    fe.setTiming(True)
    st = []
    for it in range(8):
        (img, disp), guess, true = seq[it & 1]
        fe.processFrame(img, guess, traj[4], disp=disp)
        st.append(fe.stageTimes())
    err = float(np.abs(np.array(res.T_cur_from_actkey).reshape(3, 4) - true).max())
    out[nwg] = dict(host_io_ms=round(float(np.median(ms[4:])), 4), host_io_staged_ms=round(float(np.median(ms0[4:])), 4), passes=int(res.dense_passes), matched=int(res.n_matched), err=err,
                    stages={k: round(float(np.mean([s[k] for s in st[2:]])), 4) for k in st[0]})
    fe.close(); ctx.close()
print(json.dumps(out, indent=1))
