#!/usr/bin/env python3
"""Does running two half-batches of the one-call front end on two HIP streams beat one full batch?  (stage kernels of different halves overlap)"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scavislam_amd import capi, synth
from scavislam_amd.frontend import StereoFrontend
from scavislam_amd.ctypes_types import CANDIDATE_DTYPE

cam = synth.CAM_DEFAULT
sc = synth.Scene(2011)
traj = synth.trajectory(8)
NP = 8
rng = np.random.default_rng(77)
TA, TB = [], []
for p in range(NP):
    T_p = traj[4 + p % 4]
    step, yaw = rng.uniform(0.02, 0.08), np.deg2rad(rng.uniform(0.05, 0.5)) * rng.choice([-1, 1])
    T_rel = synth.pose(synth.so3_exp(np.array([rng.normal(0, 0.0005), yaw, rng.normal(0, 0.0005)])), np.array([rng.normal(0, 0.003), rng.normal(0, 0.002), -step]))
    TA.append(T_p); TB.append(synth.pose_mul(T_rel, T_p))
RA = [sc.render(cam, TA[p], seed=100 + p) for p in range(NP)]
RB = [sc.render(cam, TB[p], seed=200 + p) for p in range(NP)]
I = np.hstack([np.eye(3), np.zeros((3, 1))]).reshape(12)


class Half:
    def __init__(self, B):
        self.ctx, self.stream = capi.torch_context(0)
        self.B = B
        dev = torch.device("cuda", 0)
        self.fe = StereoFrontend(self.ctx, cam, max_points=2048, max_keyframes=1, n_streams=B)
        pair = [b % NP for b in range(B)]
        with torch.cuda.stream(self.stream):
            self.left = [torch.as_tensor(np.stack([(RA if k == 0 else RB)[p][0] for p in pair])).to(dev) for k in range(2)]
            self.disp = [torch.as_tensor(np.stack([(RA if k == 0 else RB)[p][1] for p in pair]).astype(np.float32)).to(dev) for k in range(2)]
        self.stream.synchronize()
        self.T_act = np.stack([TA[p].reshape(12) for p in pair])
        self.T_pose = [np.tile(I, (B, 1)), np.stack([synth.pose_mul(TB[p], synth.pose_inv(TA[p])).reshape(12) for p in pair])]
        self.fe.processFirstFrames(left=self.left[0], disp=self.disp[0])
        pts_of = {}
        for b in range(B):
            self.fe.keepKeyframe(0, TA[pair[b]], stream=b)
            if pair[b] not in pts_of:
                rows = []
                for l in range(3):
                    xy = self.fe.corners(b, l)[0].astype(np.int64); u0, v0 = xy[:, 0] << l, xy[:, 1] << l
                    d = RA[pair[b]][1][v0, u0]; k = d > 0.5
                    r = np.zeros(int(k.sum()), CANDIDATE_DTYPE); s_ = float(1 << l)
                    z = cam["f"] * cam["b"] / d[k]
                    r["xyz_anchor"] = np.stack([(u0[k] - cam["cx"]) / cam["f"] * z, (v0[k] - cam["cy"]) / cam["f"] * z, z], 1)
                    r["anchor_obs_pyr"] = np.stack([u0[k] / s_, v0[k] / s_, (u0[k] - d[k]) / s_], 1); r["anchor_level"] = l
                    rows.append(r)
                pts = np.concatenate(rows)[:2000]; pts["kf_index"] = 0
                pts_of[pair[b]] = pts
            self.fe.setCandidates(pts_of[pair[b]], 1000, stream=b)
        self.k = 0

    def step(self):
        self.k += 1
        f = self.k & 1
        self.fe.processFrames(self.T_pose[1 - f], self.T_act, left=self.left[f], disp=self.disp[f])


def run(halves, steps=10):
    for _ in range(2):
        for h in halves: h.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for h in halves: h.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


out = {}
for cfg in ((1024,), (512, 512), (384, 384), (256, 256, 256)):
    hs = [Half(b) for b in cfg]
    ms = run(hs)
    out[str(cfg)] = dict(ms_per_step=round(ms, 4), frames_per_s=round(sum(cfg) / ms * 1e3, 1))
    for h in hs: h.fe.close(); h.ctx.close()
print(json.dumps(out))
