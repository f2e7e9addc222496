"""Timing of the device-resident full-resolution dense tracker (svs_dense_track_full) by batch size.
usage: python tools/time_dense_full.py [B ...]"""
import sys
import os
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scavislam_amd import capi, synth
from scavislam_amd.frontend import DenseTrackerGpu, GpuFrameData

ctx, stream = capi.torch_context(0)
cam = synth.CAM_RGBD
sc = synth.Scene(2013)
NV = 4
cases = [synth.dense_full_case(cam=cam, seed=2013 + i, scene=sc, frame=2 + i, step=0.02 + 0.004 * i, yaw_deg=0.2 + 0.05 * i) for i in range(NV)]
clouds = [[synth.cloud_full_level(c["disp_prev"], cam, l) for l in range(3)] for c in cases]
I = np.hstack([np.eye(3), np.zeros((3, 1))]).reshape(12)
px = sum((cam["w"] >> l) * (cam["h"] >> l) for l in range(3))
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 64, 128]:
    prev, cur = GpuFrameData(ctx, stream, cam, B), GpuFrameData(ctx, stream, cam, B)
    prev.upload(np.stack([cases[b % NV]["img_prev"] for b in range(B)]))
    cur.upload(np.stack([cases[b % NV]["img_cur"] for b in range(B)]))
    prev.preprocessing(); cur.preprocessing()
    dt = DenseTrackerGpu(ctx, cur)
    with torch.cuda.stream(stream):
        for l in range(3):
            dt.dev_ref_dense_points[l].copy_(torch.as_tensor(np.stack([clouds[b % NV][l] for b in range(B)])))
    for fuse in (False, True):
        a = dt.track_args(prev, fuse_gradients=fuse)
        T, passes, rec = dt.denseTrackingGpu(prev, I, args=a)
        # algorithmic bytes: 32 B per pixel per sweep of a level
        byts = 0
        for b in range(B):
            for l in range(3):
                byts += 32 * (cam["w"] >> l) * (cam["h"] >> l) * int((rec[b]["level"] == l).sum())
        ts = []
        for _ in range(5):
            dt._set_T(I)
            ctx.sync()
            ctx.timer_start()
            dt.denseTrackingGpu(prev, None, args=a, download=False)
            ts.append(ctx.timer_stop_ms())
        ms = float(np.median(ts))
        print(f"B={B:4d} fuse={int(fuse)} {ms:8.3f} ms/batch {ms / B:7.4f} ms/frame  sweeps/frame {passes.mean():5.1f} (min {passes.min()} max {passes.max()}) "
              f"alg {byts / 1e6:8.1f} MB -> {byts / ms / 1e6:7.1f} GB/s  err {np.abs(T[0] - cases[0]['T_true']).max():.2e}", flush=True)
    del dt, prev, cur
    torch.cuda.empty_cache()
