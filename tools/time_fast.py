"""grid FAST alone on B frames (640x480, three levels): ms per batch; with rocprofv3 --kernel-trace --stats around it the per-kernel split.
usage: python tools/time_fast.py [B] [name=value ...]      (context options, e.g. xcd_swizzle=0)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scavislam_amd import capi, synth
from scavislam_amd.frontend import FramePyramid, FastGrid
B = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 512
ctx, stream = capi.torch_context(0)
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        ctx.set_option(k, int(v))
sc = synth.Scene(2011)
traj = synth.trajectory(5)
imgs = [synth.render_stereo(sc, synth.CAM_DEFAULT, traj[i], seed=10 + i)[0] for i in range(4)]
fr = FramePyramid(ctx, stream, synth.CAM_DEFAULT, batch=B, with_float=False)
fr.upload(np.stack([imgs[b % 4] for b in range(B)]))
fr.preprocessing(with_float=False)
fg = FastGrid(ctx, fr)
for _ in range(3):
    fg.detectAdaptively()
torch.cuda.synchronize()
R = 10
t0 = time.perf_counter()
for _ in range(R):
    fg.detectAdaptively()
torch.cuda.synchronize()
print("fast: %.4f ms per %d frames" % ((time.perf_counter() - t0) / R * 1e3, B), sys.argv[1:])
xy, cc, et, ts = fg.corners(B - 1, 0)
print("corners level 0 of the last frame:", len(xy), "checksum", int(xy.astype(np.int64).sum()), "thresholds", ts.tolist())
