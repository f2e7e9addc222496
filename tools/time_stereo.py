"""block-matching stage on B stereo pairs (640x480): ms per batch; with rocprofv3 --kernel-trace --stats around it the per-kernel split.
usage: python tools/time_stereo.py [B] [first pair] [name=value ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scavislam_amd import capi, synth
from scavislam_amd.frontend import FramePyramid, StereoMatcher
opts = [a for a in sys.argv[1:] if "=" in a]      # context options, e.g. xcd_swizzle=0
args = [a for a in sys.argv[1:] if "=" not in a]
B = int(args[0]) if len(args) > 0 else 512
OFF = int(args[1]) if len(args) > 1 else 0      # first frame pair
ctx, stream = capi.torch_context(0)
for a in opts:
    ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
sc = synth.Scene(2011)
traj = synth.trajectory(5)
pairs = [synth.render_stereo(sc, synth.CAM_DEFAULT, traj[i], seed=10 + i) for i in range(4)]
fr = FramePyramid(ctx, stream, synth.CAM_DEFAULT, batch=B, with_float=False)
fr.upload(np.stack([pairs[(b + OFF) % 4][0] for b in range(B)]))
sm = StereoMatcher(ctx, fr)
sm.upload_right(np.stack([pairs[(b + OFF) % 4][1] for b in range(B)]))
for _ in range(2):
    sm.calcDisparityCpu()
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 5
for _ in range(R):
    sm.calcDisparityCpu()
torch.cuda.synchronize()
print("stereo_bm: %.3f ms per %d frames" % ((time.perf_counter() - t0) / R * 1e3, B))
import oracle as O
for b in sorted(set((0, min(1, B - 1), min(2, B - 1), B - 1))):
    ref = O.stereo_bm(pairs[(b + OFF) % 4][0], pairs[(b + OFF) % 4][1])
    got = sm.disparity_host(b)
    print("frame", b, "differs from the oracle in", int((got != ref).sum()), "px; valid", float((got >= 0).mean()))
