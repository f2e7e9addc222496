"""per-kernel times of one optimize (event brackets) at 50 KF / 20k"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BaParams, Cam
ctx, stream = capi.torch_context(0)
prob = synth.ba_window(50, 20000, seed=2012)
cam = Cam(*(prob["cam"][k] for k in ("f", "cx", "cy", "b", "w", "h")))
o = SlamGraphOptimizer(ctx, stream)
for kv in sys.argv[1:]:
    k, v = kv.split("="); o.set_option(k, int(v))
o.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, BaParams.reference_defaults())
o.set_timing(True)
acc = []
for i in range(12):
    o.reset_state(prob["poses"], prob["psi"]); o.optimize(); kt = o.kernel_times()
    acc.append([kt[k] / kt["n_trials"] * 1e3 for k in ("reduce_ms", "solve_ms", "backsub_ms")])
print("us per trial: reduce %.1f solve %.1f backsub %.1f" % tuple(np.median(np.array(acc[2:]), 0)), sys.argv[1:])
