"""230-pose double window (30 inner + 200 outer, loop closures): ms per optimize and per kernel (event brackets)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BaParams, Cam
import time
ctx, stream = capi.torch_context(0)
pr = synth.double_window(n_inner=30, n_outer=200, L=12000, seed=21, n_long=(100, 180, 70), n_loops=2)
cm = Cam(*(pr["cam"][k] for k in ("f", "cx", "cy", "b", "w", "h")))
o = SlamGraphOptimizer(ctx, stream)
for kv in sys.argv[1:]:
    k, v = kv.split("="); o.set_option(k, int(v))
o.copyDataToG2o(pr["poses"], pr["psi"], pr["edges"], pr["cons"], cm, BaParams.reference_defaults())
ts = []
for i in range(8):
    o.reset_state(pr["poses"], pr["psi"]); ctx.sync()
    t0 = time.perf_counter(); st = o.optimize(); ts.append(time.perf_counter() - t0)
o.set_timing(True)
o.reset_state(pr["poses"], pr["psi"]); o.optimize(); kt = o.kernel_times()
print("optimize %.3f ms; per trial (us): reduce %.0f solve %.0f backsub %.0f" % (np.median(ts[2:]) * 1e3, *(kt[k] / kt["n_trials"] * 1e3 for k in ("reduce_ms", "solve_ms", "backsub_ms"))), o.info(), st.trials, st.accepted)
