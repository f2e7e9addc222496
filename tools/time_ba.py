"""Times SlamGraph::optimize on the BASELINE window sizes (run on a GPU box): python tools/time_ba.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BaParams, Cam

ctx, stream = capi.torch_context(0)
sizes = ((int(sys.argv[1]), int(sys.argv[2])),) if len(sys.argv) > 2 else ((15, 3000), (50, 20000), (100, 40000))
for P, L in sizes:
    prob = synth.ba_window(P, L, seed=2012)
    c = prob["cam"]
    cam = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])
    prm = BaParams.reference_defaults()
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    for _ in range(3):
        opt.reset_state(prob["poses"], prob["psi"]); opt.optimize()
    t = 0.0
    for _ in range(20):
        opt.reset_state(prob["poses"], prob["psi"]); ctx.sync()
        t0 = time.perf_counter(); st = opt.optimize(); ctx.sync(); t += time.perf_counter() - t0
    t_e2e = time.perf_counter()
    for _ in range(10):
        opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm); opt.optimize(); opt.restoreDataFromG2o()
    t_e2e = (time.perf_counter() - t_e2e) / 10
    t0 = time.perf_counter(); O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm); t_cpu = time.perf_counter() - t0
    print("P=%d L=%d E=%d: optimize %.3f ms (%d trials), drop-in call %.3f ms, CPU oracle %.1f ms => %.0fx" %
          (P, L, len(prob["edges"]), t / 20 * 1e3, st.trials, t_e2e * 1e3, t_cpu * 1e3, t_cpu / (t / 20)))
    opt.close()
