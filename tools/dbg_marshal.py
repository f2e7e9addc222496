"""Debug helper: svs_ba_set_problem phase timing on one optimizer object (SVS_BA_DEBUG=1 prints the phases)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BaParams, Cam

ctx, stream = capi.torch_context(0)
prob = synth.ba_window(50, 20000, seed=2012)
c = prob["cam"]
cam = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])
prm = BaParams.reference_defaults()
opt = SlamGraphOptimizer(ctx, stream)
for rep in range(6):
    t0 = time.perf_counter()
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    t1 = time.perf_counter()
    st = opt.optimize()
    t2 = time.perf_counter()
    poses, psi = opt.restoreDataFromG2o()
    t3 = time.perf_counter()
    print("copyDataToG2o %.3f ms, optimize %.3f ms, restore %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), st.chi2_final)
