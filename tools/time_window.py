"""where the time of a persistent-window call goes (50 KF / 20k)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BaParams, Cam
ctx, stream = capi.torch_context(0)
prob = synth.ba_window(50, 20000, seed=2012)
camc = Cam(*(prob["cam"][k] for k in ("f", "cx", "cy", "b", "w", "h")))
prm = BaParams.reference_defaults()
pe = prob["edges"]
newest = pe["pose"] == pe["pose"].max()
anchor_of = np.zeros(len(prob["psi"]), np.int32); anchor_of[pe["point"]] = pe["anchor"]
act = np.unique(pe["point"]).astype(np.int32)
ids_p = np.arange(50, dtype=np.int32)
ow = SlamGraphOptimizer(ctx, stream)
if len(sys.argv) > 1: ow.set_option("debug", int(sys.argv[1]))
psi_act, anch_act = np.ascontiguousarray(prob["psi"][act]), np.ascontiguousarray(anchor_of[act])
obs_hist, obs_new = np.ascontiguousarray(pe[~newest]), np.ascontiguousarray(pe[newest])
tu, to, tr = [], [], []
for rep in range(8):
    ow.windowReset()
    ow.windowUpdate(ids_p, prob["poses"], act, psi_act, anch_act, obs_hist, prob["cons"], camc, prm)
    ctx.sync()
    t0 = time.perf_counter()
    ow.windowUpdate(ids_p, prob["poses"], act, psi_act, anch_act, obs_new, prob["cons"], camc, prm)
    t1 = time.perf_counter()
    st = ow.optimize()
    t2 = time.perf_counter()
    ow.restoreDataFromG2o()
    t3 = time.perf_counter()
    tu.append(t1 - t0); to.append(t2 - t1); tr.append(t3 - t2)
print("new obs", int(newest.sum()), "update %.3f optimize %.3f restore %.3f ms" % (np.median(tu[2:]) * 1e3, np.median(to[2:]) * 1e3, np.median(tr[2:]) * 1e3), ow.info())
