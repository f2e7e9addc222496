"""Drop-in SlamGraph::optimize call at 50 KF / 20 k split into set_problem / optimize / get_state, device and host marshalling (run on a GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BaParams, Cam
ctx, stream = capi.torch_context(0)
PP, LL = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 20000)
prob = synth.ba_window(PP, LL, seed=2012)
print("window", PP, LL, "edges", len(prob["edges"]))
c = prob["cam"]; cam = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"]); prm = BaParams.reference_defaults()
for hm in (2, 1):      # svs_ba_set_option "host_marshal": 2 = marshal on the device, 1 = on the host threads
    opt = SlamGraphOptimizer(ctx, stream)
    opt.set_option("host_marshal", hm)
    for _ in range(3):
        opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm); opt.optimize(); opt.restoreDataFromG2o()
    ts = {"set": 0, "opt": 0, "get": 0}
    N = 20
    for _ in range(N):
        t0 = time.perf_counter(); opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm); t1 = time.perf_counter()
        opt.optimize(); t2 = time.perf_counter(); opt.restoreDataFromG2o(); t3 = time.perf_counter()
        ts["set"] += t1 - t0; ts["opt"] += t2 - t1; ts["get"] += t3 - t2
    print("host_marshal", "device" if hm == 2 else "host", {k: round(v / N * 1e3, 3) for k, v in ts.items()}, "ms")
    if os.environ.get("SVS_DROPIN_TRACE"):      # kernel timelines (tools/timeline.py): the trace ends with plain device-route calls
        opt.close()
        break
    opt.set_option("debug", 1)
    for _ in range(3):
        opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        opt.optimize(); opt.restoreDataFromG2o()
    opt.close()
