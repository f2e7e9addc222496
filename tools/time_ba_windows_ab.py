"""windows in flight (svs_ba_optimize_batch): does the figure depend on what else the process holds?  (bench.py measures 6 k windows/s where this measures 11 k)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer, optimize_batch
from scavislam_amd.ctypes_types import BaParams, Cam
prm = BaParams.reference_defaults()
prob = synth.ba_window(50, 20000, seed=2012)
cm = Cam(*(prob["cam"][k] for k in ("f", "cx", "cy", "b", "w", "h")))
def measure(tag, Wn=32, nograph=0):
    ctxs = [(capi.Context(0), None) for _ in range(Wn)]
    opts_w = []
    for cw, sw in ctxs:
        ow = SlamGraphOptimizer(cw, sw)
        ow.set_option("no_graph", nograph)
        ow.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cm, prm)
        opts_w.append(ow)
    tt = []
    for rep in range(6):
        for ow in opts_w:
            ow.reset_state(prob["poses"], prob["psi"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        optimize_batch(opts_w)
        tt.append(time.perf_counter() - t0)
    print(f"{tag:40s} batch {Wn:3d} no_graph {nograph}: {Wn / np.median(tt[1:]):9.1f} windows/s", flush=True)
    for ow in opts_w: ow.close()
    for cw, _ in ctxs: cw.close()
measure("fresh process")
extra = [capi.Context(0) for _ in range(48)]
measure("48 idle library streams alive")
ts = [torch.cuda.Stream() for _ in range(40)]
x = torch.zeros(1 << 20, device="cuda")
for s in ts:
    with torch.cuda.stream(s):
        x += 1
torch.cuda.synchronize()
measure("+ 40 torch streams used once")
main_ctx, main_stream = capi.torch_context(0)
o = SlamGraphOptimizer(main_ctx, main_stream)
o.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cm, prm); o.optimize()
with torch.cuda.stream(main_stream):
    for _ in range(5):
        o.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cm, prm); o.optimize(); o.restoreDataFromG2o()
measure("+ an optimizer on a torch stream")
with torch.cuda.stream(main_stream):
    measure("inside `with torch.cuda.stream(...)`")
measure("no_graph", nograph=1)
