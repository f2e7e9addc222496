"""optimize() timing of the reference's double window (30 inner + 200 outer) with and without loop closures / long-lived points"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BaParams, Cam
ctx, stream = capi.torch_context(0)
prm = BaParams.reference_defaults()
for name, kw in (("loops+long", dict(n_long=(100, 180, 70), n_loops=2)), ("loops", dict(n_long=(), n_loops=2)), ("band", dict(n_long=(), n_loops=0))):
    pr = synth.double_window(n_inner=30, n_outer=200, L=12000, seed=2014, **kw)
    cm = Cam(*(pr["cam"][k] for k in ("f", "cx", "cy", "b", "w", "h")))
    for opts in ({}, {"no_order": 1}):
        o = SlamGraphOptimizer(ctx, stream)
        for k, v in opts.items():
            o.set_option(k, v)
        o.copyDataToG2o(pr["poses"], pr["psi"], pr["edges"], pr["cons"], cm, prm)
        o.set_timing(True)
        ts = []
        for rep in range(4):
            o.reset_state(pr["poses"], pr["psi"])
            ctx.sync()
            t0 = time.perf_counter()
            st = o.optimize()
            ctx.sync()
            ts.append(time.perf_counter() - t0)
        kt = o.kernel_times()
        print(f"{name:12s} {str(opts):22s} {np.median(ts) * 1e3:8.3f} ms/optimize  trials {st.trials} reduce {kt['reduce_ms'] / kt['n_trials']:.3f} solve {kt['solve_ms'] / kt['n_trials']:.3f} "
              f"backsub {kt['backsub_ms'] / kt['n_trials']:.3f} ms  {o.info()}", flush=True)
        o.close()

# batch throughput: W windows of 50 KF / 20 k in flight, two-front vs one-front solve
from scavislam_amd.backend import optimize_batch
import torch
prob = synth.ba_window(50, 20000, seed=2012)
cm = Cam(*(prob["cam"][k] for k in ("f", "cx", "cy", "b", "w", "h")))
for one_front in (0, 1):
    for Wn in (1, 8, 32):
        ctxs = [capi.torch_context(0) for _ in range(Wn)]
        opts_w = []
        for cw, sw in ctxs:
            ow = SlamGraphOptimizer(cw, sw)
            ow.set_option("one_front", one_front)
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
        print(f"one_front {one_front} batch {Wn:3d}: {Wn / np.median(tt[1:]):9.1f} windows/s  ({np.median(tt[1:]) * 1e3:.3f} ms per batch)  {opts_w[0].info()['solve_kernel']}", flush=True)
        for ow in opts_w:
            ow.close()
        for cw, _ in ctxs:
            cw.close()
