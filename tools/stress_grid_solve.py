"""repeat the multi-workgroup solve on a wide-envelope window and compare every run with the one-workgroup kernel's result"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BaParams, Cam
from test_gpu_ba import _with_loop_closure
ctx, stream = capi.torch_context(0)
prm = BaParams.reference_defaults()
rng = np.random.default_rng(3)
probs = {"P40_full": _with_loop_closure(synth.ba_window(40, 4000, seed=11, n_outer=2), 0, 39, rng),
         "P230_loops": synth.double_window(n_inner=30, n_outer=200, L=6000, seed=21, n_long=(100, 180, 70), n_loops=2)}
for name, pr in probs.items():
    cm = Cam(*(pr["cam"][k] for k in ("f", "cx", "cy", "b", "w", "h")))
    ref = SlamGraphOptimizer(ctx, stream)
    ref.set_option("no_grid_solve", 1)
    ref.copyDataToG2o(pr["poses"], pr["psi"], pr["edges"], pr["cons"], cm, prm)
    st0 = ref.optimize()
    p0, s0 = ref.restoreDataFromG2o()
    o = SlamGraphOptimizer(ctx, stream)
    o.copyDataToG2o(pr["poses"], pr["psi"], pr["edges"], pr["cons"], cm, prm)
    print(name, o.info())
    bad = 0
    worst = 0.0
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    for rep in range(N):
        o.reset_state(pr["poses"], pr["psi"])
        st = o.optimize()
        p, s = o.restoreDataFromG2o()
        err = np.abs(p - p0).max() / np.abs(p0 - pr["poses"]).max()
        worst = max(worst, err)
        if (st.trials, st.accepted) != (st0.trials, st0.accepted) or not err < 1e-7:
            bad += 1
            if bad <= 5:
                print("  rep", rep, "trials", st.trials, "accepted", st.accepted, "err", err)
    print(name, "bad runs", bad, "of", N, "worst rel err", worst)
