"""Debug helper: accuracy of the three solve kernels on a window with a loop-closure constraint (wide envelope)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle as O
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BaParams, Cam
from test_gpu_ba import _with_loop_closure, _rel_update_err

ctx, stream = capi.torch_context(0)
rng = np.random.default_rng(3)
prob = _with_loop_closure(synth.ba_window(40, 4000, seed=11, n_outer=2), 5, 27, rng)
c = prob["cam"]
cam = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])
prm = BaParams.reference_defaults()
poses_ref, psi_ref, st_ref = O.ba_optimize(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
for name, env in (("lds", {}), ("global", {"SVS_BA_NO_LDS_SOLVE": "1"})):
    for k in ("SVS_BA_NO_LDS_SOLVE",):
        os.environ.pop(k, None)
    os.environ.update(env)
    for rep in range(3):
        opt = SlamGraphOptimizer(ctx, stream)
        opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
        H, b, chi2 = opt.reduced_system(50.0)
        st = opt.optimize()
        poses, psi = opt.restoreDataFromG2o()
        print(name, "cond(H)=%.2e" % np.linalg.cond(H), "pose err %.2e psi err %.2e" % (_rel_update_err(poses, poses_ref, prob["poses"]), _rel_update_err(psi, psi_ref, prob["psi"])),
              "worst psi idx", int(np.abs(psi - psi_ref).max(1).argmax()))
        opt.close()
