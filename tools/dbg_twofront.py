"""Debug helper: two-front vs one-front fused solve on the same window (run on a GPU box)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BaParams, Cam

ctx, stream = capi.torch_context(0)
P, L = int(sys.argv[1]), int(sys.argv[2])
pre = len(sys.argv) > 3 and sys.argv[3] == "pre"
prob = synth.ba_window(P, L, seed=2012)
c = prob["cam"]
cam = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])
out = {}
for mode in ("one", "two", "two"):
    if mode == "one":
        os.environ["SVS_BA_ONE_FRONT"] = "1"
    else:
        os.environ.pop("SVS_BA_ONE_FRONT", None)
    prm = BaParams.reference_defaults()
    opt = SlamGraphOptimizer(ctx, stream)
    opt.copyDataToG2o(prob["poses"], prob["psi"], prob["edges"], prob["cons"], cam, prm)
    if pre:
        opt.reduced_system(50.0)
    st = opt.optimize()
    poses, psi = opt.restoreDataFromG2o()
    print(mode, st.iterations, st.trials, st.accepted, st.terminated, st.chi2_init, st.chi2_final, st.lambda_final)
    if mode in out:
        print("  repeat diff", np.abs(poses - out[mode]).max())
    out[mode] = poses
    opt.close()
print("max |two - one| =", np.abs(out["two"] - out["one"]).max(), " update scale", np.abs(out["one"] - prob["poses"]).max())
