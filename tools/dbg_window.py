import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scavislam_amd import capi, synth
from scavislam_amd.backend import SlamGraphOptimizer
from scavislam_amd.ctypes_types import BA_CONSTRAINT_DTYPE, BaParams, Cam
ctx, stream = capi.torch_context(0)
Pall, W = 48, 30
uni = synth.ba_window(Pall, 4000, seed=91, n_outer=0)
c = uni["cam"]; cam = Cam(c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"])
prm = BaParams.reference_defaults()
edges = uni["edges"]
anchor_of = np.full(len(uni["psi"]), -1); anchor_of[edges["point"]] = edges["anchor"]
first = int(os.environ.get('FIRST', '0'))
last = first + W - 1
win = np.arange(first, first + W)
in_win = np.zeros(Pall, bool); in_win[win] = True
e_ok = in_win[edges["pose"]] & in_win[edges["anchor"]]
active = np.unique(edges["point"][e_ok])
inc = SlamGraphOptimizer(ctx, stream); full = SlamGraphOptimizer(ctx, stream)
inc.set_option("debug", 2)
pose_id = 1000 + 3 * np.arange(Pall); point_id = 5 + 7 * np.arange(len(uni["psi"]))
sel = np.ones(len(edges), bool)
new_obs = edges[sel].copy()
new_obs["point"] = point_id[edges["point"][sel]]; new_obs["pose"] = pose_id[edges["pose"][sel]]; new_obs["anchor"] = -123
gt = uni["poses_gt"].reshape(-1, 3, 4)
cons = np.zeros(2, BA_CONSTRAINT_DTYPE)
for k, (i, j) in enumerate(((win[0], win[1]), (win[2], win[4]))):
    cons[k]["T_21"] = synth.pose_mul(gt[j], synth.pose_inv(gt[i])).reshape(12)
    cons[k]["info"] = (np.eye(6) * 200.0).reshape(36)
    cons[k]["pose1"], cons[k]["pose2"] = pose_id[i], pose_id[j]
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode == "nocons": cons = cons[:0]
inc.windowUpdate(pose_id[win], uni["poses"][win], point_id[active], uni["psi"][active], pose_id[anchor_of[active]], new_obs, cons, cam, prm)
remap_point = np.full(len(uni["psi"]), -1); remap_point[active] = np.arange(len(active))
ef = edges[e_ok].copy(); ef["point"] = remap_point[ef["point"]]
cf = cons.copy()
if len(cf): cf["pose1"], cf["pose2"] = [(c_["pose1"] - 1000) // 3 - first for c_ in cons], [(c_["pose2"] - 1000) // 3 - first for c_ in cons]
ef["pose"] -= first; ef["anchor"] -= first
full.copyDataToG2o(uni["poses"][win], uni["psi"][active], ef, cf, cam, prm)
print(inc.info(), full.info())
H1, b1, c1 = inc.reduced_system(50.0); H2, b2, c2 = full.reduced_system(50.0)
print("chi2", c1, c2, "H diff", np.abs(H1 - H2).max() / np.abs(H2).max(), "b diff", np.abs(b1 - b2).max() / np.abs(b2).max())
d = np.abs(H1 - H2).reshape(W, 6, W, 6).max((1, 3))
print("blocks differing:", np.argwhere(d > 1e-9 * np.abs(H2).max())[:10])
inc.reset_state(uni["poses"][win], uni["psi"][active]); full.reset_state(uni["poses"][win], uni["psi"][active])
s1 = inc.optimize(); s2 = full.optimize()
p1, q1 = inc.restoreDataFromG2o(); p2, q2 = full.restoreDataFromG2o()
print("optimize chi2_final", s1.chi2_final, s2.chi2_final, "trials", s1.trials, s2.trials, "pose diff", np.abs(p1 - p2).max(), "psi diff", np.abs(q1 - q2).max())
bad = np.argwhere(np.abs(q1 - q2).max(1) > 1e-9).ravel()
print("landmarks differing:", len(bad), bad[:10])
cnt = np.bincount(ef["point"], minlength=len(active))
print("their obs counts:", cnt[bad[:10]])
