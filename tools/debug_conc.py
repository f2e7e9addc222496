import sys, numpy as np, threading
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import two_threads as TT
from scavislam_amd import synth
fwl = TT.frontend_workload(synth.CAM_DEFAULT)
A = TT.FrontendLoop(fwl)
o1, o2, o3 = {}, {}, {}
A.run(64, {})
A.run(200, o1); A.run(200, o2)
T_ab = np.asarray(fwl["T_ab"]).reshape(12)
print("true B pose", T_ab.round(4))
print("frame0", o1["poses"][0].round(4), "frame1", o1["poses"][1].round(4), o1["n_matched"][:6], o1["passes"][:6])
print("serial vs serial: equal", np.array_equal(o1["poses"], o2["poses"]), np.abs(o1["poses"] - o2["poses"]).max())
d = np.abs(o1["poses"] - o2["poses"]).max(1)
print("first differing frame", int(np.argmax(d > 0)) if (d > 0).any() else None)
dev = np.abs(o1["poses"][0::2] - T_ab).max(1)
print("B-frame deviation from true motion: max", dev.max(), "median", np.median(dev))

Bk = TT.BackendLoop(TT.backend_workload(True))
b1, b2 = {}, {}
Bk.run(4, b1); Bk.run(4, b2)
print("backend stats", b1["stats"][:3], "match ok", int((b1["match"]["status"] == 0).sum()), "of", b1["match"].size, "equal", b1["match"].tobytes() == b2["match"].tobytes())
print("inner ms", b1["ms_inner"], "double", b1["ms_double"], "match", b1["ms_match"], b1["info"])
