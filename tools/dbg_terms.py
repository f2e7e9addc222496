import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle as O
from scavislam_amd import capi, synth
import test_gpu_dense_full as t
ctx, stream = capi.torch_context(0)
case = t._case(synth.CAM_RGBD, 2013)
level = 0
c = case["cams"][level]
T34 = t.colmajor34(t.I34).astype(np.float32)
args = (case["cloud"][level], case["fp"][level], case["fc"][level], case["dx"][level], case["dy"][level], np.float32(c["f"]), np.float32(c["cx"]), np.float32(c["cy"]), T34)
ref = O.dense_pixel_terms_full(*args)
got = t._terms_gpu(ctx, stream, *args, False)
d = got.view(np.uint32) != ref.view(np.uint32)
print("per channel diffs", d.reshape(-1, 8).sum(0))
idx = np.argwhere(d)
for (v, u, k) in idx[:12]:
    print(v, u, k, got[v, u, k], ref[v, u, k], got[v,u,k]-ref[v,u,k], "cloud", case["cloud"][level][v, u])
