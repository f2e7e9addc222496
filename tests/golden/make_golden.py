#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  The reference has no golden vectors and cannot run here, so these
are produced by the INDEPENDENT NumPy model (tests/np_model.py), not by the oracle under test:
the oracle is then required to reproduce them (tests/test_oracle_cpu.py::test_golden_fixtures).
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import np_model as M  # noqa: E402
from scavislam_amd import synth  # noqa: E402

img = synth.noise_image(96, 72, 2024)
out = dict(img=img, pyr1=M.pyr_down(img))
for t in (10, 25, 40):
    out[f"fast_{t}"] = M.fast_corners(img, t)
np.savez_compressed(os.path.join(HERE, "frontend_small.npz"), **out)

prob = synth.ba_window(5, 30, seed=21, n_outer=1)
c = prob["cam"]
camt = (c["f"], c["cx"], c["cy"], c["b"])
xp, xl, H, b = M.ba_dense_step(prob["poses"], prob["psi"], prob["edges"], prob["cons"], camt, 50.0)
np.savez_compressed(os.path.join(HERE, "ba_small.npz"), poses=prob["poses"], psi=prob["psi"],
                    edges=prob["edges"].view(np.uint8), cons=prob["cons"].view(np.uint8),
                    cam=np.array([c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"]]), xp_dense_model=xp, xl_dense_model=xl)
# stereo block matching: 128 x 65 pair (odd height) through the NumPy model of cv::StereoBM
sc = synth.Scene(11)
scam = dict(synth.CAM_DEFAULT)
scam.update(w=128, h=65, cx=64.0, cy=32.0, f=120.0, b=0.25)
sl, sr, _ = synth.render_stereo(sc, scam, synth.trajectory(2)[1], seed=5)
slp, srp = M.stereo_prefilter(sl), M.stereo_prefilter(sr)
sd16, scost = M.stereo_bm_core(slp, srp)
sval = M.stereo_validate(sd16, scost)
np.savez_compressed(os.path.join(HERE, "stereo_small.npz"), left=sl, right=sr, lp=slp, rp=srp, disp16_raw=sd16, cost=scost,
                    disp16_validated=sval, disp16_final=M.stereo_filter_speckles(sval))
print("golden written")
