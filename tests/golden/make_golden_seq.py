#!/usr/bin/env python3
"""Generates tests/golden/ref_seq_{default,newcollege}.npz FROM THE REFERENCE ITSELF: BASELINE configs[0], 200 frames through the reference's own front-end loop
(oracle/_ref/libsvs_ref_seq.so = stereo_frontend.cpp:39-528,656-1065 + matcher + dense tracker + FastGrid compiled from where they lie, CPU build, ONE StereoFrontend
alive across the frames; oracle/Makefile, oracle/ref_shim/refseq_post.inc).  Per frame: processFrame's return value, the keyframe decisions (dropped / switched), ids,
T_cur_from_actkey, average track length, the persistent FAST thresholds, a digest of the accepted points (kind + pixel position per level) and of the seeded point ids,
the keyframe-side line ends of every 8th frame, the seeded candidate points of every dropped keyframe, and the crc of the input frame (the fixture only applies where
the synthetic renderer reproduces the same bytes).  Run from the repo root (needs /root/reference):  python tests/golden/make_golden_seq.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle as O  # noqa: E402
import seq_common as S  # noqa: E402
from scavislam_amd.ctypes_types import level_cams  # noqa: E402

for camname in ("default", "newcollege"):
    cam = S.cam_of(camname)
    cams = level_cams(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
    seq = O.RefSequence(cams)
    rec = S.run(seq, camname)
    assert len(rec) == S.N_FRAMES and all(r["ok"] for r in rec)
    d = S.condense(rec)
    # recomputeFastCorners on two stored keyframes (the first and the last dropped)
    kf_ids = sorted({r["actkey_id"] for r in rec if r["dropped"]})
    d["recompute_kf"] = np.array([kf_ids[0], kf_ids[-1]], np.int32)
    for k, kf in enumerate(d["recompute_kf"]):
        for l in range(3):
            d[f"recompute_{k}_{l}"] = seq.recompute_fast_corners(kf, l).astype(np.int16)
    seq.close()
    path = os.path.join(HERE, f"ref_seq_{camname}.npz")
    np.savez_compressed(path, **d)
    print(camname, "keyframes dropped:", int(d["head"][:, 1].sum()), "switches:", int(d["head"][:, 2].sum()), "accepted points/frame:", len(d["pts"]) / S.N_FRAMES,
          os.path.getsize(path), "bytes")
