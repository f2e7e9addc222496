"""Yardstick for the counted part of tests/test_gpu_sequence.py::test_batched_call_runs_eight_sequences_of_the_reference_loop.

The reference's front end is not a deterministic function of its input frames: it hashes tr1::shared_ptr by the pointee's ADDRESS (global.h:47-54) and addNewKeyframe walks a
tr1::unordered_set of them (stereo_frontend.cpp:337-342), so the order of the neighbourhood's point list -- and with it the order in which calcFastMotionOnly adds its
observations into ONE double (`new_chi2 += sqrW(f, id_obs)`, pose_optimizer.h:236-269) and takes the step iff `chi2 - new_chi2 > 0` -- follows the heap.  This script runs the
reference's OWN CPU build (oracle/_ref/libsvs_ref_seq.so) over the eight 40-frame sub-sequences of that test twice: once on the fixed arena of the stand-in headers, and N times
with the arena's allocations padded pseudo-randomly (RefSequence(heap_jitter_seed=...)): the same binary, the same frames, another heap.  What differs between those runs is
the spread any other build of the same arithmetic has to be allowed; the summary goes to tests/golden/yardstick_heap_order.json.

    python tools/yardstick_heap_order.py [N=16]        (CPU only; needs /root/reference to have built oracle/_ref; ~3 s per run)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
B, M, STEP = 8, 40, 16      # as in the GPU test


def run_sub(cams, sub, seed):
    import oracle as O
    import seq_common as S
    seq = O.RefSequence(cams, heap_jitter_seed=seed)
    out = []
    for img, disp in sub:
        r = seq.step(img, disp)
        r["crc"] = S.frame_crc(img, disp)
        r["lines"] = [ln[np.lexsort((ln[:, 4], ln[:, 3], ln[:, 2], ln[:, 1], -ln[:, 0]))] if len(ln) else ln for ln in r["lines"]]
        if r["dropped"]:
            r["new_ids"], r["new_val"] = seq.new_points(r["actkey_id"])
        out.append(r)
    seq.close()
    return out


def measure(n_seeds, streams=range(B), log=print):
    import seq_common as S
    from scavislam_amd.ctypes_types import level_cams
    cam = S.cam_of("default")
    cams = level_cams(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
    fl = list(S.frames("default", STEP * (B - 1) + M))
    rows = []
    for b in streams:
        sub = fl[b * STEP:b * STEP + M]
        ref = run_sub(cams, sub, 0)
        other, frames, dT, hard = [], [], [], []
        for k in range(n_seeds):
            a = run_sub(cams, sub, 1 + 100 * b + k)
            try:
                st = S.compare(a, ref, "the reference on another heap vs the reference")
                other.append(int(st["other_points"])); frames.append(int(st["frames_with_other_points"])); dT.append(float(st["max_dT"]))
            except AssertionError as e:      # even seq_common.compare's HARD part (identical decisions / ids / line ends to 1e-6) can fail between two runs of the reference
                hard.append(str(e)[:160])
        rows.append(dict(stream=b, runs=n_seeds, other_points=other, frames_with_other_points=frames, max_dT=dT, hard_mismatches=hard))
        log(f"stream {b}: other accepted points per run {other}, hard mismatches {len(hard)}, worst pose deviation {max(dT) if dT else None}")
    return rows


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    t0 = time.time()
    rows = measure(n)
    out = dict(what="the reference's CPU build (oracle/_ref/libsvs_ref_seq.so) against ITSELF on another heap layout: 8 sub-sequences x 40 frames of the 'default' camera, "
                    f"{n} pseudo-random paddings of the arena each (tools/yardstick_heap_order.py)",
               max_other_points=max([max(r["other_points"]) for r in rows if r["other_points"]] + [0]),
               runs_with_other_points=sum(sum(1 for v in r["other_points"] if v) for r in rows),
               runs_with_hard_mismatch=sum(len(r["hard_mismatches"]) for r in rows),
               runs=n * B, streams=rows)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "yardstick_heap_order.json"), "w"), indent=1)
    print(f"{time.time() - t0:.0f} s; max other points {out['max_other_points']}, runs with other points {out['runs_with_other_points']} / {out['runs']}, with a hard mismatch {out['runs_with_hard_mismatch']}")
