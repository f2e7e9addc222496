"""Shared by the sequence tests (BASELINE configs[0]): the 200-frame there-and-back sequence through the reference's own front-end loop (oracle.RefSequence), what is
recorded per frame and how a run is condensed into the committed fixture tests/golden/ref_seq_*.npz (made by tests/golden/make_golden_seq.py from the reference's CPU build)."""
import hashlib
import zlib

import numpy as np

N_FRAMES, TURN = 200, 110
P2_EVERY = 8      # the fixture keeps the double-valued line ends (curkey_uv_pyr) of every 8th frame; counts, flags and pixel positions of every frame


def cam_of(name):
    from scavislam_amd import synth
    return synth.CAM_DEFAULT if name == "default" else synth.CAM_NEWCOLLEGE


def frames(camname, n=N_FRAMES):
    """generator of (u8 image, f32 disparity) of the sequence"""
    from scavislam_amd import synth
    cam = cam_of(camname)
    sc = synth.Scene(2011)
    traj = synth.trajectory_there_and_back(N_FRAMES, TURN)
    for i in range(n):
        yield sc.render(cam, traj[i], seed=i)


def frame_crc(img, disp):
    return zlib.crc32(disp.tobytes(), zlib.crc32(img.tobytes()))


def run(seq, camname, n=N_FRAMES, keep_frames=None):
    """drives `seq` (an oracle.RefSequence) over the first n frames; returns the per-frame records (+ 'crc' of the frame, + the new points seeded whenever a keyframe
    was dropped)"""
    out = []
    for i, (img, disp) in enumerate(frames(camname, n)):
        r = seq.step(img, disp)
        r["crc"] = frame_crc(img, disp)
        # The ORDER of the tracked points is not a property of the reference: addNewKeyframe walks a tr1::unordered_set of shared pointers (stereo_frontend.cpp:337-342),
        # i.e. in the order of heap addresses, when it hands the matched new points to the neighbourhood's list -- two runs of the same binary differ.  Lists are
        # compared as multisets: rows sorted by (kind, pixel position, keyframe-side end).
        r["lines"] = [ln[np.lexsort((ln[:, 4], ln[:, 3], ln[:, 2], ln[:, 1], -ln[:, 0]))] if len(ln) else ln for ln in r["lines"]]
        if r["dropped"]:
            r["new_ids"], r["new_val"] = seq.new_points(r["actkey_id"])
        out.append(r)
        if keep_frames is not None:
            keep_frames.append((img, disp))
        if not r["ok"]:
            break
    return out


def int_part(r):
    """what must be IDENTICAL between two builds for one frame: the decisions, the ids, the accepted points with their kind and pixel position per level, the thresholds"""
    parts = [np.array([r["ok"], r["dropped"], r["switched"], r["actkey_id"], r["n_keyframes"], r["id_counter"], r["n_neighbourhood_points"], r["n_new_points"],
                       r["n_vertices"]], np.int64)]
    for l in range(3):
        ln = r["lines"][l]
        parts.append(np.array([len(ln)], np.int64))
        parts.append(np.round(ln[:, :3] * 4).astype(np.int64).ravel())      # is_new, uv_pyr (integer corner positions)
    parts.append(r["fast_thr"].astype(np.int64))
    if r["dropped"]:
        parts.append(r["new_ids"].astype(np.int64).ravel())
    return np.concatenate(parts)


def digest(a):
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest()[:8], np.uint64)[0]


def condense(records):
    """per-frame arrays for the fixture"""
    n = len(records)
    d = dict(crc=np.array([r["crc"] for r in records], np.uint64),
             head=np.array([[r["ok"], r["dropped"], r["switched"], r["actkey_id"], r["n_keyframes"], r["id_counter"], r["n_neighbourhood_points"], r["n_new_points"],
                             r["n_vertices"]] + [len(r["lines"][l]) for l in range(3)] for r in records], np.int32),
             fast_thr=np.array([r["fast_thr"] for r in records], np.int16),
             T=np.array([r["T"] for r in records]),
             av=np.array([r["av_track_length"] for r in records]),
             int_digest=np.array([digest(int_part(r)) for r in records], np.uint64))
    p2, p2_off = [], [0]
    for i in range(0, n, P2_EVERY):
        p2.append(np.concatenate([records[i]["lines"][l][:, 3:5] for l in range(3)]).astype(np.float64))
        p2_off.append(p2_off[-1] + len(p2[-1]))
    d["p2"] = np.concatenate(p2); d["p2_off"] = np.array(p2_off, np.int64)
    nv = [r["new_val"] for r in records if r["dropped"]]
    d["new_val"] = np.concatenate(nv); d["new_val_off"] = np.cumsum([0] + [len(v) for v in nv]).astype(np.int64)
    return d


def compare_live(a, b, pose_tol=1e-6, what="HIP branch vs CPU build"):
    """frame by frame: identical decisions / ids / accepted points / thresholds; poses within pose_tol; the double-valued line ends and new-point values within 1e-6.
    Returns (max pose deviation, number of lines compared)."""
    assert len(a) == len(b), (len(a), len(b))
    worst, n_lines = 0.0, 0
    for i, (ra, rb) in enumerate(zip(a, b)):
        ia, ib = int_part(ra), int_part(rb)
        assert ia.shape == ib.shape and np.array_equal(ia, ib), f"{what}: frame {i}: decisions / ids / accepted points / FAST thresholds differ"
        dT = np.abs(ra["T"] - rb["T"]).max()
        worst = max(worst, dT)
        assert dT <= pose_tol, f"{what}: frame {i}: pose deviation {dT:.3e}"
        for l in range(3):
            if len(ra["lines"][l]):
                assert np.abs(ra["lines"][l][:, 3:5] - rb["lines"][l][:, 3:5]).max() <= 1e-6, f"{what}: frame {i} level {l}: keyframe-side line ends"
            n_lines += len(ra["lines"][l])
        if np.isfinite(rb["av_track_length"]):
            assert abs(ra["av_track_length"] - rb["av_track_length"]) <= 1e-6 * max(1.0, abs(rb["av_track_length"])), f"{what}: frame {i}: average track length"
        if ra["dropped"]:
            assert np.abs(ra["new_val"] - rb["new_val"]).max() <= 1e-6, f"{what}: frame {i}: seeded points"
    return worst, n_lines


def compare_fixture(records, fx, pose_tol=1e-6, what="HIP branch vs fixture"):
    n = len(records)
    assert n <= len(fx["crc"])
    worst, k_new = 0.0, 0
    for i, r in enumerate(records):
        head = np.array([r["ok"], r["dropped"], r["switched"], r["actkey_id"], r["n_keyframes"], r["id_counter"], r["n_neighbourhood_points"], r["n_new_points"],
                         r["n_vertices"]] + [len(r["lines"][l]) for l in range(3)])
        assert np.array_equal(head, fx["head"][i]), f"{what}: frame {i}: decisions / counts {head} vs {fx['head'][i]}"
        assert np.array_equal(r["fast_thr"], fx["fast_thr"][i]), f"{what}: frame {i}: FAST thresholds"
        assert digest(int_part(r)) == fx["int_digest"][i], f"{what}: frame {i}: accepted points (kind, pixel position) or seeded ids differ"
        dT = np.abs(r["T"] - fx["T"][i]).max()
        worst = max(worst, dT)
        assert dT <= pose_tol, f"{what}: frame {i}: pose deviation {dT:.3e}"
        if i % P2_EVERY == 0:
            p2 = np.concatenate([r["lines"][l][:, 3:5] for l in range(3)])
            ref = fx["p2"][fx["p2_off"][i // P2_EVERY]:fx["p2_off"][i // P2_EVERY + 1]]
            assert p2.shape == ref.shape and (len(ref) == 0 or np.abs(p2 - ref).max() <= 1e-6), f"{what}: frame {i}: keyframe-side line ends"
        if r["dropped"]:
            ref = fx["new_val"][fx["new_val_off"][k_new]:fx["new_val_off"][k_new + 1]]
            assert r["new_val"].shape == ref.shape and np.abs(r["new_val"] - ref).max() <= 1e-6, f"{what}: frame {i}: seeded points"
            k_new += 1
    return worst
