"""The reference's actual concurrency on ONE GPU (stereo_slam.cpp:196, backend.cpp:157-224,735-779): the front end's processFrame loop on one thread (latency mode: one
camera stream, host image + disparity in, pose / match / gate records out per call) while a second thread runs the back end's work -- SlamGraph::optimize on the inner
window (50 keyframes / 20 k landmarks) and on the 30 + 200 double window with loop closures, and the re-registration of a keyframe (FastGrid::detect at stored thresholds +
GuidedMatcher::match, backend.cpp:735-779).  Each thread has its OWN svs_ctx (its own HIP stream and scratch), as include/scavislam_hip.h asks.

Used by tests/test_gpu_concurrency.py (results under contention == results of the serial runs, no SVS_ERR_BUSY) and by bench.py (both latencies under contention)."""
import os
import threading
import time

import numpy as np

I34 = np.hstack([np.eye(3), np.zeros((3, 1))])


# ---- workloads (host data only) -------------------------------------------------------------------------------------------------------------------------------
def frontend_workload(cam, n_points=2000, seed=2011):
    """two frames A (= the active keyframe) and B one step further; the loop alternates B (guess: A's pose), A (guess: B's pose), ... so every call tracks a real motion"""
    from scavislam_amd import synth
    sc = synth.Scene(seed)
    traj = synth.trajectory(8)
    fa, fb = sc.render(cam, traj[2], seed=2), sc.render(cam, traj[3], seed=3)
    T_ab = synth.pose_mul(traj[3], synth.pose_inv(traj[2]))
    return dict(cam=cam, A=fa, B=fb, n_points=n_points, T_kf=traj[2], T_ab=T_ab)


def candidates_from_corners(cam, disp, corners, n_points, seed=7):
    """candidate points = FAST corners of the keyframe with their stereo depth (what addNewPoints seeds from the keyframe's feature_tree, stereo_frontend.cpp:682-830)"""
    from scavislam_amd.ctypes_types import CANDIDATE_DTYPE
    rng = np.random.default_rng(seed)
    quota = [n_points * 6 // 10, n_points * 3 // 10, n_points - n_points * 6 // 10 - n_points * 3 // 10]
    rows = []
    for l in range(3):
        xy = corners[l].astype(np.int64)
        u0, v0 = xy[:, 0] << l, xy[:, 1] << l
        d = disp[v0, u0].astype(np.float64)
        keep = np.nonzero(d > 0.5)[0]
        sel = keep[rng.permutation(len(keep))[:quota[l]]]
        s_ = float(1 << l)
        z = cam["f"] * cam["b"] / d[sel]
        r = np.zeros(len(sel), CANDIDATE_DTYPE)
        r["xyz_anchor"] = np.stack([(u0[sel] - cam["cx"]) / cam["f"] * z, (v0[sel] - cam["cy"]) / cam["f"] * z, z], 1)
        r["anchor_obs_pyr"] = np.stack([u0[sel] / s_, v0[sel] / s_, (u0[sel] - d[sel]) / s_], 1)
        r["anchor_level"] = l
        rows.append(r)
    pts = np.concatenate(rows)
    pts["kf_index"] = 0
    pts["point_id"] = np.arange(len(pts))
    return pts


def backend_workload(small=False):
    from scavislam_amd import synth
    inner = synth.ba_window(15 if small else 50, 3000 if small else 20000, seed=2012)
    double = synth.double_window(n_inner=30, n_outer=60 if small else 200, L=4000 if small else 12000, seed=2014, n_long=() if small else (100, 180, 70), n_loops=2)
    sc = synth.Scene(2011)
    cam = synth.CAM_NEWCOLLEGE
    traj = synth.trajectory(8)
    root, cur = sc.render(cam, traj[1], seed=1), sc.render(cam, traj[3], seed=3)
    return dict(inner=inner, double=double, cam=cam, root=root, cur=cur, T_root=traj[1], T_cur_from_root=synth.pose_mul(traj[3], synth.pose_inv(traj[1])))


# ---- the two loops --------------------------------------------------------------------------------------------------------------------------------------------
class FrontendLoop:
    def __init__(self, wl, device=0):
        from scavislam_amd import capi
        from scavislam_amd.frontend import StereoFrontend
        self.ctx = capi.Context(device)                    # own stream
        self.wl = wl
        self.fe = StereoFrontend(self.ctx, wl["cam"], max_points=wl["n_points"], max_keyframes=1)
        self.fe.processFirstFrame(wl["A"][0], disp=wl["A"][1])
        self.fe.keepKeyframe(0, wl["T_kf"])
        pts = candidates_from_corners(wl["cam"], wl["A"][1], [self.fe.corners(0, l)[0] for l in range(3)], wl["n_points"])
        self.fe.setCandidates(pts, len(pts) // 2)
        self.seq = [(wl["B"], I34.reshape(12)), (wl["A"], np.asarray(wl["T_ab"]).reshape(12))]

    def run(self, n, out):
        """n calls of svs_frontend_process_frame; out: dict filled with poses [n,12], n_matched, passes, ms per call, and the match / gate records of the last two calls"""
        poses, nm, passes, ms, last = np.zeros((n, 12)), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n), []
        for i in range(n):
            (img, disp), guess = self.seq[i & 1]
            t0 = time.perf_counter()
            res, m, g = self.fe.processFrame(img, guess, self.wl["T_kf"], disp=disp)
            ms[i] = (time.perf_counter() - t0) * 1e3
            poses[i] = res.T_cur_from_actkey
            nm[i], passes[i] = res.n_matched, res.dense_passes
            if i >= n - 2:
                last.append((m.copy(), g.copy()))
        out.update(poses=poses, n_matched=nm, passes=passes, ms=ms, last=last)

    def close(self):
        self.fe.close()
        self.ctx.close()


class BackendLoop:
    def __init__(self, wl, device=0, ba_options=None):
        import torch
        from scavislam_amd import capi
        from scavislam_amd.backend import SlamGraphOptimizer
        from scavislam_amd.frontend import FastGrid, FramePyramid, GuidedMatcher
        torch.cuda.set_device(device)
        self.stream = torch.cuda.Stream(device=device)
        self.ctx = capi.Context(device, self.stream.cuda_stream)      # own stream, shared with torch for the uploads of the matcher
        self.wl = wl
        self.opts = {}
        for name in ("inner", "double"):
            p = wl[name]
            o = SlamGraphOptimizer(self.ctx, self.stream)
            for kv in os.environ.get("SVS_TT_BA_OPTIONS", "").split(","):      # experiments: "grid_g=144,no_tile_solve=1"
                if "=" in kv:
                    o.set_option(kv.split("=")[0], int(kv.split("=")[1]))
            for k, v in (ba_options or {}).items():
                o.set_option(k, int(v))
            o.copyDataToG2o(p["poses"], p["psi"], p["edges"], p["cons"], p["cam"])
            self.opts[name] = o
        # re-registration (backend.cpp:735-779): the root keyframe's corners at ITS stored thresholds, candidates matched into it
        cam = wl["cam"]
        self.root = FramePyramid(self.ctx, self.stream, cam, batch=1, with_float=False)
        self.root.upload(wl["root"][0][None], wl["root"][1][None]); self.root.preprocessing(with_float=False)
        self.cur = FramePyramid(self.ctx, self.stream, cam, batch=1, with_float=False)
        self.cur.upload(wl["cur"][0][None], wl["cur"][1][None]); self.cur.preprocessing(with_float=False)
        self.fast = FastGrid(self.ctx, self.cur)
        self.fast.detectAdaptively(pyr=self.root.pyr, trials=6)
        self.ctx.sync()
        pts = candidates_from_corners(cam, wl["root"][1], [self.fast.corners(0, l)[0] for l in range(3)], 1000)
        self.fast.detectAdaptively(trials=6)
        self.matcher = GuidedMatcher(self.ctx, self.cur, self.fast)
        self.match_args = self.matcher.prepare([(self.root.pyr, 0, np.asarray(wl["T_root"]).reshape(12))], np.asarray(wl["T_cur_from_root"]).reshape(12),
                                               np.asarray(wl["T_root"]).reshape(12), pts)
        self.ctx.sync()

    def run(self, n, out):
        """n rounds of: optimize(inner window), re-registration (FastGrid::detect + match), and every 4th round optimize(double window)"""
        import torch
        ms_inner, ms_double, ms_match, stats = [], [], [], []
        state = {}
        with torch.cuda.stream(self.stream):
            for i in range(n):
                for name, acc in (("inner", ms_inner),) + ((("double", ms_double),) if i % 4 == 0 else ()):
                    o, p = self.opts[name], self.wl[name]
                    o.reset_state(p["poses"], p["psi"])
                    t0 = time.perf_counter()
                    st = o.optimize()
                    acc.append((time.perf_counter() - t0) * 1e3)
                    stats.append((name, st.trials, st.accepted, st.terminated, st.chi2_final))
                    if i >= n - 4:
                        state[name] = o.restoreDataFromG2o()
                t0 = time.perf_counter()
                self.fast.detect()
                self.matcher.launch(self.match_args)
                res = self.matcher.download()
                ms_match.append((time.perf_counter() - t0) * 1e3)
        out.update(ms_inner=np.array(ms_inner), ms_double=np.array(ms_double), ms_match=np.array(ms_match), stats=stats, state=state, match=res.copy(),
                   info={k: o.info() for k, o in self.opts.items()})

    def close(self):
        for o in self.opts.values():
            o.close()
        self.fast.close()
        self.ctx.close()


def run_serial_and_concurrent(n_frames=1000, n_rounds=160, small=False, device=0, cam=None, third=None):
    """returns (serial_front, serial_back, conc_front, conc_back, errors): each loop once alone, then both at the same time from two threads.
    third: BA options of a SECOND back-end thread (e.g. dict(grid_g=16, no_tile_solve=1): its double window takes the 16-workgroup grid solve, i.e. the spin gate's
    priority lane); its serial / contended records are returned as sb["third"] / cb["third"], the contexts' gate counters as cb["gate"]"""
    from scavislam_amd import synth
    fwl = frontend_workload(cam or synth.CAM_DEFAULT)
    bwl = backend_workload(small)
    A, Bk = FrontendLoop(fwl, device), BackendLoop(bwl, device)
    C3 = BackendLoop(bwl, device, ba_options=third) if third else None
    sf, sb, cf, cb = {}, {}, {}, {}
    s3, c3 = {}, {}
    A.run(min(n_frames, 64), {})                       # warm-up (first launches, pinned staging)
    Bk.run(2, {})
    A.run(n_frames, sf)
    Bk.run(n_rounds, sb)
    if C3:
        C3.run(2, {})
        C3.run(n_rounds, s3)
    errors = []

    def guarded(fn, *a):
        try:
            fn(*a)
        except Exception as e:      # SvsError (e.g. SVS_ERR_BUSY) in a thread must reach the caller
            errors.append(repr(e))

    gate0 = [(c.ctx.get_stat("spin_lane_launches"), c.ctx.get_stat("spin_gated_launches")) for c in (A, Bk) + ((C3,) if C3 else ())]
    ta = threading.Thread(target=guarded, args=(A.run, n_frames, cf))
    tb = threading.Thread(target=guarded, args=(Bk.run, n_rounds, cb))
    tc = threading.Thread(target=guarded, args=(C3.run, n_rounds, c3)) if C3 else None
    t0 = time.perf_counter()
    ta.start(); tb.start()
    if tc:
        tc.start()
    ta.join(); tb.join()
    if tc:
        tc.join()
    wall = time.perf_counter() - t0
    # launches of the contended phase that took the spin gate's priority lane / went through the gate, per thread
    gate1 = [(c.ctx.get_stat("spin_lane_launches"), c.ctx.get_stat("spin_gated_launches")) for c in (A, Bk) + ((C3,) if C3 else ())]
    cb["gate"] = {name: dict(lane=int(g1[0] - g0[0]), gated=int(g1[1] - g0[1])) for name, g0, g1 in zip(("frontend", "backend", "third"), gate0, gate1)}
    if C3:
        sb["third"], cb["third"] = s3, c3
        C3.close()
    A.close(); Bk.close()
    return sf, sb, cf, cb, errors, wall


def summarize(sf, sb, cf, cb, wall):
    med = lambda a: round(float(np.median(a)), 4)
    p99 = lambda a: round(float(np.percentile(a, 99)), 4)
    return {
        "frames": int(len(cf["ms"])), "optimizes": int(len(cb["ms_inner"]) + len(cb["ms_double"])), "matcher_calls": int(len(cb["ms_match"])), "wall_s": round(wall, 3),
        "frontend_ms_per_frame": {"alone_median": med(sf["ms"]), "contended_median": med(cf["ms"]), "contended_p99": p99(cf["ms"])},
        "optimize_50KF_ms": {"alone_median": med(sb["ms_inner"]), "contended_median": med(cb["ms_inner"]), "contended_p99": p99(cb["ms_inner"])},
        "optimize_double_window_ms": {"alone_median": med(sb["ms_double"]), "contended_median": med(cb["ms_double"]), "contended_p99": p99(cb["ms_double"])},
        "reregistration_match_ms": {"alone_median": med(sb["ms_match"]), "contended_median": med(cb["ms_match"])},
        "solve_kernels": {k: v["solve_kernel"] for k, v in cb["info"].items()},
    }
