"""ctypes binding of the CPU oracle (oracle/libsvs_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by scavislam_amd/.  The reference's own code is pinned by the libraries under oracle/_ref (ref_* functions below),
third-party arithmetic is unpinned (see svs_oracle.h header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from scavislam_amd.ctypes_types import (BA_CONSTRAINT_DTYPE, BA_EDGE_DTYPE, CANDIDATE_DTYPE,
                                        DENSE_SUMS_DTYPE, GATED_POINT_DTYPE, KEYFRAME_DTYPE, MATCH_RESULT_DTYPE,
                                        POINT_STATS_DTYPE,
                                        BaParams, BaStats, Cam, FastGrid, PoseOptParams, PoseOptStats, StereoParams)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libsvs_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("vision.c", "ba.c", "stereo.c", "svs_oracle.h", "svs_math.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.svs_ref_ba_chi2.restype = C.c_double
    return _LIB


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


# ---- image ops -------------------------------------------------------------------------------
def pyr_down_u8(img):
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    img = np.ascontiguousarray(img)
    lib().svs_ref_pyr_down_u8(_p(img), w, h, img.strides[0], _p(out), out.strides[0])
    return out


def build_pyramid(img, levels=3):
    pyr = [np.ascontiguousarray(img)]
    for _ in range(levels - 1):
        pyr.append(pyr_down_u8(pyr[-1]))
    return pyr


def convert_sobel(img):
    h, w = img.shape
    img = np.ascontiguousarray(img)
    f = np.zeros((h, w), np.float32)
    dx = np.zeros_like(f)
    dy = np.zeros_like(f)
    lib().svs_ref_convert_sobel(_p(img), w, h, img.strides[0], _p(f), _p(dx), _p(dy), w)
    return f, dx, dy


# ---- FAST ------------------------------------------------------------------------------------
def fast9_16(img, thr, cap=1 << 20):
    img = np.ascontiguousarray(img)
    h, w = img.shape
    xy = np.zeros((cap, 2), np.int16)
    n = lib().svs_ref_fast9_16(_p(img), w, h, img.strides[0], int(thr), _p(xy), cap)
    return xy[:n].copy()


def fast_score(img, x, y):
    img = np.ascontiguousarray(img)
    return lib().svs_ref_fast_score(_p(img), img.strides[0], int(x), int(y))


def fastgrid_for_level(w, h, level):
    g = FastGrid()
    lib().svs_ref_fastgrid_init_level(C.byref(g), w, h, level)
    return g


def fastgrid_detect_adaptively(g, img, trials, cap=1 << 16):
    img = np.ascontiguousarray(img)
    xy = np.zeros((cap, 2), np.int16)
    nc = g.gx * g.gy
    cc = np.zeros(nc, np.int32)
    et = np.zeros(nc, np.int32)
    n = lib().svs_ref_fastgrid_detect_adaptively(C.byref(g), _p(img), img.strides[0], int(trials),
                                                 _p(xy), cap, _p(cc), _p(et))
    assert n <= cap
    return xy[:n].copy(), cc, et


def fastgrid_detect(g, img, cap=1 << 16):
    img = np.ascontiguousarray(img)
    xy = np.zeros((cap, 2), np.int16)
    cc = np.zeros(g.gx * g.gy, np.int32)
    n = lib().svs_ref_fastgrid_detect(C.byref(g), _p(img), img.strides[0], _p(xy), cap, _p(cc))
    assert n <= cap
    return xy[:n].copy(), cc


# ---- quadtree --------------------------------------------------------------------------------
class QuadTree:
    def __init__(self, w, h, delta=1.0):
        L = lib()
        L.svs_ref_qt_create.restype = C.c_void_p
        L.svs_ref_qt_create.argtypes = [C.c_double] * 5
        self.h = C.c_void_p(L.svs_ref_qt_create(0.0, 0.0, float(w), float(h), float(delta)))

    def insert(self, x, y, content):
        L = lib()
        L.svs_ref_qt_insert.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        return L.svs_ref_qt_insert(self.h, float(x), float(y), int(content))

    def query(self, x, y, w, h, cap=4096):
        L = lib()
        L.svs_ref_qt_query.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.c_void_p, C.c_int]
        out = np.zeros((cap, 3), np.int32)
        n = L.svs_ref_qt_query(self.h, float(x), float(y), float(w), float(h), _p(out), cap)
        return out[:n].copy()

    def __del__(self):
        try:
            L = lib()
            L.svs_ref_qt_destroy.argtypes = [C.c_void_p]
            L.svs_ref_qt_destroy(self.h)
        except Exception:
            pass


def quadtree_from_corners(xy, cell_count, w, h):
    """Insert corners the way FastGrid does (content = index within the cell)."""
    qt = QuadTree(w, h, 1.0)
    xy = np.ascontiguousarray(xy, np.int16)
    cc = np.ascontiguousarray(cell_count, np.int32)
    L = lib()
    L.svs_ref_qt_insert_corners.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.svs_ref_qt_insert_corners(qt.h, _p(xy), _p(cc), len(cc))
    return qt


# ---- matcher ---------------------------------------------------------------------------------
def warp_affine(frame, T, depth, key_uv, cam, halfpatch=5):
    frame = np.ascontiguousarray(frame)
    T = np.ascontiguousarray(T, np.float64).reshape(12)
    kuv = np.ascontiguousarray(key_uv, np.float64)
    out = np.zeros((2 * halfpatch, 2 * halfpatch), np.uint8)
    L = lib()
    L.svs_ref_warp_affine.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_void_p]
    L.svs_ref_warp_affine(_p(frame), frame.strides[0], _p(T), float(depth), _p(kuv), C.byref(cam),
                          halfpatch, _p(out))
    return out


def znssd(key, cur):
    key = np.ascontiguousarray(key, np.uint8).reshape(64)
    cur = np.ascontiguousarray(cur, np.uint8).reshape(64)
    sA = int(key.astype(np.int64).sum())
    sAA = int((key.astype(np.int64) ** 2).sum())
    return lib().svs_ref_znssd(_p(key), _p(cur), sA, sAA)


def match(kf_pyrs, kf_poses, T_cur_from_actkey, T_actkey_from_w, cur_pyr, disp, trees, cams, pts,
          radius=8, thr_mean=22, thr_std=10):
    """kf_pyrs: list (per keyframe) of 3 u8 arrays; kf_poses: [n_kf,12]; trees: 3 QuadTree."""
    n_kf = len(kf_pyrs)
    kfs = np.zeros(n_kf, KEYFRAME_DTYPE)
    keep = []
    for i, pyr in enumerate(kf_pyrs):
        kfs[i]["T_anchor_from_w"] = np.asarray(kf_poses[i], np.float64).reshape(12)
        for l in range(3):
            a = np.ascontiguousarray(pyr[l])
            keep.append(a)
            kfs[i]["pyr"][l] = a.ctypes.data
            kfs[i]["stride"][l] = a.strides[0]
    cur = [np.ascontiguousarray(a) for a in cur_pyr]
    cur_ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in cur])
    cur_strides = (C.c_int * 3)(*[a.strides[0] for a in cur])
    disp = np.ascontiguousarray(disp, np.float32)
    tree_ptrs = (C.c_void_p * 3)(*[t.h for t in trees])
    pts = np.ascontiguousarray(pts, CANDIDATE_DTYPE)
    out = np.zeros(len(pts), MATCH_RESULT_DTYPE)
    Tc = np.ascontiguousarray(T_cur_from_actkey, np.float64).reshape(12)
    Ta = np.ascontiguousarray(T_actkey_from_w, np.float64).reshape(12)
    L = lib()
    L.svs_ref_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.svs_ref_match(_p(kfs), n_kf, _p(Tc), _p(Ta), cur_ptrs, cur_strides, _p(disp),
                    disp.strides[0] // 4, tree_ptrs, cams, _p(pts), len(pts), radius, thr_mean,
                    thr_std, _p(out))
    return out


# ---- dense tracking --------------------------------------------------------------------------
def dense_pass_cpu(cloud, prev_u8, cur, dx, dy, cam, T, do_jac, want_rimg=False):
    cloud = np.ascontiguousarray(cloud, np.float32)
    ch, cw = cloud.shape[:2]
    prev_u8 = np.ascontiguousarray(prev_u8)
    cur, dx, dy = [np.ascontiguousarray(a, np.float32) for a in (cur, dx, dy)]
    T = np.ascontiguousarray(T, np.float64).reshape(12)
    out = np.zeros(1, DENSE_SUMS_DTYPE)
    rimg = np.zeros((ch, cw, 4), np.float32) if want_rimg else None
    L = lib()
    L.svs_ref_dense_pass_cpu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.svs_ref_dense_pass_cpu(_p(cloud), cw, ch, _p(prev_u8), prev_u8.strides[0], _p(cur), _p(dx),
                             _p(dy), cur.strides[0] // 4, C.byref(cam), _p(T), int(do_jac), _p(out),
                             _p(rimg) if want_rimg else None)
    return (out[0], rimg) if want_rimg else out[0]


def dense_tracking_cpu(clouds, prev_pyr, cur_f, dx_f, dy_f, cams, T, want_rimg=False, want_rec=False):
    clouds = [np.ascontiguousarray(a, np.float32) for a in clouds]
    prev = [np.ascontiguousarray(a) for a in prev_pyr]
    cur = [np.ascontiguousarray(a, np.float32) for a in cur_f]
    dx = [np.ascontiguousarray(a, np.float32) for a in dx_f]
    dy = [np.ascontiguousarray(a, np.float32) for a in dy_f]
    P3 = C.c_void_p * 3
    I3 = C.c_int * 3
    T = np.array(T, np.float64).reshape(12).copy()
    # DenseTracker's constructor fills residual_img with (0,0,0,1) (dense_tracking.cpp:54)
    rimg = [np.tile(np.array([0, 0, 0, 1], np.float32), (*c.shape[:2], 1)) for c in clouds]
    L = lib()
    L.svs_ref_dense_tracking_cpu_rec.argtypes = [C.c_void_p] * 2 + [C.c_void_p] + [C.c_void_p] * 3 + \
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rec = np.zeros((256, 4))
    nrec = C.c_int(0)
    passes = L.svs_ref_dense_tracking_cpu_rec(
        P3(*[a.ctypes.data for a in clouds]), P3(*[a.ctypes.data for a in prev]),
        I3(*[a.strides[0] for a in prev]), P3(*[a.ctypes.data for a in cur]),
        P3(*[a.ctypes.data for a in dx]), P3(*[a.ctypes.data for a in dy]),
        I3(*[a.strides[0] // 4 for a in cur]), cams, _p(T),
        P3(*[a.ctypes.data for a in rimg]) if want_rimg else None, _p(rec), 256, C.byref(nrec))
    if want_rec:
        return T.reshape(3, 4), passes, rec[:nrec.value].copy()
    if want_rimg:
        return T.reshape(3, 4), passes, rimg
    return T.reshape(3, 4), passes


def pointcloud_cpu(disp, cam, level, T_cur_from_actkey):
    disp = np.ascontiguousarray(disp, np.float32)
    out = np.zeros((cam.h // 4, cam.w // 4, 4), np.float32)
    T = np.ascontiguousarray(T_cur_from_actkey, np.float64).reshape(12)
    L = lib()
    L.svs_ref_pointcloud_cpu.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.svs_ref_pointcloud_cpu(_p(disp), disp.strides[0] // 4, C.byref(cam), level, _p(T), _p(out))
    return out


def dense_pass_full(cloud, prev, cur, dx, dy, f, cx, cy, T34_colmajor, do_jac):
    cloud = np.ascontiguousarray(cloud, np.float32)
    h, w = cloud.shape[:2]
    prev, cur, dx, dy = [np.ascontiguousarray(a, np.float32) for a in (prev, cur, dx, dy)]
    T = np.ascontiguousarray(T34_colmajor, np.float32).reshape(12)
    out = np.zeros(1, DENSE_SUMS_DTYPE)
    L = lib()
    L.svs_ref_dense_pass_full.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4 + \
        [C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    L.svs_ref_dense_pass_full(_p(cloud), w, h, w, _p(prev), _p(cur), _p(dx), _p(dy), w, f, cx, cy,
                              _p(T), int(do_jac), _p(out))
    return out[0]


def residual_image_full(cloud, prev, cur, f, cx, cy, T34_colmajor):
    cloud = np.ascontiguousarray(cloud, np.float32)
    h, w = cloud.shape[:2]
    prev, cur = [np.ascontiguousarray(a, np.float32) for a in (prev, cur)]
    T = np.ascontiguousarray(T34_colmajor, np.float32).reshape(12)
    out = np.zeros((h, w, 4), np.float32)
    L = lib()
    L.svs_ref_residual_image_full.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.svs_ref_residual_image_full(_p(cloud), w, h, w, _p(prev), _p(cur), w, f, cx, cy, _p(T), _p(out))
    return out


def pointcloud_full(TQ_colmajor, disp, w, h, factor):
    disp = np.ascontiguousarray(disp, np.float32)
    TQ = np.ascontiguousarray(TQ_colmajor, np.float32).reshape(16)
    out = np.zeros((h, w, 4), np.float32)
    L = lib()
    L.svs_ref_pointcloud_full.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    L.svs_ref_pointcloud_full(_p(TQ), _p(disp), w, h, disp.strides[0] // 4, w, factor, _p(out))
    return out


# ---- SE3 -------------------------------------------------------------------------------------
def se3_exp(x):
    T = np.zeros(12)
    x = np.ascontiguousarray(x, np.float64)
    lib().svs_ref_se3_exp(_p(x), _p(T))
    return T.reshape(3, 4)


def se3_log(T):
    x = np.zeros(6)
    T = np.ascontiguousarray(T, np.float64).reshape(12)
    lib().svs_ref_se3_log(_p(T), _p(x))
    return x


def se3_mul(A, B):
    Cm = np.zeros(12)
    A = np.ascontiguousarray(A, np.float64).reshape(12)
    B = np.ascontiguousarray(B, np.float64).reshape(12)
    lib().svs_ref_se3_mul(_p(A), _p(B), _p(Cm))
    return Cm.reshape(3, 4)


def se3_inv(A):
    B = np.zeros(12)
    A = np.ascontiguousarray(A, np.float64).reshape(12)
    lib().svs_ref_se3_inv(_p(A), _p(B))
    return B.reshape(3, 4)


# ---- BA --------------------------------------------------------------------------------------
def edge_psi2uvu(psi, T_obs, T_anc, obs, cam):
    psi, obs = [np.ascontiguousarray(a, np.float64) for a in (psi, obs)]
    T_obs, T_anc = [np.ascontiguousarray(a, np.float64).reshape(12) for a in (T_obs, T_anc)]
    err, Jp, Jo, Ja = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 6)), np.zeros((3, 6))
    lib().svs_ref_edge_psi2uvu(_p(psi), _p(T_obs), _p(T_anc), _p(obs), C.byref(cam), _p(err), _p(Jp),
                               _p(Jo), _p(Ja))
    return err, Jp, Jo, Ja


def edge_se3(T21, T1, T2):
    T21, T1, T2 = [np.ascontiguousarray(a, np.float64).reshape(12) for a in (T21, T1, T2)]
    err, J1, J2 = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
    lib().svs_ref_edge_se3(_p(T21), _p(T1), _p(T2), _p(err), _p(J1), _p(J2))
    return err, J1, J2


def _ba_args(poses, psi, edges, cons):
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
    psi = np.ascontiguousarray(psi, np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    cons = np.ascontiguousarray(cons if cons is not None else np.zeros(0, BA_CONSTRAINT_DTYPE),
                                BA_CONSTRAINT_DTYPE)
    return poses, psi, edges, cons


def ba_chi2(poses, psi, edges, cons, cam, prm):
    poses, psi, edges, cons = _ba_args(poses, psi, edges, cons)
    return lib().svs_ref_ba_chi2(len(poses), _p(poses), len(psi), _p(psi), len(edges), _p(edges),
                                 len(cons), _p(cons), C.byref(cam), C.byref(prm))


def ba_reduced_system(poses, psi, edges, cons, cam, prm, lam):
    poses, psi, edges, cons = _ba_args(poses, psi, edges, cons)
    n = 6 * len(poses)
    H = np.zeros((n, n))
    b = np.zeros(n)
    L = lib()
    L.svs_ref_ba_reduced_system.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_double, C.c_void_p, C.c_void_p]
    L.svs_ref_ba_reduced_system(len(poses), _p(poses), len(psi), _p(psi), len(edges), _p(edges),
                                len(cons), _p(cons), C.byref(cam), C.byref(prm), float(lam), _p(H), _p(b))
    return H, b


def ba_reduced_system_mt(threads, poses, psi, edges, cons, cam, prm, lam):
    """The reduced camera system from `threads` host threads (benchmark context only)."""
    poses, psi, edges, cons = _ba_args(poses, psi, edges, cons)
    n = 6 * len(poses)
    H = np.zeros((n, n))
    b = np.zeros(n)
    L = lib()
    L.svs_ref_ba_reduced_system_mt.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    L.svs_ref_ba_reduced_system_mt(int(threads), len(poses), _p(poses), len(psi), _p(psi), len(edges), _p(edges), len(cons), _p(cons),
                                   C.byref(cam), C.byref(prm), float(lam), _p(H), _p(b))
    return H, b


def ba_optimize(poses, psi, edges, cons, cam, prm):
    poses, psi, edges, cons = _ba_args(poses, psi, edges, cons)
    poses = poses.copy()
    psi = psi.copy()
    st = BaStats()
    lib().svs_ref_ba_optimize(len(poses), _p(poses), len(psi), _p(psi), len(edges), _p(edges),
                              len(cons), _p(cons), C.byref(cam), C.byref(prm), C.byref(st))
    return poses, psi, st


# ---- stereo block matching (oracle/stereo.c) ---------------------------------------------------
def stereo_prefilter(img, cap=31):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().svs_ref_stereo_prefilter_xsobel(_p(img), w, h, w, int(cap), _p(out))
    return out


def stereo_bm_core(lp, rp, prm=None):
    prm = prm or StereoParams.reference()
    lp = np.ascontiguousarray(lp, np.uint8)
    rp = np.ascontiguousarray(rp, np.uint8)
    h, w = lp.shape
    d16 = np.empty((h, w), np.int16)
    cost = np.empty((h, w), np.int32)
    lib().svs_ref_stereo_bm_core(_p(lp), _p(rp), w, h, C.byref(prm), _p(d16), _p(cost))
    return d16, cost


def stereo_validate(d16, cost, prm=None):
    prm = prm or StereoParams.reference()
    d16 = np.array(d16, np.int16, order="C")
    cost = np.ascontiguousarray(cost, np.int32)
    h, w = d16.shape
    lib().svs_ref_stereo_validate(_p(d16), _p(cost), w, h, C.byref(prm))
    return d16


def stereo_filter_speckles(d16, new_val, max_size, max_diff):
    d16 = np.array(d16, np.int16, order="C")
    h, w = d16.shape
    lib().svs_ref_stereo_filter_speckles(_p(d16), w, h, int(new_val), int(max_size), int(max_diff))
    return d16


def stereo_bm(left, right, prm=None):
    """cv::StereoBM as configured at stereo_frontend.cpp:620-653 -> float32 disparity (-1 = filtered)."""
    prm = prm or StereoParams.reference()
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    h, w = left.shape
    disp = np.empty((h, w), np.float32)
    lib().svs_ref_stereo_bm(_p(left), _p(right), w, h, w, C.byref(prm), _p(disp), w)
    return disp


# ---- motion-only refinement (oracle/vision.c: svs_ref_motion_only) ----------------------------------
def motion_only(results, cam, T, prm=None):
    """calcFastMotionOnly over the status-OK entries of a MATCH_RESULT_DTYPE array; returns (T_new 3x4, PoseOptStats)."""
    prm = prm or PoseOptParams.reference()
    res = np.ascontiguousarray(results, MATCH_RESULT_DTYPE)
    Tio = np.array(T, np.float64).reshape(12).copy()
    st = PoseOptStats()
    lib().svs_ref_motion_only(_p(res), len(res), C.byref(cam), C.byref(prm), _p(Tio), C.byref(st))
    return Tio.reshape(3, 4), st


def ref_motion_only(results, cam, T, prm=None):
    """The reference's own PoseOptimizer::calcFastMotionOnly (oracle/_ref/libsvs_ref_pose.so); same arguments as motion_only()."""
    prm = prm or PoseOptParams.reference()
    res = np.ascontiguousarray(results, MATCH_RESULT_DTYPE)
    Tio = np.array(T, np.float64).reshape(12).copy()
    st = PoseOptStats()
    L = _ref_lib("libsvs_ref_pose.so")
    L.svs_refpose_motion_only.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.svs_refpose_motion_only(_p(res), len(res), C.byref(cam), C.byref(prm), _p(Tio), C.byref(st))
    return Tio.reshape(3, 4), st


def process_matched_points(results, pts, n_new_records, cam, T, max_reproj_error=2.0):
    """StereoFrontend::processMatchedPoints over a MATCH_RESULT_DTYPE array and its CANDIDATE_DTYPE points;
    returns (GATED_POINT_DTYPE[n], POINT_STATS_DTYPE scalar)."""
    res = np.ascontiguousarray(results, MATCH_RESULT_DTYPE)
    pts = np.ascontiguousarray(pts, CANDIDATE_DTYPE)
    assert len(res) == len(pts)
    T = np.ascontiguousarray(T, np.float64).reshape(12)
    gated = np.zeros(len(res), GATED_POINT_DTYPE)
    stats = np.zeros(1, POINT_STATS_DTYPE)
    L = lib()
    L.svs_ref_process_matched_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                                 C.c_void_p, C.c_void_p]
    L.svs_ref_process_matched_points.restype = None
    L.svs_ref_process_matched_points(_p(res), _p(pts), len(res), int(n_new_records), C.byref(cam), _p(T),
                                     C.c_float(max_reproj_error), _p(gated), _p(stats))
    return gated, stats[0]


def ref_process_matched_points(results, pts, n_new_records, cam, T, max_reproj_error=2.0):
    """The reference's own StereoFrontend::processMatchedPoints (oracle/_ref/libsvs_ref_gate.so); arguments as process_matched_points().
    Returns (gated, stats, tree) -- stats["sum_track_length"] holds the reference's AVERAGE track length (its member av_track_length_),
    tree = [m, 4] (level, x, y, point_id) content of the point trees it filled, in query order."""
    res = np.ascontiguousarray(results, MATCH_RESULT_DTYPE)
    pts = np.ascontiguousarray(pts, CANDIDATE_DTYPE)
    assert len(res) == len(pts)
    T = np.ascontiguousarray(T, np.float64).reshape(12)
    gated = np.zeros(len(res), GATED_POINT_DTYPE)
    stats = np.zeros(1, POINT_STATS_DTYPE)
    tree = np.zeros((len(res) + 1, 4), np.float64)
    tn = C.c_int(0)
    L = _ref_lib("libsvs_ref_gate.so")
    L.svs_refgate_process_matched_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.svs_refgate_process_matched_points.restype = None
    L.svs_refgate_process_matched_points(_p(res), _p(pts), len(res), int(n_new_records), C.byref(cam), _p(T), C.c_float(max_reproj_error),
                                         _p(gated), _p(stats), _p(tree), len(tree), C.byref(tn))
    return gated, stats[0], tree[:tn.value].copy()


# ---- full-resolution (CUDA-build) dense tracker: restatement + the reference-compiled pin -----------------
SUM_F64, SUM_F32_TREE = 0, 1


def dense_pass_full_ex(cloud, prev, cur, dx, dy, f, cx, cy, T34_colmajor, do_jac, sum_mode=SUM_F64):
    cloud = np.ascontiguousarray(cloud, np.float32)
    h, w = cloud.shape[:2]
    prev, cur, dx, dy = [np.ascontiguousarray(a, np.float32) for a in (prev, cur, dx, dy)]
    T = np.ascontiguousarray(T34_colmajor, np.float32).reshape(12)
    out = np.zeros(1, DENSE_SUMS_DTYPE)
    L = lib()
    L.svs_ref_dense_pass_full_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4 + \
        [C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.svs_ref_dense_pass_full_ex(_p(cloud), w, h, w, _p(prev), _p(cur), _p(dx), _p(dy), w, f, cx, cy,
                                 _p(T), int(do_jac), int(sum_mode), _p(out))
    return out[0]


def dense_pixel_terms_full(cloud, prev, cur, dx, dy, f, cx, cy, T34_colmajor):
    cloud = np.ascontiguousarray(cloud, np.float32)
    h, w = cloud.shape[:2]
    prev, cur, dx, dy = [np.ascontiguousarray(a, np.float32) for a in (prev, cur, dx, dy)]
    T = np.ascontiguousarray(T34_colmajor, np.float32).reshape(12)
    out = np.zeros((h, w, 8), np.float32)
    L = lib()
    L.svs_ref_dense_pixel_terms_full.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4 + \
        [C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.svs_ref_dense_pixel_terms_full(_p(cloud), w, h, w, _p(prev), _p(cur), _p(dx), _p(dy), w, f, cx, cy, _p(T), _p(out))
    return out


def dense_tracking_gpu(cloud, prev, cur, dx, dy, f, cx, cy, T, sum_mode=SUM_F64, force=None):
    """DenseTracker::denseTrackingGpu restated (dense_tracking.cpp:60-193).  cloud/prev/cur/dx/dy: lists of 3 level
    arrays; f, cx, cy: per-level intrinsics.  Returns (T 3x4, passes, records [n][4], T_jac [3][3][4]).
    force: optional {record index: 0 / 1} -- take that accept decision at that record instead of the loop's own (near-ties, tests only)."""
    cloud = [np.ascontiguousarray(a, np.float32) for a in cloud]
    prev, cur, dx, dy = [[np.ascontiguousarray(a, np.float32) for a in lst] for lst in (prev, cur, dx, dy)]
    w = (C.c_int * 3)(*[c.shape[1] for c in cloud])
    h = (C.c_int * 3)(*[c.shape[0] for c in cloud])
    P3 = C.c_void_p * 3
    D3 = C.c_double * 3
    T = np.ascontiguousarray(T, np.float64).reshape(12).copy()
    rec = np.zeros((256, 4))
    nrec = C.c_int(0)
    Tj = np.zeros((3, 12))
    L = lib()
    L.svs_ref_dense_tracking_gpu_forced.restype = C.c_int
    L.svs_ref_dense_tracking_gpu_forced.argtypes = None
    fo = np.full(256, -1, np.int32)
    for k, v in (force or {}).items():
        fo[int(k)] = int(v)
    passes = L.svs_ref_dense_tracking_gpu_forced(
        P3(*[a.ctypes.data for a in cloud]), w, P3(*[a.ctypes.data for a in prev]), P3(*[a.ctypes.data for a in cur]),
        P3(*[a.ctypes.data for a in dx]), P3(*[a.ctypes.data for a in dy]), w, w, h,
        D3(*[float(v) for v in f]), D3(*[float(v) for v in cx]), D3(*[float(v) for v in cy]), _p(T), C.c_int(int(sum_mode)),
        _p(rec), C.c_int(256), C.byref(nrec), _p(Tj), _p(fo), C.c_int(256))
    return T.reshape(3, 4), passes, rec[:nrec.value].copy(), Tj.reshape(3, 3, 4)


def preprocess_gpu_sem(img_u8, levels=3):
    """FrameGrabber::preprocessing of the CUDA build: f32 pyramid by pyrDown on f32 + REPLICATE derivatives."""
    img_u8 = np.ascontiguousarray(img_u8, np.uint8)
    L = lib()
    h, w = img_u8.shape
    f0 = np.zeros((h, w), np.float32)
    L.svs_ref_convert_f32(_p(img_u8), w, h, img_u8.strides[0], _p(f0), w)
    pyr = [f0]
    for _ in range(levels - 1):
        s = pyr[-1]
        sh, sw = s.shape
        d = np.zeros(((sh + 1) // 2, (sw + 1) // 2), np.float32)
        L.svs_ref_pyr_down_f32(_p(s), sw, sh, sw, _p(d), d.shape[1])
        pyr.append(d)
    dxs, dys = [], []
    for s in pyr:
        sh, sw = s.shape
        dx = np.zeros_like(s)
        dy = np.zeros_like(s)
        L.svs_ref_deriv_replicate(_p(s), sw, sh, sw, _p(dx), _p(dy), sw)
        dxs.append(dx)
        dys.append(dy)
    return pyr, dxs, dys


class RefGpu:
    """oracle/_ref/libsvs_ref_gpu.so: the reference's gpu/dense_tracking.{cuh,cu} compiled on the host (oracle/Makefile,
    ref_shim/).  Test infrastructure; built only where /root/reference exists, prebuilt file travels to the GPU box."""

    def __init__(self):
        so = os.path.join(_HERE, "_ref", "libsvs_ref_gpu.so")
        if os.path.isdir("/root/reference/scavislam/gpu"):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        if not os.path.exists(so):
            raise FileNotFoundError(so)
        self.L = L = C.CDLL(so)
        L.svsref_tracker_create.restype = C.c_void_p
        L.svsref_tracker_chi2.restype = C.c_float
        L.svsref_tracker_chi2.argtypes = [C.c_void_p] * 4 + [C.c_double] * 3 + [C.c_int] * 4
        L.svsref_tracker_jacobian_reduction.argtypes = [C.c_void_p] * 4 + [C.c_double] * 3 + [C.c_int] * 4 + [C.c_void_p] * 2
        L.svsref_tracker_residual_image.argtypes = [C.c_void_p] * 4 + [C.c_double] * 3 + [C.c_int] * 4 + [C.c_void_p]
        L.svsref_tracker_bind_texture.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3
        L.svsref_tracker_destroy.argtypes = [C.c_void_p]
        L.svsref_camera_project.argtypes = [C.c_double] * 3 + [C.c_void_p] * 2
        L.svsref_frame_jacobian.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.svsref_accumulate.argtypes = [C.c_void_p] * 3 + [C.c_float]
        L.svsref_compute_point_cloud.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]

    def set_tex_frac_bits(self, bits):
        self.L.svsref_set_tex_frac_bits(int(bits))

    def mat34_times_vec(self, T_colmajor, v):
        T = np.ascontiguousarray(T_colmajor, np.float64).reshape(12)
        v = np.ascontiguousarray(v, np.float32)
        out = np.zeros(4, np.float32)
        self.L.svsref_mat34_times_vec(_p(T), _p(v), _p(out))
        return out

    def mat4_times_vec(self, M_colmajor, v):
        M = np.ascontiguousarray(M_colmajor, np.float64).reshape(16)
        v = np.ascontiguousarray(v, np.float32)
        out = np.zeros(4, np.float32)
        self.L.svsref_mat4_times_vec(_p(M), _p(v), _p(out))
        return out

    def camera_project(self, f, cx, cy, p):
        p = np.ascontiguousarray(p, np.float32)
        out = np.zeros(2, np.float32)
        self.L.svsref_camera_project(f, cx, cy, _p(p), _p(out))
        return out

    def frame_jacobian(self, p, f, dx, dy):
        p = np.ascontiguousarray(p, np.float32)
        out = np.zeros(6, np.float32)
        self.L.svsref_frame_jacobian(_p(p), f, dx, dy, _p(out))
        return out

    def accumulate(self, H21, b6, v6, s):
        H21 = np.ascontiguousarray(H21, np.float32).copy()
        b6 = np.ascontiguousarray(b6, np.float32).copy()
        v6 = np.ascontiguousarray(v6, np.float32)
        self.L.svsref_accumulate(_p(H21), _p(b6), _p(v6), float(s))
        return H21, b6

    def sym_copy_to(self, H21):
        H21 = np.ascontiguousarray(H21, np.float32)
        out = np.zeros(36)
        self.L.svsref_sym_copy_to(_p(H21), _p(out))
        return out.reshape(6, 6).T        # column-major storage

    def compute_point_cloud(self, TQ_colmajor, disp, w, h, factor):
        TQ = np.ascontiguousarray(TQ_colmajor, np.float64).reshape(16)
        disp = np.ascontiguousarray(disp, np.float32)
        out = np.zeros((h, w, 4), np.float32)
        self.L.svsref_compute_point_cloud(_p(TQ), _p(disp), w, h, disp.strides[0] // 4, w, factor, _p(out))
        return out

    class Tracker:
        def __init__(self, ref, w, h):
            self.ref, self.w, self.h = ref, w, h
            self.t = ref.L.svsref_tracker_create(w, h)

        def close(self):
            if self.t:
                self.ref.L.svsref_tracker_destroy(self.t)
                self.t = None

        def bind(self, cur, dx, dy):
            self.cur, self.dx, self.dy = [np.ascontiguousarray(a, np.float32) for a in (cur, dx, dy)]
            self.ref.L.svsref_tracker_bind_texture(self.t, _p(self.cur), _p(self.dx), _p(self.dy), self.w, self.h, self.w)

        def _common(self, prev, cloud, T34_colmajor):
            prev = np.ascontiguousarray(prev, np.float32)
            cloud = np.ascontiguousarray(cloud, np.float32)
            T = np.ascontiguousarray(T34_colmajor, np.float64).reshape(12)
            return prev, cloud, T

        def jacobian_reduction(self, prev, cloud, T34_colmajor, f, cx, cy):
            prev, cloud, T = self._common(prev, cloud, T34_colmajor)
            H = np.zeros(21, np.float32)
            b = np.zeros(6, np.float32)
            self.ref.L.svsref_tracker_jacobian_reduction(self.t, _p(prev), _p(cloud), _p(T), f, cx, cy, self.w, self.h, self.w, self.w,
                                                         _p(H), _p(b))
            return H, b

        def chi2(self, prev, cloud, T34_colmajor, f, cx, cy):
            prev, cloud, T = self._common(prev, cloud, T34_colmajor)
            return float(self.ref.L.svsref_tracker_chi2(self.t, _p(prev), _p(cloud), _p(T), f, cx, cy, self.w, self.h, self.w, self.w))

        def residual_image(self, prev, cloud, T34_colmajor, f, cx, cy):
            prev, cloud, T = self._common(prev, cloud, T34_colmajor)
            out = np.zeros((self.h, self.w, 4), np.float32)
            self.ref.L.svsref_tracker_residual_image(self.t, _p(prev), _p(cloud), _p(T), f, cx, cy, self.w, self.h, self.w, self.w, _p(out))
            return out


class RefQuadTree:
    """oracle/_ref/libsvs_ref_qt.so: the reference's own QuadTree<int> (scavislam/quadtree.h) compiled on the host against stand-in headers
    (oracle/Makefile, ref_shim/fake, ref_shim/refqt_wrap.cc).  Test infrastructure; built only where /root/reference exists."""
    _L = None

    @classmethod
    def lib(cls):
        if cls._L is None:
            so = os.path.join(_HERE, "_ref", "libsvs_ref_qt.so")
            if os.path.isdir("/root/reference/scavislam"):
                subprocess.check_call(["make", "-C", _HERE, "-s"])
            if not os.path.exists(so):
                raise FileNotFoundError(so)
            L = C.CDLL(so)
            L.svs_refqt_create.restype = C.c_void_p
            L.svs_refqt_create.argtypes = [C.c_double] * 5
            L.svs_refqt_destroy.argtypes = [C.c_void_p]
            L.svs_refqt_insert.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
            L.svs_refqt_query.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.c_void_p, C.c_int]
            L.svs_refqt_is_window_empty.argtypes = [C.c_void_p] + [C.c_double] * 4
            cls._L = L
        return cls._L

    def __init__(self, w, h, delta=1.0, x=0.0, y=0.0):
        self.h = C.c_void_p(self.lib().svs_refqt_create(float(x), float(y), float(w), float(h), float(delta)))

    def insert(self, x, y, content):
        return self.lib().svs_refqt_insert(self.h, float(x), float(y), int(content))

    def query(self, x, y, w, h, cap=8192):
        out = np.zeros((cap, 3), np.int32)
        n = self.lib().svs_refqt_query(self.h, float(x), float(y), float(w), float(h), _p(out), cap)
        assert n <= cap
        return out[:n].copy()

    def is_window_empty(self, x, y, w, h):
        return bool(self.lib().svs_refqt_is_window_empty(self.h, float(x), float(y), float(w), float(h)))

    def __del__(self):
        try:
            self.lib().svs_refqt_destroy(self.h)
        except Exception:
            pass


def _ref_lib(name):
    so = os.path.join(_HERE, "_ref", name)
    if os.path.isdir("/root/reference/scavislam"):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    if not os.path.exists(so):
        raise FileNotFoundError(so)
    return C.CDLL(so)


class RefFastGrid:
    """oracle/_ref/libsvs_ref_fastgrid.so: the reference's own FastGrid (fast_grid.{h,cpp}) compiled on the host; its corner detector is bound
    to the oracle's FAST-9/16 restatement (svs_ref_fast9_16).  Test infrastructure."""
    _L = None

    @classmethod
    def lib(cls):
        if cls._L is None:
            L = _ref_lib("libsvs_ref_fastgrid.so")
            L.svs_reffg_create.restype = C.c_void_p
            L.svs_reffg_create.argtypes = [C.c_int] * 9
            L.svs_reffg_destroy.argtypes = [C.c_void_p]
            L.svs_reffg_set_fast.argtypes = [C.c_void_p]
            L.svs_reffg_detect_adaptively.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int]
            L.svs_reffg_detect.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_int]
            L.svs_reffg_cells.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            L.svs_reffg_set_fast(C.cast(lib().svs_ref_fast9_16, C.c_void_p))
            cls._L = L
        return cls._L

    def __init__(self, w, h, n_per_cell, boundary, fast_thr, gx, gy, fast_min=10, fast_max=40):
        self.w, self.h = int(w), int(h)
        self.g = C.c_void_p(self.lib().svs_reffg_create(self.w, self.h, int(n_per_cell), int(boundary), int(fast_thr), int(gx), int(gy), int(fast_min), int(fast_max)))

    def _dump(self, fn, img, *extra):
        img = np.ascontiguousarray(img)
        assert img.shape == (self.h, self.w) and img.dtype == np.uint8
        out = np.zeros((self.w * self.h, 3), np.int32)
        n = fn(self.g, _p(img), img.strides[0], self.w, self.h, *extra, _p(out), len(out))
        return out[:n].copy()

    def detect_adaptively(self, img, trials):
        return self._dump(self.lib().svs_reffg_detect_adaptively, img, int(trials))

    def detect(self, img):
        return self._dump(self.lib().svs_reffg_detect, img)

    def cells(self):
        out = np.zeros((256, 5), np.int32)
        n = self.lib().svs_reffg_cells(self.g, _p(out), 256)
        return out[:n].copy()

    def __del__(self):
        try:
            self.lib().svs_reffg_destroy(self.g)
        except Exception:
            pass


class RefFrontendGrids:
    """oracle/_ref/libsvs_ref_gate.so: the reference's own StereoFrontend::initialize / ::computeFastCorners / ::recomputeFastCorners (per-level
    FAST grids of the front end) around its own FastGrid; FAST-9/16 itself is bound to the oracle's restatement.  cams: 3 dicts (f, cx, cy, b, w, h)."""

    def __init__(self, cams, use_n_levels=-1):
        L = _ref_lib("libsvs_ref_gate.so")
        L.svs_reffe_create.restype = C.c_void_p
        L.svs_reffe_create.argtypes = [C.c_void_p, C.c_int]
        L.svs_reffe_destroy.argtypes = [C.c_void_p]
        L.svs_reffe_num_levels.argtypes = [C.c_void_p]
        L.svs_reffe_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.svs_reffe_compute_fast_corners.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.svs_reffe_recompute_fast_corners.argtypes = [C.c_void_p]
        L.svs_reffe_tree.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.svs_reffe_set_fast.argtypes = [C.c_void_p]
        L.svs_reffe_set_fast(C.cast(lib().svs_ref_fast9_16, C.c_void_p))
        self.L, self.cams = L, cams
        self.h = C.c_void_p(L.svs_reffe_create(_p(_cam6d(cams)), int(use_n_levels)))
        self.num_levels = L.svs_reffe_num_levels(self.h)

    def grid(self, level):
        """-> (dict of the FastGrid's members, cells [n, 5] = u0, u1, v0, v1, threshold)"""
        out = np.zeros(10, np.int32)
        cells = np.zeros((256, 5), np.int32)
        n = self.L.svs_reffe_grid(self.h, level, _p(out), _p(cells), 256)
        keys = ("num_features_per_cell", "boundary_per_cell", "min_inner", "min_outer", "max_inner", "max_outer", "fast_min", "fast_max", "gy", "gx")
        return dict(zip(keys, (int(v) for v in out))), cells[:n].copy()

    def compute_fast_corners(self, pyr, trials):
        self._pyr = [np.ascontiguousarray(a) for a in pyr]
        self.L.svs_reffe_compute_fast_corners(self.h, (C.c_void_p * 3)(*[a.ctypes.data for a in self._pyr]), (C.c_int * 3)(*[a.strides[0] for a in self._pyr]),
                                              int(trials))

    def recompute_fast_corners(self):
        self.L.svs_reffe_recompute_fast_corners(self.h)

    def tree(self, level):
        c = self.cams[level]
        out = np.zeros((c["w"] * c["h"], 3), np.int32)
        n = self.L.svs_reffe_tree(self.h, level, _p(out), len(out))
        return out[:n].copy()

    def __del__(self):
        try:
            self.L.svs_reffe_destroy(self.h)
        except Exception:
            pass


def ref_slamgraph_optimize(pose_ids, window_types, poses, point_ids, anchor_ids, xyz_anchor, obs_point, obs_pose, obs_level, obs_center,
                           pe_ids, pe_marginalized, pe_T12, pe_L12, pe_L21, cam, num_iters=2, use_robust_kernel=True, huber_kernel_width=1.0,
                           move=0.0, hip_branch=False):
    """The reference's own SlamGraph<SE3, StereoCamera, SE3XYZ_STEREO, 3>::optimize (oracle/_ref/libsvs_ref_slamgraph.so) on tables filled from the
    arguments, with a RECORDING g2o behind it.  Returns dict(vertices (kind, id, fixed, marginalized), estimates [n, 12], edges (kind, v0, v1, v2,
    robust, parameter id), edge_data [n, 49] (measurement 12, information 36, kernel delta), settings [10], poses_out, points_out)."""
    if hip_branch:
        # SlamGraph::optimize compiled with the SCAVISLAM_HIP_SUPPORT branch in place (oracle/Makefile: libsvs_hipbranch_slamgraph.so): no g2o graph is built, the
        # tables go to scavislam_amd/libscavislam_hip.so (the GPU) and come back; returns dict(poses_out, points_out, stats)
        from scavislam_amd import capi
        capi.load()
    L = _ref_lib("libsvs_hipbranch_slamgraph.so" if hip_branch else "libsvs_ref_slamgraph.so")
    pose_int = np.ascontiguousarray(np.stack([pose_ids, window_types], 1), np.int32)
    pose_T = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
    point_int = np.ascontiguousarray(np.stack([point_ids, anchor_ids], 1), np.int32)
    point_xyz = np.ascontiguousarray(xyz_anchor, np.float64).reshape(-1, 3)
    obs_int = np.ascontiguousarray(np.stack([obs_point, obs_pose, obs_level], 1), np.int32)
    obs_c = np.ascontiguousarray(obs_center, np.float64).reshape(-1, 3)
    n_pe = len(pe_ids)
    pe_int = np.ascontiguousarray(np.concatenate([np.asarray(pe_ids, np.int32).reshape(-1, 2), np.asarray(pe_marginalized, np.int32).reshape(-1, 1)], 1), np.int32)
    pe_dbl = np.ascontiguousarray(np.concatenate([np.asarray(pe_T12, np.float64).reshape(n_pe, 12), np.asarray(pe_L12, np.float64).reshape(n_pe, 36),
                                                  np.asarray(pe_L21, np.float64).reshape(n_pe, 36)], 1))
    cam6 = np.array([cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"]], np.float64)
    poses_out = np.zeros_like(pose_T); points_out = np.zeros_like(point_xyz)
    L.svs_refsg_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.svs_refsg_optimize(_p(pose_int), _p(pose_T), len(pose_int), _p(point_int), _p(point_xyz), len(point_int), _p(obs_int), _p(obs_c), len(obs_int),
                         _p(pe_int), _p(pe_dbl), n_pe, _p(cam6), int(num_iters), int(bool(use_robust_kernel)), float(huber_kernel_width), float(move),
                         _p(poses_out), _p(points_out))
    if hip_branch:
        i4, d3 = np.zeros(4, np.int32), np.zeros(3)
        L.svs_refsg_hip_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.svs_refsg_hip_stats(_p(i4), _p(d3))
        return dict(poses_out=poses_out, points_out=points_out,
                    stats=dict(iterations=int(i4[0]), trials=int(i4[1]), accepted=int(i4[2]), terminated=int(i4[3]), chi2_init=d3[0], chi2_final=d3[1], lambda_final=d3[2]))
    nv, ne = L.svs_refsg_num_vertices(), L.svs_refsg_num_edges()
    v_int = np.zeros((nv, 4), np.int32); v_est = np.zeros((nv, 12)); e_int = np.zeros((ne, 6), np.int32); e_dbl = np.zeros((ne, 49)); st = np.zeros(10)
    L.svs_refsg_get.argtypes = [C.c_void_p] * 5
    L.svs_refsg_get(_p(v_int), _p(v_est), _p(e_int), _p(e_dbl), _p(st))
    return dict(vertices=v_int, estimates=v_est, edges=e_int, edge_data=e_dbl, settings=st, poses_out=poses_out, points_out=points_out)


def ref_match_and_track(kf_pyrs, kf_poses, actkey_index, neighbours, T_cur_from_actkey, cur_pyr, disp, corners, cams, pts, list_of, num_max_points=0):
    """The reference's own StereoFrontend::matchAndTrack (oracle/_ref/libsvs_ref_track.so).  neighbours: [(keyframe index, strength)] of the active
    keyframe; list_of[i]: -1 = neighbourhood point list, k = new-point list of keyframe k.  Returns (ok, T 3x4, num_new_feat_matched, point index /
    obs / xyz_actkey of TrackData)."""
    n_kf = len(kf_pyrs)
    kfs = np.zeros(n_kf, KEYFRAME_DTYPE)
    keep = []
    for i, pyr in enumerate(kf_pyrs):
        kfs[i]["T_anchor_from_w"] = np.asarray(kf_poses[i], np.float64).reshape(12)
        for l in range(3):
            a = np.ascontiguousarray(pyr[l])
            keep.append(a)
            kfs[i]["pyr"][l] = a.ctypes.data
            kfs[i]["stride"][l] = a.strides[0]
    kf_ids = np.arange(100, 100 + 7 * n_kf, 7, dtype=np.int32)
    nb = np.ascontiguousarray(np.asarray(neighbours, np.int32).reshape(-1, 2))
    cur = [np.ascontiguousarray(a) for a in cur_pyr]
    disp = np.ascontiguousarray(disp, np.float32)
    cor = [np.ascontiguousarray(c, np.float64).reshape(-1, 2) for c in corners]
    pts = np.ascontiguousarray(pts, CANDIDATE_DTYPE)
    lo = np.ascontiguousarray(list_of, np.int32)
    T = np.array(T_cur_from_actkey, np.float64).reshape(12).copy()
    n = len(pts)
    obs_point = np.zeros(n, np.int32); obs = np.zeros((n, 3)); xyz = np.zeros((n, 3))
    num_new, n_obs = C.c_int(0), C.c_int(0)
    L = _ref_lib("libsvs_ref_track.so")
    L.svs_reftrack_match_and_track.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
    ok = L.svs_reftrack_match_and_track(_p(kfs), n_kf, _p(kf_ids), int(actkey_index), _p(nb), len(nb), _p(T), (C.c_void_p * 3)(*[a.ctypes.data for a in cur]),
                                        (C.c_int * 3)(*[a.strides[0] for a in cur]), _p(disp), disp.strides[0] // 4, (C.c_void_p * 3)(*[c.ctypes.data for c in cor]),
                                        (C.c_int * 3)(*[len(c) for c in cor]), cams, _p(pts), _p(lo), n, int(num_max_points), C.byref(num_new), C.byref(n_obs),
                                        _p(obs_point), _p(obs), _p(xyz))
    k = n_obs.value
    return bool(ok), T.reshape(3, 4), num_new.value, obs_point[:k].copy(), obs[:k].copy(), xyz[:k].copy()


def ref_process_frame(kf_pyrs, kf_poses, actkey_index, neighbours, cams, pts, list_of, T_cur_from_actkey, clouds, prev_pyr, cur_pyr, cur_f32, cur_dx, cur_dy, disp,
                      cuda_build=False, hip_branch=False):
    """The reference's own StereoFrontend::processFrame (oracle/_ref/libsvs_ref_frame.so: dense tracking, grid FAST with 6 trials on fresh grids, matchAndTrack,
    processMatchedPoints, dense cloud; the keyframe decisions answer "no").  Returns dict(ok, T, clouds, rimg, lines [per level: rows (is_new, uv_pyr 2,
    curkey_uv_pyr 2)], av_track_length, is_frame_dropped).  cuda_build: the reference's CUDA build of the path (libsvs_ref_frame_cuda.so: denseTrackingGpu on
    the emulated kernels, matcher radius 4) -- then clouds are full-resolution [h][w][4] and prev_pyr is the previous frame's F32 pyramid."""
    if hip_branch:
        # the reference's processFrame compiled with the SCAVISLAM_HIP_SUPPORT branch in place (oracle/Makefile: libsvs_hipbranch_frame.so): its arithmetic runs
        # in scavislam_amd/libscavislam_hip.so on the GPU -- load that library first (through torch's HIP runtime), the branch library binds to it
        from scavislam_amd import capi
        capi.load()
    L = _ref_lib(("libsvs_hipbranch_frame_cuda.so" if cuda_build else "libsvs_hipbranch_frame.so") if hip_branch else
                 ("libsvs_ref_frame_cuda.so" if cuda_build else "libsvs_ref_frame.so"))
    L.svs_refframe_set_fast.argtypes = [C.c_void_p]
    L.svs_refframe_set_fast(C.cast(lib().svs_ref_fast9_16, C.c_void_p))
    n_kf = len(kf_pyrs)
    kfs = np.zeros(n_kf, KEYFRAME_DTYPE)
    keep = []
    for i, pyr in enumerate(kf_pyrs):
        kfs[i]["T_anchor_from_w"] = np.asarray(kf_poses[i], np.float64).reshape(12)
        for l in range(3):
            a = np.ascontiguousarray(pyr[l])
            keep.append(a)
            kfs[i]["pyr"][l] = a.ctypes.data
            kfs[i]["stride"][l] = a.strides[0]
    kf_ids = np.arange(100, 100 + 7 * n_kf, 7, dtype=np.int32)
    nb = np.ascontiguousarray(np.asarray(neighbours, np.int32).reshape(-1, 2))
    pts = np.ascontiguousarray(pts, CANDIDATE_DTYPE)
    lo = np.ascontiguousarray(list_of, np.int32)
    T = np.array(T_cur_from_actkey, np.float64).reshape(12).copy()
    clouds = [np.ascontiguousarray(c, np.float32).copy() for c in clouds]
    prev = [np.ascontiguousarray(a, np.float32 if cuda_build else np.uint8) for a in prev_pyr]
    cur = [np.ascontiguousarray(a, np.uint8) for a in cur_pyr]
    f32, dx, dy = [[np.ascontiguousarray(a, np.float32) for a in lst] for lst in (cur_f32, cur_dx, cur_dy)]
    disp = np.ascontiguousarray(disp, np.float32)
    rimg = [np.zeros_like(c) for c in clouds]
    cap = len(pts) + 1
    lines = np.zeros((cap, 5)); n_lines = (C.c_int * 3)()
    av = C.c_double(0); dropped = C.c_int(0)
    P3 = C.c_void_p * 3
    L.svs_refframe_process_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + \
        [C.c_void_p] * 8 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    ok = L.svs_refframe_process_frame(_p(kfs), n_kf, _p(kf_ids), int(actkey_index), _p(nb), len(nb), cams, _p(pts), _p(lo), len(pts), _p(T),
                                      P3(*[a.ctypes.data for a in clouds]), P3(*[a.ctypes.data for a in prev]), P3(*[a.ctypes.data for a in cur]),
                                      P3(*[a.ctypes.data for a in f32]), P3(*[a.ctypes.data for a in dx]), P3(*[a.ctypes.data for a in dy]), _p(disp),
                                      P3(*[a.ctypes.data for a in rimg]), _p(lines), cap, n_lines, C.byref(av), C.byref(dropped))
    out_lines, k = [], 0
    for l in range(3):
        out_lines.append(lines[k:k + n_lines[l]].copy()); k += n_lines[l]
    return dict(ok=bool(ok), T=T.reshape(3, 4), clouds=clouds, rimg=rimg, lines=out_lines, av_track_length=av.value, is_frame_dropped=bool(dropped.value))


class RefSequence:
    """BASELINE configs[0]: ONE StereoFrontend of the reference alive across a sequence of frames, driven the way stereo_slam.cpp's main loop drives it
    (oracle/_ref/libsvs_ref_seq.so = the reference's CPU build; hip_branch=True: libsvs_hipbranch_seq.so = the same translation unit compiled with the
    SCAVISLAM_HIP_SUPPORT branch in place, its arithmetic on the GPU).  Nothing of the keyframe logic is stubbed: processFirstFrame, processFrame, shallWeSwitchKeyframe,
    shallWeDropNewKeyframe, addNewKeyframe, addNewPoints / addMorePoints, recomputeFastCorners are the reference's lines (stereo_frontend.cpp:39-528,656-1065)."""

    def __init__(self, cams, hip_branch=False, use_n_levels=3, sample_seed=2011, one_call=False, stereo_input=False, heap_jitter_seed=0):
        """one_call (with hip_branch): libsvs_hipbranch_seq_onecall.so -- processFrame's body from the dense tracker to the return of matchAndTrack is ONE
        svs_frontend_process_frame (the binding bench.py times); otherwise one library call per switch point of the reference"""
        if hip_branch:
            from scavislam_amd import capi
            capi.load()
        # stereo_input: the New College kind of input -- left + right image, no disparity; the "stereo" stage is the reference's calcDisparityCpu (stereo_frontend.cpp:
        # 620-653) in both builds: cv::StereoBM -> the oracle's block matcher (CPU build) / the HIP branch at its head (libsvs_hipbranch_seq_bm.so)
        self.stereo_input = stereo_input
        if stereo_input:
            assert not one_call
            name = "libsvs_hipbranch_seq_bm.so" if hip_branch else "libsvs_ref_seq_bm.so"
        else:
            name = ("libsvs_hipbranch_seq_onecall.so" if one_call else "libsvs_hipbranch_seq.so") if hip_branch else "libsvs_ref_seq.so"
        self.L = L = _ref_lib(name)
        L.svs_refseq_create.restype = C.c_void_p
        L.svs_refseq_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint]
        L.svs_refseq_destroy.argtypes = [C.c_void_p]
        L.svs_refseq_push_frame.argtypes = [C.c_void_p] * 6
        if stereo_input:
            L.svs_refseq_push_right.argtypes = [C.c_void_p] * 3
        L.svs_refseq_step.argtypes = [C.c_void_p] * 4
        L.svs_refseq_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.svs_refseq_fast_thresholds.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.svs_refseq_new_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.svs_refseq_recompute_fast_corners.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.svs_refseq_set_var.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.svs_refseq_nudge.argtypes = [C.c_void_p, C.c_double]
        self.cams = cams
        L.svs_refseq_heap_jitter.argtypes = [C.c_uint]
        L.svs_refseq_heap_jitter(int(heap_jitter_seed))      # 0 (default): the arena's fixed addresses; else: another run's heap (yardstick, see heap_jitter)
        self.h = L.svs_refseq_create(cams, use_n_levels, C.cast(lib().svs_ref_fast9_16, C.c_void_p), sample_seed)

    def set_var(self, name, value):
        """a live pangolin::Var of the reference (ui.parallax_thr, ui.num_max_points, ui.max_reproj_error, ui.min_num_points ...); "svs.<option>": an option of the
        HIP context behind the branch (svs_ctx_set_option), ignored by the CPU build"""
        self.L.svs_refseq_set_var(self.h, name.encode(), float(value))

    def step(self, img_u8, disp, right_u8=None):
        """FrameGrabber::processNextFrame (pyramid + f32 / Sobel images by the oracle's restatement of the OpenCV calls) + processFirstFrame / processFrame.
        Returns dict(ok, dropped, actkey_id, switched, n_keyframes, id_counter, n_neighbourhood_points, n_new_points, n_vertices, T, av_track_length, lines, fast_thr)."""
        pyr = build_pyramid(img_u8)
        fl = [convert_sobel(q) for q in pyr]
        P3 = C.c_void_p * 3
        keep = [np.ascontiguousarray(a) for a in pyr] + [np.ascontiguousarray(f[k]) for k in range(3) for f in fl]
        d = np.ascontiguousarray(disp, np.float32)
        self.L.svs_refseq_push_frame(self.h, P3(*[a.ctypes.data for a in keep[:3]]), P3(*[a.ctypes.data for a in keep[3:6]]), P3(*[a.ctypes.data for a in keep[6:9]]),
                                     P3(*[a.ctypes.data for a in keep[9:12]]), _p(d))
        if self.stereo_input:      # the right image instead of a disparity (disp is ignored: the front end computes its own)
            r8 = np.ascontiguousarray(right_u8, np.uint8)
            self.L.svs_refseq_push_right(self.h, _p(r8), C.cast(lib().svs_ref_stereo_bm, C.c_void_p))
        info = np.zeros(8, np.int32); T = np.zeros(12); av = C.c_double(0)
        import time as _time
        _t0 = _time.perf_counter()
        ok = self.L.svs_refseq_step(self.h, _p(info), _p(T), C.byref(av))
        self.last_step_s = _time.perf_counter() - _t0      # processFirstFrame / processFrame alone (bench.py: cpu_baseline.reference_compiled)
        cap = 8192
        lines = np.zeros((cap, 5)); n_lines = (C.c_int * 3)()
        m = self.L.svs_refseq_lines(self.h, _p(lines), cap, n_lines)
        assert m <= cap
        out_lines, k = [], 0
        for l in range(3):
            out_lines.append(lines[k:k + n_lines[l]].copy()); k += n_lines[l]
        thr = np.zeros(64, np.int32)
        nt = self.L.svs_refseq_fast_thresholds(self.h, _p(thr), 64)
        return dict(ok=bool(ok), dropped=bool(info[0]), actkey_id=int(info[1]), switched=bool(info[2]), n_keyframes=int(info[3]), id_counter=int(info[4]),
                    n_neighbourhood_points=int(info[5]), n_new_points=int(info[6]), n_vertices=int(info[7]), T=T.reshape(3, 4), av_track_length=av.value,
                    lines=out_lines, fast_thr=thr[:nt].copy())

    def onecall_record(self, cap=16384):
        """(one_call build) what the step just made handed to svs_frontend_process_frame and what came back: dict(kept_slot, kept_pose, T_guess, T_act, pts, group_end,
        res (FrameResult), matches, recloud / recloud_pose: the cloud was made again behind the call because the keyframe logic changed the pose), or None if the step made no call (the first frame)"""
        from scavislam_amd import capi
        from scavislam_amd.ctypes_types import CANDIDATE_DTYPE, MATCH_RESULT_DTYPE
        self.L.svs_refseq_onecall_record.argtypes = [C.c_void_p] * 3 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        meta = np.zeros(5, np.int32); poses = np.zeros(48); pts = np.zeros(cap, CANDIDATE_DTYPE); ge = np.zeros(64, np.int32)
        res = capi.FrameResult(); m = np.zeros(cap, MATCH_RESULT_DTYPE)
        rc = self.L.svs_refseq_onecall_record(self.h, _p(meta), _p(poses), _p(pts), cap, _p(ge), C.byref(res), _p(m))
        assert rc >= 0, "candidate capacity"
        if rc == 0:
            return None
        n, ng = int(meta[1]), int(meta[2])
        return dict(kept_slot=int(meta[3]), kept_pose=poses[:12].copy(), T_guess=poses[12:24].copy(), T_act=poses[24:36].copy(), pts=pts[:n].copy(), group_end=ge[:ng].copy(),
                    res=res, matches=m[:n].copy(), recloud=bool(meta[4]), recloud_pose=poses[36:48].copy())

    def heap_jitter(self, seed):
        """(CPU build) from now on the objects the reference orders by heap address land at pseudo-randomly padded addresses: another run of the same binary (yardstick)"""
        self.L.svs_refseq_heap_jitter(int(seed))

    def nudge(self, rel):
        """scales the translation of T_cur_from_actkey by (1 + rel) between two frames (the yardstick of tests/test_gpu_sequence.py)"""
        self.L.svs_refseq_nudge(self.h, float(rel))

    def new_points(self, keyframe_id, cap=4096):
        ids = np.zeros((cap, 2), np.int32); val = np.zeros((cap, 6))
        n = self.L.svs_refseq_new_points(self.h, int(keyframe_id), _p(ids), _p(val), cap)
        return ids[:n].copy(), val[:n].copy()

    def recompute_fast_corners(self, keyframe_id, level, cap=8192):
        xy = np.zeros((cap, 3))
        n = self.L.svs_refseq_recompute_fast_corners(self.h, int(keyframe_id), int(level), _p(xy), cap)
        return None if n < 0 else xy[:n].copy()

    def close(self):
        if self.h:
            self.L.svs_refseq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _cam6(cams):
    return np.ascontiguousarray([[c.f, c.cx, c.cy, c.b, c.w, c.h] for c in cams], np.float64)


def ref_dense_tracking_cpu(clouds, prev_pyr, cur_f, dx_f, dy_f, cams, T):
    """the reference's own DenseTracker::denseTrackingCpu (oracle/_ref/libsvs_ref_dense.so) -> (T, residual images)"""
    L = _ref_lib("libsvs_ref_dense.so")
    clouds = [np.ascontiguousarray(a, np.float32).copy() for a in clouds]
    prev = [np.ascontiguousarray(a) for a in prev_pyr]
    cur = [np.ascontiguousarray(a, np.float32) for a in cur_f]
    dx = [np.ascontiguousarray(a, np.float32) for a in dx_f]
    dy = [np.ascontiguousarray(a, np.float32) for a in dy_f]
    rimg = [np.tile(np.array([0, 0, 0, 1], np.float32), (*c.shape[:2], 1)) for c in clouds]      # the constructor's fill (dense_tracking.cpp:54)
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    T = np.array(T, np.float64).reshape(12).copy()
    cam6 = _cam6(cams)
    L.svs_refdense_track.argtypes = [C.c_void_p] * 10
    L.svs_refdense_track(P3(*[a.ctypes.data for a in clouds]), P3(*[a.ctypes.data for a in prev]), I3(*[a.strides[0] for a in prev]),
                         P3(*[a.ctypes.data for a in cur]), P3(*[a.ctypes.data for a in dx]), P3(*[a.ctypes.data for a in dy]),
                         I3(*[a.strides[0] for a in cur]), _p(cam6), _p(T), P3(*[a.ctypes.data for a in rimg]))
    return T.reshape(3, 4), rimg


def _cam6d(cams):
    return np.ascontiguousarray([[c["f"], c["cx"], c["cy"], c["b"], c["w"], c["h"]] for c in cams], np.float64)


def ref_dense_tracking_gpu(cloud, prev, cur, dx, dy, cams, T):
    """The reference's own DenseTracker::denseTrackingGpu host loop (dense_tracking.cpp:60-193) around its own emulated kernels
    (oracle/_ref/libsvs_ref_densegpu.so).  cams: list of 3 dicts (f, cx, cy, b, w, h).  Returns (T 3x4, residual images [h][w][4])."""
    L = _ref_lib("libsvs_ref_densegpu.so")
    cloud = [np.ascontiguousarray(a, np.float32) for a in cloud]
    prev, cur, dx, dy = [[np.ascontiguousarray(a, np.float32) for a in lst] for lst in (prev, cur, dx, dy)]
    for l, c in enumerate(cams):
        assert cloud[l].shape == (c["h"], c["w"], 4) and prev[l].shape == cur[l].shape == dx[l].shape == dy[l].shape == (c["h"], c["w"])
    rimg = [np.zeros((c["h"], c["w"], 4), np.float32) for c in cams]
    P3 = C.c_void_p * 3
    T = np.array(T, np.float64).reshape(12).copy()
    cam6 = _cam6d(cams)
    L.svs_refdg_dense_tracking_gpu.argtypes = [C.c_void_p] * 8
    L.svs_refdg_dense_tracking_gpu(_p(cam6), P3(*[a.ctypes.data for a in cloud]), P3(*[a.ctypes.data for a in prev]), P3(*[a.ctypes.data for a in cur]),
                                   P3(*[a.ctypes.data for a in dx]), P3(*[a.ctypes.data for a in dy]), _p(T), P3(*[a.ctypes.data for a in rimg]))
    return T.reshape(3, 4), rimg


def ref_pointcloud_gpu(disp, cams, T_cur_from_actkey):
    """The reference's own DenseTracker::computeDensePointCloudGpu (dense_tracking.cpp:195-216) -> the three level clouds"""
    L = _ref_lib("libsvs_ref_densegpu.so")
    disp = np.ascontiguousarray(disp, np.float32)
    assert disp.shape == (cams[0]["h"], cams[0]["w"])
    out = [np.zeros((c["h"], c["w"], 4), np.float32) for c in cams]
    T = np.ascontiguousarray(T_cur_from_actkey, np.float64).reshape(12)
    cam6 = _cam6d(cams)
    L.svs_refdg_pointcloud_gpu.argtypes = [C.c_void_p] * 4
    L.svs_refdg_pointcloud_gpu(_p(cam6), _p(disp), _p(T), (C.c_void_p * 3)(*[a.ctypes.data for a in out]))
    return out


def ref_pointcloud_cpu(disp, cams, T_cur_from_actkey):
    """the reference's own DenseTracker::computeDensePointCloudCpu -> the three quarter-grid clouds"""
    L = _ref_lib("libsvs_ref_dense.so")
    disp = np.ascontiguousarray(disp, np.float32)
    cam6 = _cam6(cams)
    clouds = [np.zeros((c.h // 4, c.w // 4, 4), np.float32) for c in cams]
    T = np.ascontiguousarray(T_cur_from_actkey, np.float64).reshape(12)
    L.svs_refdense_cloud.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.svs_refdense_cloud(_p(disp), disp.strides[0], disp.shape[1], disp.shape[0], _p(cam6), _p(T), (C.c_void_p * 3)(*[a.ctypes.data for a in clouds]))
    return clouds


def ref_edges_lib():
    """oracle/_ref/libsvs_ref_edges.so: the reference's own BA edge / vertex code (anchored_points.{h,cpp} slices + transformations.h:62-95)"""
    L = _ref_lib("libsvs_ref_edges.so")
    L.svs_refedge_psi2uvu.argtypes = [C.c_void_p] * 9
    L.svs_refedge_se3.argtypes = [C.c_void_p] * 6
    L.svs_refvertex_oplus_se3.argtypes = [C.c_void_p] * 3
    L.svs_refvertex_oplus_xyz.argtypes = [C.c_void_p] * 3
    L.svs_refcam_uvu.argtypes = [C.c_void_p] * 3
    return L


def ref_matcher_lib():
    """oracle/_ref/libsvs_ref_matcher.so: the reference's own GuidedMatcher<StereoCamera> (matcher.cpp:31-459 + matcher-impl.cpp:30-51)."""
    L = _ref_lib("libsvs_ref_matcher.so")
    L.svs_refznssd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.svs_refznssd_patch_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.svs_refmatch_warp_affine.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.svs_refmatch_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]
    L.svs_refmatch_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    return L


def ref_warp_affine(frame, T, depth, key_uv, cam, halfpatch=5):
    """The reference's own GuidedMatcher::warpAffinve (matcher.cpp:403-458); same arguments as warp_affine()."""
    frame = np.ascontiguousarray(frame)
    T = np.ascontiguousarray(T, np.float64).reshape(12)
    kuv = np.ascontiguousarray(key_uv, np.float64)
    out = np.zeros((2 * halfpatch, 2 * halfpatch), np.uint8)
    ref_matcher_lib().svs_refmatch_warp_affine(_p(frame), frame.strides[0], _p(T), float(depth), _p(kuv), C.byref(cam), halfpatch, _p(out))
    return out


def match_candidates(cur_img, cam, cand_xyc, key, sumA, sumAA, init_dist, ref=False, level=0):
    """matchCandidates (matcher.cpp:144-181) over (x, y, content) triples in list order -> (min_dist, index, u, v); ref=True: the
    reference's own function."""
    cur = np.ascontiguousarray(cur_img)
    cand = np.ascontiguousarray(cand_xyc, np.int32).reshape(-1, 3)
    key = np.ascontiguousarray(key, np.uint8).reshape(64)
    out = np.zeros(4, np.int32)
    if ref:
        ref_matcher_lib().svs_refmatch_candidates(_p(cur), cur.strides[0], C.byref(cam), level, _p(cand), len(cand), _p(key), int(sumA), int(sumAA),
                                                  int(init_dist), _p(out))
    else:
        L = lib()
        L.svs_ref_match_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.svs_ref_match_candidates(_p(cur), cur.strides[0], C.byref(cam), _p(cand), len(cand), _p(key), int(sumA), int(sumAA), int(init_dist), _p(out))
    return tuple(int(v) for v in out)


def ref_match(kf_pyrs, kf_poses, T_cur_from_actkey, actkey_index, cur_pyr, disp, corners, cams, pts, radius=8, thr_mean=22, thr_std=10):
    """The reference's own GuidedMatcher<StereoCamera>::match.  corners: per level an [n, 2] array, inserted into that level's QuadTree in
    this order; the active keyframe is kf_poses[actkey_index].  Returns (point index, obs uvu, xyz_actkey) per appended observation, in
    TrackData order."""
    n_kf = len(kf_pyrs)
    kfs = np.zeros(n_kf, KEYFRAME_DTYPE)
    keep = []
    for i, pyr in enumerate(kf_pyrs):
        kfs[i]["T_anchor_from_w"] = np.asarray(kf_poses[i], np.float64).reshape(12)
        for l in range(3):
            a = np.ascontiguousarray(pyr[l])
            keep.append(a)
            kfs[i]["pyr"][l] = a.ctypes.data
            kfs[i]["stride"][l] = a.strides[0]
    kf_ids = np.arange(100, 100 + 7 * n_kf, 7, dtype=np.int32)      # arbitrary keyframe ids: the reference looks them up in hash maps
    cur = [np.ascontiguousarray(a) for a in cur_pyr]
    cur_ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in cur])
    cur_strides = (C.c_int * 3)(*[a.strides[0] for a in cur])
    disp = np.ascontiguousarray(disp, np.float32)
    cor = [np.ascontiguousarray(c, np.float64).reshape(-1, 2) for c in corners]
    cor_ptrs = (C.c_void_p * 3)(*[c.ctypes.data for c in cor])
    cor_n = (C.c_int * 3)(*[len(c) for c in cor])
    pts = np.ascontiguousarray(pts, CANDIDATE_DTYPE)
    Tc = np.ascontiguousarray(T_cur_from_actkey, np.float64).reshape(12)
    n = len(pts)
    obs_point = np.zeros(n, np.int32); obs = np.zeros((n, 3)); xyz = np.zeros((n, 3))
    k = ref_matcher_lib().svs_refmatch_match(_p(kfs), n_kf, _p(kf_ids), int(kf_ids[actkey_index]), _p(Tc), cur_ptrs, cur_strides, _p(disp),
                                             disp.strides[0] // 4, disp.shape[1], disp.shape[0], cor_ptrs, cor_n, cams, _p(pts), n, radius,
                                             thr_mean, thr_std, _p(obs_point), _p(obs), _p(xyz))
    return obs_point[:k].copy(), obs[:k].copy(), xyz[:k].copy()


_REF = None


def ref_gpu():
    global _REF
    if _REF is None:
        _REF = RefGpu()
    return _REF
