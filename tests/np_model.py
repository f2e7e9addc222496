"""Independent NumPy/SciPy float64 models used ONLY to cross-check the C oracle (which itself has no
reference ground truth: "parity unpinned", see oracle/svs_oracle.h).  Written from the definitions
(SURVEY.md Appendix A), sharing no code with oracle/*.c:

  fast9_mask / fast_score_map  FAST-9/16 segment test from its definition (vectorised over the image)
  pyr_down                     5x5 binomial via scipy.ndimage (mirror = REFLECT_101), (s+128)>>8
  se3_exp / se3_log            matrix exponential / logarithm of the 4x4 twist (scipy.linalg)
  ba_dense_step                one damped Gauss-Newton step on the FULL (poses+points) normal
                               equations assembled from numerically differentiated residuals
"""
import numpy as np
from scipy import linalg, ndimage

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _ring_stack(img):
    h, w = img.shape
    c = img[3:h - 3, 3:w - 3].astype(np.int32)
    ring = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int32) for dx, dy in RING])
    return c, ring


def fast9_mask(img, t):
    """Boolean map (valid region rows/cols 3..n-4) of FAST-9/16 corners at threshold t."""
    c, ring = _ring_stack(img)
    darker = ring < c - t
    brighter = ring > c + t
    out = np.zeros(c.shape, bool)
    for flags in (darker, brighter):
        ext = np.concatenate([flags, flags[:8]], 0)
        for s in range(16):
            out |= ext[s:s + 9].all(0)
    return out


def fast_corners(img, t):
    m = fast9_mask(img, t)
    ys, xs = np.nonzero(m)          # row-major
    return np.stack([xs + 3, ys + 3], 1).astype(np.int16)


def fast_score_map(img):
    """max t such that pixel is a corner at t (-1 if never), from the arc definition."""
    c, ring = _ring_stack(img)
    best = np.full(c.shape, -10 ** 6)
    for d in (c[None] - ring, ring - c[None]):
        ext = np.concatenate([d, d[:8]], 0)
        for s in range(16):
            best = np.maximum(best, ext[s:s + 9].min(0))
    return np.maximum(best - 1, -1)


def pyr_down(img):
    k = np.array([1, 4, 6, 4, 1], np.int64)
    a = img.astype(np.int64)
    a = ndimage.correlate1d(a, k, axis=1, mode="mirror")
    a = ndimage.correlate1d(a, k, axis=0, mode="mirror")
    return ((a[::2, ::2] + 128) >> 8).astype(np.uint8)


def sobel(img_f32):
    p = np.pad(img_f32, 1, mode="reflect")
    return p[1:-1, 2:] - p[1:-1, :-2], p[2:, 1:-1] - p[:-2, 1:-1]


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def se3_exp(x):
    """exp of the twist (upsilon, omega), translation part first; returns 3x4."""
    M = np.zeros((4, 4))
    M[:3, :3] = hat(x[3:])
    M[:3, 3] = x[:3]
    return linalg.expm(M)[:3, :]


def se3_log(T):
    M = np.eye(4)
    M[:3, :] = T
    L = np.real(linalg.logm(M))
    return np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]])


def pose_mul(A, B):
    R = A[:, :3] @ B[:, :3]
    return np.hstack([R, (A[:, :3] @ B[:, 3] + A[:, 3])[:, None]])


def pose_inv(A):
    return np.hstack([A[:, :3].T, (-A[:, :3].T @ A[:, 3])[:, None]])


def stereo_residual(psi, T_obs, T_anc, obs, cam):
    xa = np.array([psi[0], psi[1], 1.0]) / psi[2]
    T = pose_mul(T_obs, pose_inv(T_anc))
    y = T[:, :3] @ xa + T[:, 3]
    f, cx, cy, b = cam
    return obs - np.array([f * y[0] / y[2] + cx, f * y[1] / y[2] + cy, f * (y[0] - b) / y[2] + cx])


def huber(e2, delta):
    if e2 <= delta * delta:
        return e2, 1.0
    s = np.sqrt(e2)
    return 2 * s * delta - delta * delta, delta / s


def ba_chi2(poses, psi, edges, cons, cam, delta=1.0, robust=True):
    chi = 0.0
    for e in edges:
        r = stereo_residual(psi[e["point"]], poses[e["pose"]].reshape(3, 4), poses[e["anchor"]].reshape(3, 4), e["obs"], cam)
        e2 = float(np.sum(r * r * e["info"]))
        chi += huber(e2, delta)[0] if robust else e2
    for c in cons:
        T = pose_mul(pose_mul(c["T_21"].reshape(3, 4), poses[c["pose1"]].reshape(3, 4)), pose_inv(poses[c["pose2"]].reshape(3, 4)))
        r = se3_log(T)
        chi += float(r @ c["info"].reshape(6, 6) @ r)
    return chi


def _perturb_pose(T, d):
    return pose_mul(se3_exp(d), T)


def ba_dense_step(poses, psi, edges, cons, cam, lam, delta=1.0, robust=True, eps=1e-6, exact_self=True):
    """Solve the FULL damped normal equations (H + lam I) x = b with numerically differentiated
    residuals (central differences), i.e. without any Schur algebra.  Returns (x_poses[P,6], x_psi[L,3])."""
    P, L = len(poses), len(psi)
    n = 6 * P + 3 * L
    H = np.zeros((n, n))
    b = np.zeros(n)
    for e in edges:
        l, i, a = int(e["point"]), int(e["pose"]), int(e["anchor"])

        def res(dp, di, da):
            To = _perturb_pose(poses[i].reshape(3, 4), di)
            Ta = _perturb_pose(poses[a].reshape(3, 4), da) if a != i else _perturb_pose(poses[a].reshape(3, 4), di)
            return stereo_residual(psi[l] + dp, To, Ta, e["obs"], cam)
        r0 = res(np.zeros(3), np.zeros(6), np.zeros(6))
        J = np.zeros((3, n))
        for k in range(3):
            d = np.zeros(3); d[k] = eps
            J[:, 6 * P + 3 * l + k] = (res(d, np.zeros(6), np.zeros(6)) - res(-d, np.zeros(6), np.zeros(6))) / (2 * eps)
        for k in range(6):
            d = np.zeros(6); d[k] = eps
            J[:, 6 * i + k] += (res(np.zeros(3), d, np.zeros(6)) - res(np.zeros(3), -d, np.zeros(6))) / (2 * eps)
            if a != i:
                J[:, 6 * a + k] += (res(np.zeros(3), np.zeros(6), d) - res(np.zeros(3), np.zeros(6), -d)) / (2 * eps)
        e2 = float(np.sum(r0 * r0 * e["info"]))
        w = huber(e2, delta)[1] if robust else 1.0
        Om = np.diag(w * e["info"])
        H += J.T @ Om @ J
        b += -J.T @ Om @ r0
    for c in cons:
        i, j = int(c["pose1"]), int(c["pose2"])

        def res(d1, d2):
            T = pose_mul(pose_mul(c["T_21"].reshape(3, 4), _perturb_pose(poses[i].reshape(3, 4), d1)),
                         pose_inv(_perturb_pose(poses[j].reshape(3, 4), d2)))
            return se3_log(T)
        r0 = res(np.zeros(6), np.zeros(6))
        J = np.zeros((6, n))
        for k in range(6):
            d = np.zeros(6); d[k] = eps
            J[:, 6 * i + k] = (res(d, np.zeros(6)) - res(-d, np.zeros(6))) / (2 * eps)
            J[:, 6 * j + k] = (res(np.zeros(6), d) - res(np.zeros(6), -d)) / (2 * eps)
        Om = c["info"].reshape(6, 6)
        H += J.T @ Om @ J
        b += -J.T @ Om @ r0
    used = np.ones(n, bool)
    seen = set(int(e["point"]) for e in edges)
    for l in range(L):
        if l not in seen:
            used[6 * P + 3 * l:6 * P + 3 * l + 3] = False
    Hd = H + lam * np.eye(n)
    x = np.zeros(n)
    x[used] = np.linalg.solve(Hd[np.ix_(used, used)], b[used])
    return x[:6 * P].reshape(P, 6), x[6 * P:].reshape(L, 3), H, b


# ---- stereo block matching: independent vectorised model of cv::StereoBM as used at stereo_frontend.cpp:620-653 ----
def stereo_prefilter(img, cap=31):
    """XSOBEL prefilter: 3x3 x-Sobel (reflect-101 rows), saturated to [0, 2 cap]; border columns / odd last row = cap."""
    a = img.astype(np.int32)
    h, w = a.shape
    p = np.pad(a, ((1, 1), (0, 0)), mode="reflect")
    gx = np.zeros_like(a)
    gx[:, 1:-1] = (p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])
    out = np.clip(gx, -cap, cap) + cap
    out[:, 0] = out[:, -1] = cap
    if h % 2:
        out[-1, :] = cap
    return out.astype(np.uint8)


def stereo_bm_core(lp, rp, ndisp=32, wsz=7, cap=31, texthr=10, uniq=15):
    """(disp16, cost) of findStereoCorrespondenceBM on prefiltered images, via an explicit cost volume."""
    h, w = lp.shape
    r = wsz // 2
    lofs, width1 = ndisp - 1, w - ndisp + 1
    L, R = lp.astype(np.int32), rp.astype(np.int32)
    disp16 = np.full((h, w), -16, np.int16)
    cost = np.zeros((h, w), np.int32)
    if width1 <= 0:
        return disp16, cost
    xs = np.arange(-r, width1 + r)                       # window columns relative to the output index x
    lcol = np.clip(xs, -lofs, w - lofs - 1) + lofs
    rcol = np.clip(xs, 0, w - 1)
    vol = np.empty((ndisp, h, len(xs)), np.int32)
    for d in range(ndisp):
        vol[d] = np.abs(L[:, lcol] - R[:, np.minimum(rcol + d, w - 1)])
    tex = np.abs(L[:, lcol] - cap)

    def box(a):                                          # 7x7 window sum, rows replicated, columns already extended
        a = np.pad(a, [(0, 0)] * (a.ndim - 2) + [(r, r), (0, 0)], mode="edge")
        c = np.cumsum(np.pad(a, [(0, 0)] * (a.ndim - 2) + [(1, 0), (1, 0)]), axis=-2).cumsum(axis=-1)
        return c[..., wsz:, wsz:] - c[..., :-wsz, wsz:] - c[..., wsz:, :-wsz] + c[..., :-wsz, :-wsz]

    sad = box(vol)                                       # [ndisp, h, width1]
    tsum = box(tex)
    mind = np.argmin(sad, axis=0)                        # first minimum
    minsad = np.take_along_axis(sad, mind[None], 0)[0]
    ok = tsum >= texthr
    if uniq > 0:
        thresh = minsad + minsad * uniq // 100
        dd = np.arange(ndisp)[:, None, None]
        rival = (sad <= thresh[None]) & ((dd < mind[None] - 1) | (dd > mind[None] + 1))
        ok &= ~rival.any(axis=0)
    ip = np.where(mind == ndisp - 1, ndisp - 2, mind + 1)
    im = np.where(mind == 0, 1, mind - 1)
    p = np.take_along_axis(sad, ip[None], 0)[0]
    n = np.take_along_axis(sad, im[None], 0)[0]
    den = p + n - 2 * minsad + np.abs(p - n)
    num = (p - n) * 256
    frac = np.where(den != 0, np.sign(num) * (np.abs(num) // np.maximum(den, 1)), 0)      # C division truncates toward zero
    val = ((ndisp - mind - 1) * 256 + frac + 15) >> 4
    disp16[:, lofs:lofs + width1] = np.where(ok, val, -16).astype(np.int16)
    cost[:, lofs:lofs + width1] = np.where(ok, minsad, 0)
    return disp16, cost


def stereo_validate(disp16, cost, ndisp=32, maxdiff=1):
    d16 = disp16.astype(np.int32).copy()
    h, w = d16.shape
    INV = -16
    for y in range(h):
        d = d16[y]
        d2 = np.full(w, INV, np.int64)
        c2 = np.full(w, np.iinfo(np.int64).max, np.int64)
        for x in range(ndisp, w):
            if d[x] == INV:
                continue
            x2 = x - ((d[x] + 8) >> 4)
            if 0 <= x2 < w and c2[x2] > cost[y, x]:
                c2[x2], d2[x2] = cost[y, x], d[x]
        out = d.copy()
        for x in range(ndisp, w):
            if d[x] == INV:
                continue
            bad = []
            for xx in (x - (d[x] >> 4), x - ((d[x] + 15) >> 4)):
                bad.append(0 <= xx < w and d2[xx] > INV and abs(d2[xx] - d[x]) > maxdiff * 16)
            if all(bad):
                out[x] = INV
        d16[y] = out
    return d16.astype(np.int16)


def stereo_filter_speckles(disp16, max_size=100, max_diff=32, new_val=-16):
    """Components of the pixel graph (4-neighbours whose values differ by <= max_diff); small ones -> new_val."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    d = disp16.astype(np.int32)
    h, w = d.shape
    idx = np.arange(h * w).reshape(h, w)
    valid = d != new_val
    eh = valid[:, :-1] & valid[:, 1:] & (np.abs(d[:, :-1] - d[:, 1:]) <= max_diff)
    ev = valid[:-1, :] & valid[1:, :] & (np.abs(d[:-1, :] - d[1:, :]) <= max_diff)
    rows = np.concatenate([idx[:, :-1][eh], idx[:-1, :][ev]])
    cols = np.concatenate([idx[:, 1:][eh], idx[1:, :][ev]])
    g = coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(h * w, h * w))
    _, lab = connected_components(g, directed=False)
    sizes = np.bincount(lab)
    small = (sizes[lab] <= max_size).reshape(h, w) & valid
    out = disp16.copy()
    out[small] = new_val
    return out


def stereo_bm(left, right):
    lp, rp = stereo_prefilter(left), stereo_prefilter(right)
    d16, cost = stereo_bm_core(lp, rp)
    d16 = stereo_validate(d16, cost)
    d16 = stereo_filter_speckles(d16)
    return d16.astype(np.float32) / 16.0


# ---- motion-only refinement: vectorised model of PoseOptimizer::calcFastMotionOnly (pose_optimizer.h:134-298) ----
def motion_only(xyz, obs, cam, T, robust=True, b=2.0, num_iter=15, tau=1e-5):
    """xyz [n,3] points in the active keyframe frame, obs [n,3] (u,v,u_right); cam = (f,cx,cy,baseline)."""
    f_, cx, cy, bl = cam

    def resid(T):
        p = xyz @ T[:, :3].T + T[:, 3]
        pred = np.stack([p[:, 0] / p[:, 2] * f_ + cx, p[:, 1] / p[:, 2] * f_ + cy, (p[:, 0] - bl) / p[:, 2] * f_ + cx], 1)
        return obs - pred, p

    def weighted(r):
        if not robust:
            return r
        nrm = np.maximum(1e-10, np.linalg.norm(r, axis=1))
        k = np.where(nrm < b, nrm * nrm, 2 * b * nrm - b * b)
        return r * (np.sqrt(k) / nrm)[:, None]

    def jac(p):
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        A = -f_ / z
        Cc, D, E = f_ * x / z ** 2, f_ * y / z ** 2, f_ * (x - bl) / z ** 2
        zero = np.zeros_like(x)
        J = np.stack([np.stack([A, zero, Cc, y * Cc, z * A - x * Cc, -y * A], 1),
                      np.stack([zero, A, D, -z * A + y * D, -x * D, x * A], 1),
                      np.stack([A, zero, E, y * E, z * A - x * E, -y * A], 1)], 1)
        return J      # [n,3,6]

    r, p = resid(T)
    J = jac(p)
    mu = tau * np.abs(np.einsum("nij,nij->nj", J, J)).max()
    chi2 = float((weighted(r) ** 2).sum())
    init = chi2
    nu, trial, stop = 2.0, 0, False
    for _ in range(num_iter):
        while True:
            r, p = resid(T)
            J = jac(p)
            A = mu * np.eye(6) + np.einsum("nij,nik->jk", J, J)
            B = -np.einsum("nij,ni->j", J, weighted(r))
            delta = linalg.solve(A, B, assume_a="sym")
            Tn = pose_mul(se3_exp(delta), T)
            new = float((weighted(resid(Tn)[0]) ** 2).sum())
            rho = chi2 - new
            if rho > 0:
                T, chi2 = Tn, new
                stop = np.abs(B).max() <= 1e-10
                mu *= max(1 / 3, 1 - (2 * rho - 1) ** 3)
                nu, trial = 2.0, 0
            else:
                mu *= nu
                nu *= 2
                trial += 1
                stop = trial == 5
            if rho > 0 or stop:
                break
        if stop:
            break
    return T, init, chi2


# ---- StereoFrontend::processMatchedPoints, numeric part (independent vectorised restatement) ----------------------------
def process_matched_points(xyz, obs, level, is_ok, n_new_records, cam, T, max_reproj_error=2.0):
    """xyz [n,3] points in the active keyframe, obs [n,3] (u, v, u_right) in level-0 pixels, level [n], is_ok [n] bool.
    Returns dict(accepted, is_new, uv_pyr, curkey_uv_pyr, grid2x2 [2,2], grid3x3 [3,3], per_level [3], n_track, sum_len)."""
    f, cx, cy, b, w, h = cam
    T = np.asarray(T, np.float64).reshape(3, 4)
    p = xyz @ T[:, :3].T + T[:, 3]
    pred = np.stack([p[:, 0] / p[:, 2] * f + cx, p[:, 1] / p[:, 2] * f + cy, (p[:, 0] - b) / p[:, 2] * f + cx], 1)
    d = np.abs(obs - pred)
    fac = (2.0 ** level).astype(np.float64)
    mre = np.float32(max_reproj_error)
    acc = is_ok & (d[:, 0] < np.float64(mre) * fac) & (d[:, 1] < np.float64(mre) * fac) & (d[:, 2] < 3.0 * np.float64(mre))
    third = np.float32(1.0 / 3.0)
    half_w, half_h = int(w * 0.5), int(h * 0.5)
    tw, th = int(np.float32(w) * third), int(np.float32(h) * third)
    ttw, tth = int(np.float32(w * 2) * third), int(np.float32(h * 2) * third)
    i2 = (obs[:, 0] >= half_w).astype(int); j2 = (obs[:, 1] >= half_h).astype(int)
    i3 = (obs[:, 0] >= tw).astype(int) + (obs[:, 0] >= ttw).astype(int)
    j3 = (obs[:, 1] >= th).astype(int) + (obs[:, 1] >= tth).astype(int)
    g2 = np.zeros((2, 2), int); g3 = np.zeros((3, 3), int); per = np.zeros(3, int)
    np.add.at(g2, (i2[acc], j2[acc]), 1)
    np.add.at(g3, (i3[acc], j3[acc]), 1)
    np.add.at(per, level[acc], 1)
    uv = obs[:, :2] / fac[:, None]
    ck = np.stack([xyz[:, 0] / xyz[:, 2] * f + cx, xyz[:, 1] / xyz[:, 2] * f + cy], 1) / fac[:, None]
    length = np.hypot(*(uv - ck).T)
    return dict(accepted=acc, is_new=acc & (np.arange(len(obs)) < n_new_records), uv_pyr=np.where(acc[:, None], uv, 0.0),
                curkey_uv_pyr=np.where(acc[:, None], ck, 0.0), grid2x2=g2, grid3x3=g3, per_level=per, n_track=int(acc.sum()),
                sum_len=float(length[acc].sum()))


# ---- dense tracker (dense_tracking.cpp:222-423), vectorised and independent of the C oracle -------------------------------
def _bilinear_f32(img, u, v):
    """maths_utils.cpp:46-65 interpolateMat_32f: float weights, 4 taps."""
    x, y = np.floor(u), np.floor(v)
    sx, sy = (u - x).astype(np.float32), (v - y).astype(np.float32)
    xi, yi = x.astype(int), y.astype(int)
    wx0, wy0 = np.float32(1) - sx, np.float32(1) - sy
    return (wx0 * wy0) * img[yi, xi] + (wx0 * sy) * img[yi + 1, xi] + (sx * wy0) * img[yi, xi + 1] + (sx * sy) * img[yi + 1, xi + 1]


def pointcloud_cpu(disp, cam, level, T_cur_from_actkey):
    """computeDensePointCloudCpu: quarter-grid cloud of one level; cam = (f, cx, cy, b, w, h) of THAT level."""
    f, cx, cy, b, w, h = cam
    cw, ch = w // 4, h // 4
    u, v = np.meshgrid(np.arange(cw) * 4, np.arange(ch) * 4)
    d = (disp[(v << level), (u << level)].astype(np.float64) * (1.0 / (1 << level))).astype(np.float32)
    Q = np.array([[1, 0, 0, -cx], [0, 1, 0, -cy], [0, 0, 0, f], [0, 0, 1.0 / b, 0]])
    T = np.vstack([np.asarray(T_cur_from_actkey, np.float64).reshape(3, 4), [0, 0, 0, 1]])
    TQ = np.linalg.inv(T) @ Q
    uvd = np.stack([u, v, d.astype(np.float64), np.ones_like(u, float)], -1)
    r = uvd @ TQ.T
    out = np.zeros((ch, cw, 4), np.float32)
    ok = d > 0
    with np.errstate(all="ignore"):
        out[..., :3] = np.where(ok[..., None], (r[..., :3] / r[..., 3:4]), 0.0).astype(np.float32)
    out[..., 3] = np.where(ok, 1.0, -1.0)
    return out


def dense_pass(cloud, prev_u8, cur, dx, dy, cam, T):
    """One H,b pass of denseTrackingCpu's loop body; returns H (6x6), b (6), chi2, n_valid, residual image."""
    f, cx, cy, b_, w, h = cam
    T = np.asarray(T, np.float64).reshape(3, 4)
    ch, cw = cloud.shape[:2]
    P = cloud[..., :3].astype(np.float64)
    has = cloud[..., 3] > 0
    X = P @ T[:, :3].T + T[:, 3]
    with np.errstate(all="ignore"):
        uv = np.stack([f * (X[..., 0] / X[..., 2]) + cx, f * (X[..., 1] / X[..., 2]) + cy], -1).astype(np.float32)
    ui, vi = np.trunc(uv[..., 0]), np.trunc(uv[..., 1])
    inside = has & np.isfinite(uv).all(-1) & (ui >= 2) & (vi >= 2) & (ui < w - 2) & (vi < h - 2)
    u = np.where(inside, uv[..., 0], np.float32(2)); v = np.where(inside, uv[..., 1], np.float32(2))
    vv, uu = np.meshgrid(np.arange(ch) * 4, np.arange(cw) * 4, indexing="ij")
    ip = ((1.0 / 255.0) * prev_u8[vv, uu].astype(np.float64)).astype(np.float32)
    ic = _bilinear_f32(cur, u, v)
    res = np.clip((ip - ic).astype(np.float32), np.float32(-0.1), np.float32(0.1))      # clamp +-0.1 (double constants, f32 values)
    res = np.where(res > 0.1, np.float32(0.1), np.where(res < -0.1, np.float32(-0.1), res)).astype(np.float32)
    gx = (0.5 * _bilinear_f32(dx, u, v).astype(np.float64)).astype(np.float32).astype(np.float64)
    gy = (0.5 * _bilinear_f32(dy, u, v).astype(np.float64)).astype(np.float32).astype(np.float64)
    x, y, z = X[..., 0], X[..., 1], np.where(inside, X[..., 2], 1.0)
    z2 = z * z
    r0 = np.stack([-1.0 / z * f, 0 * z, x / z2 * f, x * y / z2 * f, -(1 + x * x / z2) * f, y / z * f], -1)      # transformations.h:117-139
    r1 = np.stack([0 * z, -1.0 / z * f, y / z2 * f, (1 + y * y / z2) * f, -x * y / z2 * f, -x / z * f], -1)
    J = gx[..., None] * r0 + gy[..., None] * r1
    J = np.where(inside[..., None], J, 0.0)
    rr = np.where(inside, res, np.float32(0)).astype(np.float64)
    Jf = J.reshape(-1, 6)
    H = Jf.T @ Jf
    bvec = Jf.T @ rr.reshape(-1)
    chi2 = float((rr.astype(np.float32) ** 2).astype(np.float64).sum())
    g = np.maximum(np.float32(0), np.float32(1) - np.float32(50) * res * res)
    rimg = np.zeros((ch, cw, 4), np.float32)
    rimg[..., 3] = 1
    rimg[~has, 1] = 1
    rimg[has & ~inside, 0] = 1
    rimg[inside, 0] = rimg[inside, 1] = rimg[inside, 2] = g[inside]
    return H, bvec, chi2, int(inside.sum()), rimg


# ---- GuidedMatcher<StereoCamera>::match (matcher.cpp:98-181, 312-458; matcher-impl.cpp:33-51), written from the reference
#      source independently of the C oracle.  Tie-breaking uses the quadrant-key order the quadtree's DFS induces
#      (quadtree.h:515-540, 636-708; proven equivalent to the oracle's tree walk by test_quadtree_order_*).
def quad_key(px, py, W, H, depth=16):
    x0 = y0 = 0.0
    w, h = float(W), float(H)
    key = 0
    for _ in range(depth):
        bx = 0 if (1.0 - (x0 + w - px) / w) < 0.5 else 1
        by = 0 if (1.0 - (y0 + h - py) / h) < 0.5 else 1
        key = key * 4 + bx * 2 + by
        w *= 0.5; h *= 0.5
        x0 += w * bx; y0 += h * by
    return key


def _trunc_div(a, b):
    q = abs(a) // b
    return q if a >= 0 else -q


def match_point(pt, kf_pyr, T_anchor_from_w, T_cur_from_actkey, T_actkey_from_w, cur_pyr, disp, corners, cams,
                radius=8, thr_mean=22, thr_std=10):
    """One candidate point.  pt: CANDIDATE_DTYPE record; kf_pyr / cur_pyr: 3 u8 levels; corners[l]: (n,2) int array of the
    current frame's FAST corners; cams[l] = (f, cx, cy, w, h) of level l.  Returns (status, u, v, znssd, obs, xyz_actkey)
    with the oracle's status codes (0 ok, 2 border, 3 depth, 4 texture, 5 no candidate, 6 no disparity)."""
    l = int(pt["anchor_level"])
    f, cx, cy, W, H = cams[l]
    T_cur_from_w = pose_mul(T_cur_from_actkey, T_actkey_from_w)
    T_c2_from_c1 = pose_mul(T_cur_from_w, pose_inv(T_anchor_from_w))
    act = lambda T, p: T[:, :3] @ p + T[:, 3]
    xyz_a = np.asarray(pt["xyz_anchor"], np.float64)
    xyz_cur = act(T_c2_from_c1, xyz_a)
    uv_pyr = np.array([f * (xyz_cur[0] / xyz_cur[2]) + cx, f * (xyz_cur[1] / xyz_cur[2]) + cy])
    key_uv = np.asarray(pt["anchor_obs_pyr"][:2], np.float64)
    ku, kv = int(key_uv[0]), int(key_uv[1])
    none = (0, 0, thr_mean * thr_mean * 64, np.zeros(3), np.zeros(3))
    if not (4 <= ku < W - 4 and 4 <= kv < H - 4):
        return (2,) + none
    d_cur, d_anc = 1.0 / xyz_cur[2], 1.0 / xyz_a[2]
    if d_cur > d_anc * 3 or d_anc > d_cur * 3:
        return (3,) + none
    # warpAffinve
    def fwd(uv):
        p = np.array([(uv[0] - cx) / f, (uv[1] - cy) / f, 1.0]) * xyz_a[2]
        q = act(T_c2_from_c1, p)
        return np.array([f * (q[0] / q[2]) + cx, f * (q[1] / q[2]) + cy])
    f0, fu, fv = fwd(key_uv), fwd(key_uv + np.array([1.0, 0.0])), fwd(key_uv + np.array([0.0, 1.0]))
    A = np.array([fu - f0, fv - f0])
    det = A[0, 0] * A[1, 1] - A[1, 0] * A[0, 1]
    inv = np.array([[A[1, 1], -A[0, 1]], [-A[1, 0], A[0, 0]]]) * (1.0 / det)
    img = kf_pyr[l]
    patch = np.zeros((10, 10), np.uint8)
    for ix in range(10):
        for iy in range(10):
            r0 = (inv[0, 0] * (ix - 5) + inv[0, 1] * (iy - 5)) + key_uv[0]
            r1 = (inv[1, 0] * (ix - 5) + inv[1, 1] * (iy - 5)) + key_uv[1]
            x, y = np.floor(r0), np.floor(r1)
            if x < 0 or y < 0 or x + 1 >= W or y + 1 >= H or not np.isfinite(x + y):
                val = 0
            else:
                sx, sy = r0 - x, r1 - y
                xi, yi = int(x), int(y)
                s = ((1 - sx) * (1 - sy)) * float(img[yi, xi]) + ((1 - sx) * sy) * float(img[yi + 1, xi]) + \
                    (sx * (1 - sy)) * float(img[yi, xi + 1]) + (sx * sy) * float(img[yi + 1, xi + 1])
                val = int(min(255.0, s))
            patch[iy, ix] = val
    key = patch[1:9, 1:9].astype(np.int64)
    sumA, sumAA = int(key.sum()), int((key * key).sum())
    xyz_actkey = act(pose_inv(pose_mul(T_anchor_from_w, pose_inv(T_actkey_from_w))), xyz_a)
    if sumA * sumA - sumAA < thr_std * thr_std * 64:
        return (4, 0, 0, thr_mean * thr_mean * 64, np.zeros(3), xyz_actkey)
    ui, vi = int(uv_pyr[0]), int(uv_pyr[1])
    c = corners[l]
    inwin = c[(c[:, 0] >= ui - radius) & (c[:, 0] <= ui + radius) & (c[:, 1] >= vi - radius) & (c[:, 1] <= vi + radius)]
    order = sorted(range(len(inwin)), key=lambda i: quad_key(float(inwin[i, 0]), float(inwin[i, 1]), W, H))
    best, bu, bv = thr_mean * thr_mean * 64, -1, -1
    cur = cur_pyr[l].astype(np.int64)
    for i in order:
        u, v = int(inwin[i, 0]), int(inwin[i, 1])
        if not (6 <= u < W - 6 and 6 <= v < H - 6):
            continue
        Bp = cur[v - 4:v + 4, u - 4:u + 4]
        sumB, sumBB, sumAB = int(Bp.sum()), int((Bp * Bp).sum()), int((Bp * key).sum())
        z = sumAA - 2 * sumAB - sumBB - _trunc_div(sumA * sumA - 2 * sumA * sumB - sumB * sumB, 64)
        if z < best:
            best, bu, bv = z, u, v
    if bu < 0:
        return (5, 0, 0, thr_mean * thr_mean * 64, np.zeros(3), xyz_actkey)
    d = float(disp[bv << l, bu << l]) * (1.0 / (1 << l))
    if d > 0:
        s = float(1 << l)
        return (0, bu, bv, best, np.array([float(np.float32(bu)) * s, float(np.float32(bv)) * s, (float(np.float32(bu)) - d) * s]), xyz_actkey)
    return (6, bu, bv, best, np.zeros(3), xyz_actkey)


# ---- sensitivity of the landmark back-substitution (vectorised over edges, numeric Jacobians) ---------------------------
def _residual_vec(psi, T_obs, T_anc, obs, cam):
    """stereo_residual for arrays: psi [E,3], T_* [E,3,4], obs [E,3] -> [E,3]"""
    xa = np.stack([psi[:, 0], psi[:, 1], np.ones(len(psi))], 1) / psi[:, 2:3]
    Ra, ta = T_anc[:, :, :3], T_anc[:, :, 3]
    xw = np.einsum("eji,ej->ei", Ra, xa - ta)                 # T_anc^-1 x_a
    y = np.einsum("eij,ej->ei", T_obs[:, :, :3], xw) + T_obs[:, :, 3]
    f, cx, cy, b = cam
    return obs - np.stack([f * y[:, 0] / y[:, 2] + cx, f * y[:, 1] / y[:, 2] + cy, f * (y[:, 0] - b) / y[:, 2] + cx], 1)


def _exp_left(T, d):
    """exp(d) * T for arrays T [E,3,4] and ONE tangent vector d (6)"""
    E4 = np.vstack([se3_exp(d), [0, 0, 0, 1]])
    T4 = np.concatenate([T, np.tile(np.array([[[0, 0, 0, 1.0]]]), (len(T), 1, 1))], 1)
    return np.einsum("ij,ejk->eik", E4, T4)[:, :3]


def landmark_amplification(poses, psi, edges, cam, lam, delta=1.0, eps=1e-6):
    """Per landmark l: a_l = || (H_ll + lam I)^-1 ||_2 * || [W_1 .. W_m] ||_F, the factor by which a perturbation of the pose
    update x_p is amplified in x_l = D^-1 (b_l - sum W_i^T x_i)  (BlockSolver::solve).  Numeric Jacobians of
    the stereo residual, Huber weights at the given state, self edges (observer == anchor) have no pose Jacobian."""
    P = poses.reshape(-1, 3, 4)
    To, Ta = P[edges["pose"]], P[edges["anchor"]]
    ps = psi[edges["point"]]
    r0 = _residual_vec(ps, To, Ta, edges["obs"], cam)
    e2 = (r0 * r0 * edges["info"]).sum(1)
    rho1 = np.where(e2 <= delta * delta, 1.0, delta / np.sqrt(np.maximum(e2, 1e-300)))
    om = edges["info"] * rho1[:, None]
    Jpsi = np.zeros((len(edges), 3, 3))
    for k in range(3):
        d = np.zeros(3); d[k] = eps
        Jpsi[:, :, k] = (_residual_vec(ps + d, To, Ta, edges["obs"], cam) - _residual_vec(ps - d, To, Ta, edges["obs"], cam)) / (2 * eps)
    Jo = np.zeros((len(edges), 3, 6)); Ja = np.zeros((len(edges), 3, 6))
    for k in range(6):
        d = np.zeros(6); d[k] = eps
        Jo[:, :, k] = (_residual_vec(ps, _exp_left(To, d), Ta, edges["obs"], cam) - _residual_vec(ps, _exp_left(To, -d), Ta, edges["obs"], cam)) / (2 * eps)
        Ja[:, :, k] = (_residual_vec(ps, To, _exp_left(Ta, d), edges["obs"], cam) - _residual_vec(ps, To, _exp_left(Ta, -d), edges["obs"], cam)) / (2 * eps)
    self_edge = edges["pose"] == edges["anchor"]
    Jo[self_edge] = 0; Ja[self_edge] = 0
    L = len(psi)
    Hll = np.zeros((L, 3, 3)); W2 = np.zeros(L)
    np.add.at(Hll, edges["point"], np.einsum("eki,ek,ekj->eij", Jpsi, om, Jpsi))
    Wo = np.einsum("eki,ek,ekj->eij", Jo, om, Jpsi)
    Wa = np.einsum("eki,ek,ekj->eij", Ja, om, Jpsi)
    np.add.at(W2, edges["point"], (Wo ** 2).sum((1, 2)) + (Wa ** 2).sum((1, 2)))     # upper bound: anchor blocks of a landmark add up coherently at most
    seen = np.zeros(L, bool); seen[edges["point"]] = True
    amp = np.zeros(L)
    ev = np.linalg.eigvalsh(Hll[seen] + lam * np.eye(3))
    amp[seen] = np.sqrt(W2[seen]) / ev[:, 0]
    return amp
