"""Deterministic synthetic inputs for tests and bench.py.

No dataset is available offline (New College images are a separate download, README:40-55), so
frames are ray-cast from a procedural textured scene and BA windows are generated from a seeded
model.  Pure numpy; nothing here touches the GPU or the oracle.
"""
import numpy as np

from .ctypes_types import BA_CONSTRAINT_DTYPE, BA_EDGE_DTYPE, CANDIDATE_DTYPE

# stereo_slam.cpp:655-660 defaults (640x480) and data/newcollege.cfg:1-6 (512x384)
CAM_DEFAULT = dict(f=570.342, cx=320.0, cy=240.0, b=0.075, w=640, h=480)
CAM_NEWCOLLEGE = dict(f=389.956, cx=254.903, cy=201.899, b=0.120005, w=512, h=384)
CAM_RGBD = dict(f=591.524, cx=319.5, cy=239.5, b=0.07468, w=640, h=480)


# ---------------------------------------------------------------------------------------------
def so3_exp(w):
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-10:
        return np.eye(3) + W
    return np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * (W @ W)


def pose(R, t):
    T = np.zeros((3, 4))
    T[:, :3] = R
    T[:, 3] = t
    return T


def pose_mul(A, B):
    return pose(A[:, :3] @ B[:, :3], A[:, :3] @ B[:, 3] + A[:, 3])


def pose_inv(A):
    return pose(A[:, :3].T, -A[:, :3].T @ A[:, 3])


def band_noise(rng, size, octaves=4):
    """Band-limited noise texture in [0,255]: sum of bilinearly upsampled uniform noise."""
    acc = np.zeros((size, size))
    for o in range(octaves):
        n = 2 ** (o + 3)
        g = rng.random((n + 1, n + 1))
        xs = np.linspace(0, n, size, endpoint=False)
        x0 = np.floor(xs).astype(int)
        fx = xs - x0
        rows = g[x0][:, None, :] * (1 - fx)[:, None, None] + g[x0 + 1][:, None, :] * fx[:, None, None]
        rows = rows[:, 0, :]
        up = rows[:, x0] * (1 - fx)[None, :] + rows[:, x0 + 1] * fx[None, :]
        acc += up / (o + 1)
    acc -= acc.min()
    acc /= acc.max()
    return acc * 255.0


class Scene:
    """Ground plane + back wall + two side walls, each textured with band-limited noise."""

    def __init__(self, seed=2011, tex_size=1024):
        rng = np.random.default_rng(seed)
        self.tex = []
        for _ in range(4):
            t = band_noise(rng, tex_size, 6)
            # piecewise-constant patches give FAST real corners (a smooth texture has few)
            for _ in range(2500):
                x0, y0 = rng.integers(0, tex_size - 24, 2)
                sx, sy = rng.integers(4, 24, 2)
                t[y0:y0 + sy, x0:x0 + sx] = 0.5 * t[y0:y0 + sy, x0:x0 + sx] + rng.uniform(0, 70)
            self.tex.append(np.clip(t, 0, 255))
        # planes n.x = d in world coords (x right, y down, z forward)
        self.planes = [(np.array([0.0, 1.0, 0.0]), 1.5), (np.array([0.0, 0.0, 1.0]), 15.0),
                       (np.array([1.0, 0.0, 0.0]), 4.0), (np.array([-1.0, 0.0, 0.0]), 4.0)]
        self.axes = [(0, 2), (0, 1), (2, 1), (2, 1)]
        self.tex_scale = 40.0  # texels per metre
        self.rng = rng

    def render(self, cam, T_cam_from_world, noise_sigma=2.0, seed=0):
        """Returns (u8 image, f32 disparity map) for a pin-hole camera (dict f,cx,cy,b,w,h)."""
        w, h = cam["w"], cam["h"]
        u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        d_cam = np.stack([(u - cam["cx"]) / cam["f"], (v - cam["cy"]) / cam["f"], np.ones_like(u)], -1)
        Twc = pose_inv(T_cam_from_world)
        R, c = Twc[:, :3], Twc[:, 3]
        d_w = d_cam @ R.T
        best_t = np.full((h, w), np.inf)
        img = np.zeros((h, w))
        for k, ((n, dd), ax) in enumerate(zip(self.planes, self.axes)):
            denom = d_w @ n
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (dd - c @ n) / denom
            ok = (t > 1e-3) & (t < best_t) & np.isfinite(t)
            p = c[None, None, :] + d_w * np.where(ok, t, 0.0)[..., None]
            ts = self.tex[k].shape[0]
            a = (p[..., ax[0]] * self.tex_scale) % (ts - 1)
            b = (p[..., ax[1]] * self.tex_scale) % (ts - 1)
            a0 = np.minimum(np.floor(a).astype(int), ts - 2)      # a % (ts-1) can round up to ts-1 itself
            b0 = np.minimum(np.floor(b).astype(int), ts - 2)
            fa, fb = a - a0, b - b0
            tx = self.tex[k]
            val = (tx[b0, a0] * (1 - fa) * (1 - fb) + tx[b0, a0 + 1] * fa * (1 - fb) +
                   tx[b0 + 1, a0] * (1 - fa) * fb + tx[b0 + 1, a0 + 1] * fa * fb)
            img = np.where(ok, val, img)
            best_t = np.where(ok, t, best_t)
        z = best_t  # d_cam z-component is 1 => depth along optical axis = t
        disp = np.where(np.isfinite(z), cam["f"] * cam["b"] / z, 0.0).astype(np.float32)
        nrng = np.random.default_rng(seed + 77)
        img = img + nrng.normal(0, noise_sigma, img.shape)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8), disp


def render_stereo(scene, cam, T_cam_from_world, noise_sigma=2.0, seed=0):
    """(left u8, right u8, true left disparity f32): the right camera sits `b` along the left camera's x axis."""
    left, disp = scene.render(cam, T_cam_from_world, noise_sigma, seed)
    T_right_from_left = pose(np.eye(3), np.array([-cam["b"], 0.0, 0.0]))
    right, _ = scene.render(cam, pose_mul(T_right_from_left, T_cam_from_world), noise_sigma, seed + 1000)
    return left, right, disp


def trajectory(n, step=0.05, yaw_deg=0.2):
    """T_cam_from_world for n frames: forward motion + yaw."""
    out = []
    for i in range(n):
        R_wc = so3_exp(np.array([0.0, np.deg2rad(yaw_deg) * i, 0.0]))
        c = np.array([0.02 * np.sin(0.1 * i), 0.0, step * i])
        out.append(pose_inv(pose(R_wc, c)))
    return out


def noise_image(w, h, seed, blobs=True):
    """Plain random test image with corner-rich structure (for FAST/pyramid bit-exact tests)."""
    rng = np.random.default_rng(seed)
    base = band_noise(rng, 256, 5)
    ys = (np.arange(h) * 255 // max(h - 1, 1))
    xs = (np.arange(w) * 255 // max(w - 1, 1))
    img = base[np.ix_(ys, xs)]
    if blobs:
        for _ in range(max(8, w * h // 1500)):
            x0, y0 = rng.integers(0, w), rng.integers(0, h)
            s = rng.integers(2, 9)
            img[y0:y0 + s, x0:x0 + s] = rng.integers(0, 256)
    img = img + rng.normal(0, 3, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def candidate_points(rng, cam, disp_anchor, T_anchor_from_w, n_per_level=(1200, 600, 200), kf_index=0):
    """Candidate points anchored in a keyframe with known disparity."""
    pts = np.zeros(sum(n_per_level), CANDIDATE_DTYPE)
    k = 0
    h, w = disp_anchor.shape
    for lvl, n in enumerate(n_per_level):
        s = 1 << lvl
        for _ in range(n):
            while True:
                u0 = int(rng.integers(8 * s, w - 8 * s))
                v0 = int(rng.integers(8 * s, h - 8 * s))
                u0 -= u0 % s
                v0 -= v0 % s
                d = float(disp_anchor[v0, u0])
                if d > 0.5:
                    break
            z = cam["f"] * cam["b"] / d
            xyz = np.array([(u0 - cam["cx"]) / cam["f"] * z, (v0 - cam["cy"]) / cam["f"] * z, z])
            pts[k]["xyz_anchor"] = xyz
            pts[k]["anchor_obs_pyr"] = np.array([u0 / s, v0 / s, (u0 - d) / s])
            pts[k]["anchor_level"] = lvl
            pts[k]["kf_index"] = kf_index
            pts[k]["point_id"] = k
            k += 1
    return pts


# ---------------------------------------------------------------------------------------------
def ba_window(P=15, L=3000, seed=2012, cam=CAM_NEWCOLLEGE, n_outer=3, outlier_frac=0.02,
              pose_sigma_t=0.01, pose_sigma_r_deg=0.5):
    """Double-window BA problem.

    Returns dict(poses [P,12] (perturbed T_me_from_world), psi [L,3] (perturbed inverse-depth in
    the anchor frame), edges (sorted by landmark, then pose), cons, cam dict, poses_gt, psi_gt).
    (Each relative-pose constraint is listed once; the reference's copyContraintsToG2o inserts it in both directions.)
    """
    rng = np.random.default_rng(seed)
    f, cx, cy, b = cam["f"], cam["cx"], cam["cy"], cam["b"]
    gt = []
    for i in range(P):
        ang = 0.01 * i
        R_wc = so3_exp(np.array([0.0, ang, 0.0]))
        c = np.array([2.0 * np.sin(0.05 * i), 0.0, 0.4 * i])
        gt.append(pose_inv(pose(R_wc, c)))
    poses = []
    for T in gt:
        dx = np.concatenate([rng.normal(0, pose_sigma_t, 3), rng.normal(0, np.deg2rad(pose_sigma_r_deg), 3)])
        E = pose(so3_exp(dx[3:]), dx[:3])
        poses.append(pose_mul(E, T))
    inner = P - n_outer if P > n_outer + 2 else P
    edges = []
    psi = np.zeros((L, 3))
    psi_gt = np.zeros((L, 3))
    for l in range(L):
        k = 2 + rng.binomial(6, 0.5)
        k = min(k, inner)
        a = int(rng.integers(0, inner - k + 1))
        # true point in the anchor frame, depth U(2,20), inside the anchor's image
        z = rng.uniform(2.0, 20.0)
        u0 = rng.uniform(40, cam["w"] - 40)
        v0 = rng.uniform(40, cam["h"] - 40)
        xa = np.array([(u0 - cx) / f * z, (v0 - cy) / f * z, z])
        psi_gt[l] = [xa[0] / xa[2], xa[1] / xa[2], 1.0 / xa[2]]
        psi[l] = psi_gt[l]
        psi[l, 2] *= 1.0 + rng.normal(0, 0.02)
        T_w_from_a = pose_inv(gt[a])
        xw = T_w_from_a[:, :3] @ xa + T_w_from_a[:, 3]
        for i in range(a, a + k):
            y = gt[i][:, :3] @ xw + gt[i][:, 3]
            if y[2] < 0.5:
                continue
            obs = np.array([f * y[0] / y[2] + cx, f * y[1] / y[2] + cy, f * (y[0] - b) / y[2] + cx])
            obs += rng.normal(0, 0.5, 3)
            if rng.random() < outlier_frac:
                obs += rng.uniform(-20, 20, 3)
            lvl = int(rng.choice(3, p=[0.6, 0.3, 0.1]))
            s = 0.25 ** lvl  # Po2(pyrFromZero_d(1,level)), slam_graph.cpp:1010-1015
            edges.append((obs, (s, s, 0.333 ** 2), l, i, a))
    e = np.zeros(len(edges), BA_EDGE_DTYPE)
    for j, (obs, info, l, i, a) in enumerate(edges):
        e[j]["obs"], e[j]["info"], e[j]["point"], e[j]["pose"], e[j]["anchor"] = obs, info, l, i, a
    order = np.lexsort((e["pose"], e["point"]))
    e = e[order]
    # outer-window poses tied to the window with relative-pose constraints (slam_graph.cpp:937-981)
    cons = []
    if inner < P:
        for j in range(inner, P):
            i = j - 1
            T_ji = pose_mul(gt[j], pose_inv(gt[i]))
            dn = np.concatenate([rng.normal(0, 0.005, 3), rng.normal(0, 0.002, 3)])
            T_ji = pose_mul(pose(so3_exp(dn[3:]), dn[:3]), T_ji)
            vis = 30.0
            norm_dist = np.linalg.norm(T_ji[:, 3]) / 8.0
            Lam = np.eye(6) * vis                     # slam_graph.cpp:842-845
            Lam[:3, :3] *= (350 * norm_dist) ** 2
            Lam[3:, 3:] *= 100.0 ** 2
            cons.append((T_ji, Lam, i, j))
    c = np.zeros(len(cons), BA_CONSTRAINT_DTYPE)
    for j, (T, Lam, i1, i2) in enumerate(cons):
        c[j]["T_21"], c[j]["info"], c[j]["pose1"], c[j]["pose2"] = T.reshape(12), Lam.reshape(36), i1, i2
    return dict(poses=np.array([T.reshape(12) for T in poses]), psi=psi, edges=e, cons=c, cam=cam,
                poses_gt=np.array([T.reshape(12) for T in gt]), psi_gt=psi_gt)


def double_window(n_inner=30, n_outer=200, L=12000, seed=2014, cam=CAM_NEWCOLLEGE, n_long=(), n_loops=2, outlier_frac=0.02):
    """The reference's double window as SlamGraph::copyDataToG2o hands it over (slam_graph.cpp:907-1032): poses 0 .. n_outer-1 are the
    OUTER window (older keyframes), the last n_inner poses the INNER one; ALL are free (slam_graph.cpp:932).  Active points are
    the points seen from the inner window; each is observed from every window pose in its vis_set -- a contiguous run of
    keyframes that may reach back into the outer window -- with no cap on the run length (n_long: extra landmarks with that many
    observations).  Every pair of neighbouring keyframes of which at least one is OUTER carries a relative-pose constraint
    (i, i+1) and (i, i+2) (the co-visibility edges of the pose graph), plus n_loops loop-closure constraints across the window
    (slam_graph.cpp:937-981).  Same noise model as ba_window."""
    rng = np.random.default_rng(seed)
    P = n_inner + n_outer
    f, cx, cy, b = cam["f"], cam["cx"], cam["cy"], cam["b"]
    gt = []
    for i in range(P):            # a wide loop: after ~P keyframes the camera is back near its start and looks the same way
        ang = 2 * np.pi * i / (P + 20)
        R_wc = so3_exp(np.array([0.0, ang, 0.0]))
        rad = 0.4 * (P + 20) / (2 * np.pi)
        c = np.array([rad * (1 - np.cos(ang)), 0.0, rad * np.sin(ang)])
        gt.append(pose_inv(pose(R_wc, c)))
    poses = [pose_mul(pose(so3_exp(rng.normal(0, np.deg2rad(0.5), 3)), rng.normal(0, 0.01, 3)), T) for T in gt]
    edges = []
    psi = np.zeros((L, 3)); psi_gt = np.zeros((L, 3))

    def add_landmark(l, first, k, anchor):
        z = rng.uniform(3.0, 25.0)
        u0, v0 = rng.uniform(60, cam["w"] - 60), rng.uniform(60, cam["h"] - 60)
        xa = np.array([(u0 - cx) / f * z, (v0 - cy) / f * z, z])
        psi_gt[l] = [xa[0] / xa[2], xa[1] / xa[2], 1.0 / xa[2]]
        psi[l] = psi_gt[l]; psi[l, 2] *= 1.0 + rng.normal(0, 0.02)
        Twa = pose_inv(gt[anchor])
        xw = Twa[:, :3] @ xa + Twa[:, 3]
        for i in range(first, first + k):
            y = gt[i][:, :3] @ xw + gt[i][:, 3]
            if i != anchor and (k <= 64 and y[2] < 0.5):
                continue                                   # behind the camera: not observed (long-lived points keep every view)
            obs = np.array([f * y[0] / y[2] + cx, f * y[1] / y[2] + cy, f * (y[0] - b) / y[2] + cx]) + rng.normal(0, 0.5, 3)
            if rng.random() < outlier_frac:
                obs += rng.uniform(-20, 20, 3)
            s_ = 0.25 ** int(rng.choice(3, p=[0.6, 0.3, 0.1]))
            edges.append((obs, (s_, s_, 0.333 ** 2), l, i, anchor))

    l = 0
    for k in n_long:                                       # long-lived points: runs ending in the inner window, anchored at their first view
        first = P - k - int(rng.integers(0, 5))
        add_landmark(l, first, k, first)
        l += 1
    while l < L:
        k = 2 + rng.binomial(6, 0.5)
        last = int(rng.integers(n_outer, P))               # seen from at least one inner keyframe
        first = max(0, last - k + 1)
        if rng.random() < 0.2:                             # some reach further back into the outer window
            first = max(0, first - int(rng.integers(1, 12)))
        add_landmark(l, first, last - first + 1, first)
        l += 1
    e = np.zeros(len(edges), BA_EDGE_DTYPE)
    for j, (obs, info, ll, i, a) in enumerate(edges):
        e[j]["obs"], e[j]["info"], e[j]["point"], e[j]["pose"], e[j]["anchor"] = obs, info, ll, i, a
    e = e[np.lexsort((e["pose"], e["point"]))]
    cons = []

    def add_constraint(i, j):
        T_ji = pose_mul(gt[j], pose_inv(gt[i]))
        dn = np.concatenate([rng.normal(0, 0.005, 3), rng.normal(0, 0.002, 3)])
        T_ji = pose_mul(pose(so3_exp(dn[3:]), dn[:3]), T_ji)
        Lam = np.eye(6) * 30.0                              # slam_graph.cpp:842-845
        Lam[:3, :3] *= (350 * np.linalg.norm(T_ji[:, 3]) / 8.0) ** 2
        Lam[3:, 3:] *= 100.0 ** 2
        cons.append((T_ji, Lam, i, j))

    for i in range(P - 1):
        for j in (i + 1, i + 2):
            if j < P and (i < n_outer or j < n_outer):      # an edge touching the OUTER window is a marginalised constraint
                add_constraint(i, j)
    for k in range(n_loops):
        add_constraint(3 + 7 * k, P - 4 - 9 * k)
    c = np.zeros(len(cons), BA_CONSTRAINT_DTYPE)
    for j, (T, Lam, i1, i2) in enumerate(cons):
        c[j]["T_21"], c[j]["info"], c[j]["pose1"], c[j]["pose2"] = T.reshape(12), Lam.reshape(36), i1, i2
    return dict(poses=np.array([T.reshape(12) for T in poses]), psi=psi, edges=e, cons=c, cam=cam,
                poses_gt=np.array([T.reshape(12) for T in gt]), psi_gt=psi_gt, n_inner=n_inner, n_outer=n_outer)


def shard_landmarks(edges, L, n_shards, chunk=64):
    """Landmarks dealt to shards in contiguous chunks of `chunk`, round-robin.

    Returns list of (landmark_ids, edge_mask) per shard; edges keep their global landmark ids.
    """
    owner = (np.arange(L) // chunk) % n_shards
    out = []
    for s in range(n_shards):
        ids = np.nonzero(owner == s)[0]
        mask = owner[edges["point"]] == s
        out.append((ids, mask))
    return out


# ---- full-resolution dense tracking (BASELINE config 5: RGB-D 640x480 depth frames) -----------------------------
def level_cams(cam, levels=3):
    """cam_vec of frame_grabber-impl.cpp:48-60: f / 2^l, pp / 2^l, size / 2^l, baseline * 2^l."""
    return [dict(f=cam["f"] / (1 << l), cx=cam["cx"] / (1 << l), cy=cam["cy"] / (1 << l), b=cam["b"] * (1 << l),
                 w=cam["w"] >> l, h=cam["h"] >> l) for l in range(levels)]


def cloud_full_level(disp0, cam, level):
    """float4 cloud of one pyramid level at full level resolution from the level-0 disparity image: metrically correct
    points (x, y, z, 1) in the frame's own coordinates, (0, 0, 0, -1) where there is no depth.  (The reference's
    pointcloud_kernel has two indexing quirks on levels > 0; trackers take the cloud as an input.)"""
    c = level_cams(cam)[level]
    d = disp0[::1 << level, ::1 << level][:c["h"], :c["w"]].astype(np.float64)
    u, v = np.meshgrid(np.arange(c["w"], dtype=np.float64), np.arange(c["h"], dtype=np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        z = cam["f"] * cam["b"] / d
    ok = d > 0
    out = np.zeros((c["h"], c["w"], 4), np.float32)
    out[..., 0] = np.where(ok, (u - c["cx"]) * z / c["f"], 0)
    out[..., 1] = np.where(ok, (v - c["cy"]) * z / c["f"], 0)
    out[..., 2] = np.where(ok, z, 0)
    out[..., 3] = np.where(ok, 1.0, -1.0)
    return out


def depth_holes(disp, rng, frac=0.10):
    """invalid-depth blobs as an RGB-D sensor leaves them (10 % invalid pixels in blobs)"""
    h, w = disp.shape
    out = disp.copy()
    area = 0
    while area < frac * h * w:
        x0, y0 = rng.integers(0, w), rng.integers(0, h)
        sx, sy = rng.integers(4, max(5, w // 10)), rng.integers(4, max(5, h // 10))
        out[y0:y0 + sy, x0:x0 + sx] = 0
        area += sx * sy
    return out


def dense_full_case(cam=None, seed=2013, step=0.03, yaw_deg=0.3, holes=True, scene=None, frame=3):
    """Two consecutive frames for the full-resolution tracker: previous image + its disparity (with holes), current image,
    and the true T_cur_from_prev.  Returns dict(img_prev, img_cur, disp_prev, T_true, cam)."""
    cam = dict(CAM_RGBD if cam is None else cam)
    sc = scene or Scene(seed)
    traj = trajectory(frame + 2, step=step, yaw_deg=yaw_deg)
    T_p, T_c = traj[frame], traj[frame + 1]
    img_p, disp_p = sc.render(cam, T_p, seed=seed)
    img_c, _ = sc.render(cam, T_c, seed=seed + 1)
    if holes:
        disp_p = depth_holes(disp_p, np.random.default_rng(seed + 5))
    return dict(img_prev=img_p, img_cur=img_c, disp_prev=disp_p, T_true=pose_mul(T_c, pose_inv(T_p)), cam=cam)


def trajectory_there_and_back(n=200, turn=110, step=0.05, yaw_deg=0.2):
    """BASELINE configs[0] ("first 200 frames"): forward along `trajectory` for `turn` frames, then back over the same poses -- the way back passes the keyframes the
    way out dropped, so a front end run over it drops keyframes AND switches back to old ones (stereo_frontend.cpp:445-510)."""
    fwd = trajectory(max(turn, n - turn) + 1, step, yaw_deg)
    return [fwd[i if i < turn else 2 * turn - i] for i in range(n)]
