"""Synthetic graphs / frames of the shapes BASELINE.json names (SURVEY.md §8 d).

Pure numpy, deterministic per seed; used by bench.py and the parity tests.  Nothing here touches
the oracle: observations are produced with a vectorised closed-form dual-conic projection.
"""
import numpy as np

from .abi import Graph

TUM3_K = (535.4, 539.2, 320.1, 247.6)  # Example/param/TUM3.yaml:62-65
IMG_W, IMG_H = 640, 480


def _quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def _quat_to_R(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _R_to_quat(R):
    """batched, w >= 0"""
    R = np.asarray(R)
    q = np.empty(R.shape[:-2] + (4,))
    t = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    # robust: use the largest of (w,x,y,z)
    ww = 1 + t
    xx = 1 + R[..., 0, 0] - R[..., 1, 1] - R[..., 2, 2]
    yy = 1 - R[..., 0, 0] + R[..., 1, 1] - R[..., 2, 2]
    zz = 1 - R[..., 0, 0] - R[..., 1, 1] + R[..., 2, 2]
    c = np.argmax(np.stack([ww, xx, yy, zz], -1), -1)
    for k in range(4):
        m = c == k
        if not np.any(m):
            continue
        Rm = R[m]
        if k == 0:
            s = 2 * np.sqrt(ww[m]); qm = np.stack([(Rm[:, 2, 1] - Rm[:, 1, 2]) / s, (Rm[:, 0, 2] - Rm[:, 2, 0]) / s, (Rm[:, 1, 0] - Rm[:, 0, 1]) / s, s / 4], -1)
        elif k == 1:
            s = 2 * np.sqrt(xx[m]); qm = np.stack([s / 4, (Rm[:, 0, 1] + Rm[:, 1, 0]) / s, (Rm[:, 0, 2] + Rm[:, 2, 0]) / s, (Rm[:, 2, 1] - Rm[:, 1, 2]) / s], -1)
        elif k == 2:
            s = 2 * np.sqrt(yy[m]); qm = np.stack([(Rm[:, 0, 1] + Rm[:, 1, 0]) / s, s / 4, (Rm[:, 1, 2] + Rm[:, 2, 1]) / s, (Rm[:, 0, 2] - Rm[:, 2, 0]) / s], -1)
        else:
            s = 2 * np.sqrt(zz[m]); qm = np.stack([(Rm[:, 0, 2] + Rm[:, 2, 0]) / s, (Rm[:, 1, 2] + Rm[:, 2, 1]) / s, s / 4, (Rm[:, 1, 0] - Rm[:, 0, 1]) / s], -1)
        q[m] = qm
    q = np.where(q[..., 3:4] < 0, -q, q)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def _rotvec_to_quat(v):
    v = np.asarray(v, dtype=float)
    th = np.linalg.norm(v, axis=-1, keepdims=True)
    half = 0.5 * th
    k = np.where(th > 1e-12, np.sin(half) / np.where(th > 1e-12, th, 1), 0.5)
    return np.concatenate([v * k, np.cos(half)], -1)


def _euler_zyx_to_quat(roll, pitch, yaw):
    sy, cy = np.sin(yaw * .5), np.cos(yaw * .5)
    sp, cp = np.sin(pitch * .5), np.cos(pitch * .5)
    sr, cr = np.sin(roll * .5), np.cos(roll * .5)
    return np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy], -1)


def project_bboxes(cams7, objs10, K, cam_idx, obj_idx):
    """bbox (x1,y1,x2,y2) of ellipsoid obj_idx[k] seen from camera cam_idx[k] (Tcw), plus depth of the centre."""
    fx, fy, cx, cy = K
    Rc = _quat_to_R(cams7[cam_idx, 3:7]); tc = cams7[cam_idx, 0:3]
    Ro = _quat_to_R(objs10[obj_idx, 3:7]); to = objs10[obj_idx, 0:3]
    s = objs10[obj_idx, 7:10]
    # M = [Rc|tc] [Ro to; 0 1] -> 3x4
    Rco = Rc @ Ro
    tco = (Rc @ to[..., None])[..., 0] + tc
    Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    A = Km @ Rco                     # (n,3,3)
    t = (Km @ tco[..., None])[..., 0]  # (n,3)
    D = s * s
    C = np.einsum('nik,nk,njk->nij', A, D, A) - t[:, :, None] * t[:, None, :]
    with np.errstate(invalid='ignore'):
        du = np.sqrt(C[:, 0, 2] ** 2 - C[:, 0, 0] * C[:, 2, 2])
        dv = np.sqrt(C[:, 1, 2] ** 2 - C[:, 1, 1] * C[:, 2, 2])
    u1 = (C[:, 0, 2] - du) / C[:, 2, 2]; u2 = (C[:, 0, 2] + du) / C[:, 2, 2]
    v1 = (C[:, 1, 2] - dv) / C[:, 2, 2]; v2 = (C[:, 1, 2] + dv) / C[:, 2, 2]
    bb = np.stack([np.minimum(u1, u2), np.minimum(v1, v2), np.maximum(u1, u2), np.maximum(v1, v2)], -1)
    return bb, tco[:, 2], C[:, 2, 2]


def make_graph(n_cams=500, n_objs=50, n_bbox_target=5000, seed=0, slam=False, frac_3d=0.2,
               gravity=True, room=4.0, obs_per_obj=None):
    """Room 10x10x3 m, ground z=0, ellipsoids on the ground, cameras on a closed loop looking inward.
    Returns (Graph, cams_init (F,7), objs_init (N,10), truth dict)."""
    rng = np.random.default_rng(seed)
    K = TUM3_K
    N, F = n_objs, n_cams
    # --- ellipsoids (truth)
    s = rng.uniform(0.15, 0.6, size=(N, 3))
    ctr = np.concatenate([rng.uniform(-room, room, size=(N, 2)), s[:, 2:3]], 1)
    yaw = rng.uniform(-np.pi, np.pi, size=N)
    roll = 1e-2 * rng.standard_normal(N); pitch = 1e-2 * rng.standard_normal(N)
    qo = _euler_zyx_to_quat(roll, pitch, yaw)
    qo = np.where(qo[:, 3:4] < 0, -qo, qo)
    objs_true = np.concatenate([ctr, qo, s], 1)
    # --- cameras (truth): closed loop radius 4.5 m + slow radial wobble, height 1.2 m, look inward
    ang = 2 * np.pi * np.arange(F) / F
    rad = (room + 0.5) + 1.0 * np.sin(3 * ang)
    pos = np.stack([rad * np.cos(ang), rad * np.sin(ang), np.full(F, 1.2) + 0.1 * np.sin(5 * ang)], 1)
    tgt = np.stack([1.5 * np.cos(2 * ang + 1.0), 1.5 * np.sin(3 * ang), np.full(F, 0.4)], 1)
    fwd = tgt - pos
    jit = np.deg2rad(10) * (rng.uniform(-1, 1, size=(F, 3)))
    fwd /= np.linalg.norm(fwd, axis=1, keepdims=True)
    up = np.array([0, 0, 1.0])
    right = np.cross(fwd, up); right /= np.linalg.norm(right, axis=1, keepdims=True)
    down = np.cross(fwd, right)
    Rwc = np.stack([right, down, fwd], -1)  # columns = camera x,y,z axes in world
    Rj = _quat_to_R(_rotvec_to_quat(jit))
    Rwc = Rwc @ Rj
    Rcw = np.swapaxes(Rwc, -1, -2)
    tcw = -(Rcw @ pos[..., None])[..., 0]
    cams_true = np.concatenate([tcw, _R_to_quat(Rcw)], 1)
    # --- visibility: bbox fully inside image minus 10 px border, object in front (chunked over cameras)
    border = 10
    ci_l, oi_l, bb_l = [], [], []
    chunk = max(1, 2_000_000 // max(N, 1))
    for c0 in range(0, F, chunk):
        cc, oo = np.meshgrid(np.arange(c0, min(F, c0 + chunk)), np.arange(N), indexing='ij')
        cc, oo = cc.ravel(), oo.ravel()
        bb, z, c22 = project_bboxes(cams_true, objs_true, K, cc, oo)
        with np.errstate(invalid='ignore'):
            vis = (np.isfinite(bb).all(1) & (z > 0.3) & (c22 < 0) & (bb[:, 0] > border) & (bb[:, 1] > border)
                   & (bb[:, 2] < IMG_W - border) & (bb[:, 3] < IMG_H - border) & (bb[:, 2] - bb[:, 0] > 8)
                   & (bb[:, 3] - bb[:, 1] > 8))
        ci_l.append(cc[vis]); oi_l.append(oo[vis]); bb_l.append(bb[vis])
    ci, oi, bb = np.concatenate(ci_l), np.concatenate(oi_l), np.concatenate(bb_l)
    if len(ci) > n_bbox_target:
        sel = np.sort(rng.choice(len(ci), size=n_bbox_target, replace=False))
        ci, oi, bb = ci[sel], oi[sel], bb[sel]
    # group by object, observation (frame) order inside — the order Optimizer.cpp:166-245 adds them
    order = np.lexsort((ci, oi))
    ci, oi, bb = ci[order], oi[order], bb[order]
    # the reference only adds 2-D edges for objects with > 2 observations (Optimizer.cpp:201)
    cnt = np.bincount(oi, minlength=N)
    keep = cnt[oi] > 2
    ci, oi, bb = ci[keep], oi[keep], bb[keep]
    meas = bb + 2.0 * rng.standard_normal(bb.shape)
    weight = rng.uniform(0.3, 0.95, size=len(ci))
    # --- 3-D edges for a fraction of the observations (frame order)
    n3 = int(round(frac_3d * len(ci)))
    sel3 = np.sort(rng.choice(len(ci), size=n3, replace=False)) if n3 else np.zeros(0, int)
    o3 = np.lexsort((oi[sel3], ci[sel3]))
    c3, ob3 = ci[sel3][o3], oi[sel3][o3]
    Rc = _quat_to_R(cams_true[c3, 3:7]); tc = cams_true[c3, 0:3]
    Ro = _quat_to_R(objs_true[ob3, 3:7]); to = objs_true[ob3, 0:3]
    Rl = Rc @ Ro
    tl = (Rc @ to[..., None])[..., 0] + tc
    dq = _rotvec_to_quat(0.03 * rng.standard_normal((n3, 3)))
    ql = _quat_mul(_R_to_quat(Rl), dq) if n3 else np.zeros((0, 4))
    ql = np.where(ql[:, 3:4] < 0, -ql, ql)
    tl = tl + 0.03 * rng.standard_normal((n3, 3))
    sl = objs_true[ob3, 7:10] + 0.02 * rng.standard_normal((n3, 3))
    e3d_meas = np.concatenate([tl, ql, sl], 1)
    e3d_w = 10000.0 * rng.uniform(0.5, 1.0, size=n3)  # Optimizer.Edges.3DEllipsoid.Scale * prob
    # --- initial ellipsoids: truth perturbed 0.1 m / 5 deg / 10 %
    dq0 = _rotvec_to_quat(np.deg2rad(5) * rng.standard_normal((N, 3)) * np.array([0.2, 0.2, 1.0]))
    q0 = _quat_mul(qo, dq0); q0 = np.where(q0[:, 3:4] < 0, -q0, q0)
    objs_init = np.concatenate([ctr + 0.1 * rng.standard_normal((N, 3)) * np.array([1, 1, 0.3]), q0,
                                s * (1 + 0.1 * rng.uniform(-1, 1, size=(N, 3)))], 1)
    cams_init = cams_true.copy()
    cam_fixed = None
    odom = dict(odom_i=(), odom_j=(), odom_meas=())
    if slam:
        cam_fixed = np.zeros(F, np.uint8); cam_fixed[0] = 1
        # odometry measurement = Tcw_j * Tcw_i^-1 of the INPUT poses (Optimizer.cpp:143-150) -> use truth + noise-free
        i = np.arange(F - 1); j = i + 1
        Ri = _quat_to_R(cams_true[i, 3:7]); ti = cams_true[i, :3]
        Rj_ = _quat_to_R(cams_true[j, 3:7]); tj = cams_true[j, :3]
        Rz = Rj_ @ np.swapaxes(Ri, -1, -2)
        tz = tj - (Rz @ ti[..., None])[..., 0]
        odom = dict(odom_i=i, odom_j=j, odom_meas=np.concatenate([tz, _R_to_quat(Rz)], 1))
        # perturb cameras (except 0) by 2 cm / 0.5 deg
        dqc = _rotvec_to_quat(np.deg2rad(0.5) * rng.standard_normal((F, 3)))
        qc = _quat_mul(dqc, cams_true[:, 3:7]); qc = np.where(qc[:, 3:4] < 0, -qc, qc)
        Rd = _quat_to_R(dqc)
        tcn = (Rd @ cams_true[:, :3, None])[..., 0] + 0.02 * rng.standard_normal((F, 3))
        cams_init = np.concatenate([tcn, qc], 1)
        cams_init[0] = cams_true[0]
    g = Graph(K, F, N, cam_fixed, ci, oi, meas, weight, c3, ob3, e3d_meas, e3d_w,
              np.arange(N) if gravity else (), (0, 0, 1, 0), 100.0 ** 2, **odom)
    return g, cams_init, objs_init, dict(cams=cams_true, objs=objs_true)


CONFIGS = {
    # name: (n_cams, n_objs, n_bbox_target)  — BASELINE.json configs[2], configs[3], and a tiny test case
    "tiny": (20, 5, 60),
    "C3": (500, 50, 5000),
    "C4": (10000, 2000, 200000),
}


def make_config(name, seed=0, slam=False, **kw):
    F, N, E = CONFIGS[name]
    return make_graph(F, N, E, seed=seed, slam=slam, **kw)


# ---------------------------------------------------------------------------------------------------
# synthetic RGB-D frames for the single-frame fit (BASELINE.json configs[1] and configs[4] shapes)
# ---------------------------------------------------------------------------------------------------
def make_depth_scene(width=640, height=480, n_objs=3, seed=0, scale=5000.0, cam_height=1.3, spread=1.2,
                     size=(0.15, 0.45), noise_mm=1.0):
    """Ray-cast ellipsoids standing on the ground plane z = 0 into a 16-bit depth image.
    Returns dict(depth (h,w) u16, bboxes (B,4), labels (B,), Twc (7,), intr (5,), ground (4,), objs (B,10))."""
    rng = np.random.default_rng(seed)
    sx = width / 640.0
    fx, fy, cx, cy = TUM3_K[0] * sx, TUM3_K[1] * sx, TUM3_K[2] * sx, TUM3_K[3] * sx
    # camera at (-2.6, 0, cam_height) looking at the origin region, x right / y down / z forward
    pos = np.array([-2.6, 0.0, cam_height])
    fwd = np.array([0.0, 0.0, 0.3]) - pos; fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rwc = np.stack([right, down, fwd], -1)
    Twc = np.concatenate([pos, _R_to_quat(Rwc[None])[0]])
    # objects: ellipsoids on the ground, spread across the view
    B = n_objs
    s = rng.uniform(size[0], size[1], size=(B, 3))
    ys = np.linspace(-spread, spread, B) if B > 1 else np.zeros(1)
    ctr = np.stack([rng.uniform(-0.3, 0.6, B), ys + 0.05 * rng.standard_normal(B), s[:, 2]], 1)
    yaw = rng.uniform(-0.6, 0.6, B)
    qo = _euler_zyx_to_quat(np.zeros(B), np.zeros(B), yaw)
    objs = np.concatenate([ctr, qo, s], 1)
    # rays
    u, v = np.meshgrid(np.arange(width), np.arange(height))
    dirs_c = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u, dtype=float)], -1)  # camera frame, z = 1
    dirs_w = dirs_c @ Rwc.T
    with np.errstate(divide='ignore', invalid='ignore'):
        t = -pos[2] / dirs_w[..., 2]           # ground plane z_w = 0 (t = camera-frame depth since dirs_c.z = 1)
    zc = np.where((t > 0) & np.isfinite(t), t, np.inf)
    for k in range(B):
        Ro = _quat_to_R(qo[k][None])[0]
        o = (pos - ctr[k]) @ Ro / s[k]
        d = (dirs_w @ Ro) / s[k]
        A = (d * d).sum(-1); Bq = 2 * (d * o).sum(-1); Cq = (o * o).sum() - 1
        disc = Bq * Bq - 4 * A * Cq
        with np.errstate(invalid='ignore'):
            tt = (-Bq - np.sqrt(disc)) / (2 * A)
        hit = (disc > 0) & (tt > 0)
        zc = np.where(hit & (tt < zc), tt, zc)
    zc = zc + (noise_mm * 1e-3) * rng.standard_normal(zc.shape)
    depth = np.where(np.isfinite(zc) & (zc < 13.0), np.clip(np.rint(zc * scale), 0, 65535), 0).astype(np.uint16)
    # boxes from the exact projected outline, padded by 4 px, clipped to the image
    Tcw_R = Rwc.T; Tcw_t = -Rwc.T @ pos
    cams7 = np.concatenate([Tcw_t, _R_to_quat(Tcw_R[None])[0]])[None]
    bb, _, _ = project_bboxes(cams7, objs, (fx, fy, cx, cy), np.zeros(B, int), np.arange(B))
    bb = bb + np.array([-4, -4, 4, 4.0])
    bb[:, [0, 2]] = np.clip(bb[:, [0, 2]], 1, width - 2); bb[:, [1, 3]] = np.clip(bb[:, [1, 3]], 1, height - 2)
    labels = np.array([[28, 41, 0, 58][k % 4] for k in range(B)], dtype=np.int32)  # dual / single symmetry / none / none
    return dict(depth=depth, bboxes=bb, labels=labels, Twc=Twc, intr=np.array([fx, fy, cx, cy, scale]),
                ground=np.array([0.0, 0.0, 1.0, 0.0]), objs=objs)
