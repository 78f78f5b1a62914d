"""Independent numpy restatement of the reference optimiser path — TEST INFRASTRUCTURE ONLY.

Purpose: a second, differently-structured restatement (homogeneous 4x4 matrices instead of
quaternion/translation pairs, closed-form dual-conic bbox instead of the reference's
inverse+atan2 chain, numpy linear algebra instead of a hand-rolled LDLT) that the C restatement
(oracle/esl_oracle.c) must agree with.  Two independent restatements agreeing, plus the analytic
KATs in tests/test_oracle_kat.py, is the pin for an un-pinned reference (SURVEY.md §8 c).

Citations are relative to /root/reference.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------
# SE3 as 4x4 matrices; 7-vectors x y z qx qy qz qw at the boundary (se3quat.h:144-160)
# ---------------------------------------------------------------------------------------------
def quat_to_R(q):
    x, y, z, w = q
    n = x * x + y * y + z * z + w * w
    s = 2.0 / n  # tolerate slightly non-unit input the way a normalised quaternion would
    return np.array([
        [1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
        [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
        [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]])


def R_to_quat(R):
    """Largest-component extraction, then w >= 0 (normalizeRotation, se3quat.h:345-350)."""
    K = np.array([
        [R[0, 0] - R[1, 1] - R[2, 2], 0, 0, 0],
        [R[0, 1] + R[1, 0], R[1, 1] - R[0, 0] - R[2, 2], 0, 0],
        [R[0, 2] + R[2, 0], R[1, 2] + R[2, 1], R[2, 2] - R[0, 0] - R[1, 1], 0],
        [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], R[0, 0] + R[1, 1] + R[2, 2]]]) / 3.0
    K = K + K.T - np.diag(np.diag(K))
    w_, v_ = np.linalg.eigh(K)
    q = v_[:, np.argmax(w_)]
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def T_from7(v):
    T = np.eye(4)
    T[:3, :3] = quat_to_R(np.asarray(v[3:7], dtype=float))
    T[:3, 3] = v[0:3]
    return T


def T_to7(T):
    return np.concatenate([T[:3, 3], R_to_quat(T[:3, :3])])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=float)


def se3_exp(u):
    """SE3Quat::exp (se3quat.h:274-322) incl. the small-angle quirk R = I + W + W^2, V = R."""
    w, ups = np.asarray(u[:3], float), np.asarray(u[3:], float)
    th = np.linalg.norm(w)
    W = skew(w)
    if th < 1e-5:
        R = np.eye(3) + W + W @ W
        V = R.copy()
        # Quaterniond(R) + normalise: project back to a rotation through the quaternion
        R = quat_to_R(R_to_quat_eigen(R))
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * (W @ W)
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * (W @ W)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ ups
    return T


def R_to_quat_eigen(R):
    """Eigen's Quaterniond(Matrix3d) for trace > 0 (always the case near identity), normalised."""
    t = np.trace(R)
    assert t > 0
    t = np.sqrt(t + 1.0)
    w = 0.5 * t
    t = 0.5 / t
    q = np.array([(R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t, w])
    return q / np.linalg.norm(q)


def se3_log(T):
    """SE3Quat::log (se3quat.h:229-266)."""
    R, t = T[:3, :3], T[:3, 3]
    d = 0.5 * (np.trace(R) - 1)
    dR = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if d > 0.99999:
        w = 0.5 * dR
        W = skew(w)
        Vinv = np.eye(3) - 0.5 * W + (1. / 12.) * (W @ W)
    else:
        th = np.arccos(d)
        w = th / (2 * np.sqrt(1 - d * d)) * dR
        W = skew(w)
        Vinv = np.eye(3) - 0.5 * W + (1 - th / (2 * np.tan(th / 2))) / (th * th) * (W @ W)
    return np.concatenate([w, Vinv @ t])


def T_inv(T):
    Ti = np.eye(4)
    Ti[:3, :3] = T[:3, :3].T
    Ti[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return Ti


# ---------------------------------------------------------------------------------------------
# ellipsoid = (T object->world, scale)
# ---------------------------------------------------------------------------------------------
def obj_from10(v):
    return T_from7(v[:7]), np.asarray(v[7:10], dtype=float).copy()


def obj_to10(T, s):
    return np.concatenate([T_to7(T), s])


def obj_oplus(T, s, u):
    """ellipsoid::exp_update (Ellipsoid.cpp:38-47)."""
    return T @ se3_exp(u[:6]), s + np.asarray(u[6:9])


def cam_oplus(T, u):
    """VertexSE3Expmap::oplusImpl (types_six_dof_expmap.h:73-76)."""
    return se3_exp(u) @ T


def quadric(T, s):
    """ellipsoid::generateQuadric (Ellipsoid.cpp:290-300)."""
    return T @ np.diag([s[0] ** 2, s[1] ** 2, s[2] ** 2, -1.0]) @ T.T


def project_bbox(Tcw, To, s, K):
    """Closed-form tangent lines of the dual conic C* = P Q* P^T (verified equal to
    the reference chain Ellipsoid.cpp:209-273 to 2e-13 px)."""
    fx, fy, cx, cy = K
    Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    P = Km @ Tcw[:3, :]
    C = P @ quadric(To, s) @ P.T
    if not C[2, 2] < 0:
        # the outline is a real ellipse iff the ellipsoid does not meet the camera's principal plane
        # (pi^T Q* pi < 0 for pi = P^T e3); otherwise the reference's A^2/B^2 go negative -> NaN
        # (Ellipsoid.cpp:240-246)
        return np.full(4, np.nan)
    with np.errstate(invalid="ignore"):
        du = np.sqrt(C[0, 2] ** 2 - C[0, 0] * C[2, 2])
        dv = np.sqrt(C[1, 2] ** 2 - C[1, 1] * C[2, 2])
    u = sorted([(C[0, 2] - du) / C[2, 2], (C[0, 2] + du) / C[2, 2]])
    v = sorted([(C[1, 2] - dv) / C[2, 2], (C[1, 2] + dv) / C[2, 2]])
    return np.array([u[0], v[0], u[1], v[1]])


def res_bbox(Tcw, To, s, K, meas):
    """EdgeSE3EllipsoidProj::computeError (BasicEllipsoidEdges.cpp:102-112)."""
    p = project_bbox(Tcw, To, s, K)
    return np.where(np.asarray(meas) >= 5, p - meas, 0.0)


def res_e3d(Tcw, To, s, meas10):
    """EdgeSE3Ellipsoid9DOF::computeError (BasicEllipsoidEdges.cpp:68-77) with
    min_log_error_9dof over the 4 yaw hypotheses (Ellipsoid.cpp:92-117)."""
    Tm, sm = obj_from10(meas10)
    Tmw = T_inv(Tcw) @ Tm
    best, bestn = None, None
    for k in (-1, 0, 1, 2):
        a = k * np.pi / 2
        Rz = np.eye(4)
        Rz[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        sk = sm.copy()
        if k in (-1, 1):
            sk[0], sk[1] = sm[1], sm[0]
        e = np.concatenate([se3_log(T_inv(Tmw @ Rz) @ To), s - sk])
        n = np.linalg.norm(e)
        if best is None or n < bestn:
            best, bestn = e, n
    return best


def res_grav(To, normal):
    """EdgeEllipsoidGravityPlanePrior::computeError (BasicEllipsoidEdges.cpp:129-152)."""
    z = To[:3, 2]
    n = np.asarray(normal[:3], dtype=float)
    c = z @ n / np.linalg.norm(z) / np.linalg.norm(n)
    if c > 1:
        c -= 1e-4
    elif c < -1:
        c += 1e-4
    return np.array([np.arccos(c)])


def res_odom(Ti, Tj, Z):
    """EdgeSE3Expmap::computeError (types_six_dof_expmap.h:90-99)."""
    return se3_log(Z @ Ti @ T_inv(Tj))


# ---------------------------------------------------------------------------------------------
# graph, numeric Jacobians, LM  
# ---------------------------------------------------------------------------------------------
class NpGraph:
    def __init__(self, graph, cams, objs):
        self.g = graph
        self.F, self.N = graph.n_cams, graph.n_objs
        self.cams = [T_from7(c) for c in np.asarray(cams).reshape(-1, 7)]
        self.objs = [obj_from10(o) for o in np.asarray(objs).reshape(-1, 10)]
        self.fixed_cam = np.ones(self.F, bool) if graph.cam_fixed is None else graph.cam_fixed.astype(bool)
        self.edges = []
        g = graph
        for i in range(len(g.odom_i)):
            info = np.ones(6) if g.odom_info is None else g.odom_info.reshape(-1, 6)[i]
            self.edges.append(("odom", int(g.odom_i[i]), int(g.odom_j[i]), T_from7(g.odom_meas.reshape(-1, 7)[i]), info))
        for o in range(self.N):
            for i in np.nonzero(g.grav_obj == o)[0]:
                self.edges.append(("grav", -1, o, None, np.array([g.grav_weight])))
            for i in np.nonzero(g.bbox_obj == o)[0]:
                self.edges.append(("bbox", int(g.bbox_cam[i]), o, g.bbox_meas.reshape(-1, 4)[i], np.full(4, g.bbox_weight[i])))
        for i in range(len(g.e3d_cam)):
            self.edges.append(("e3d", int(g.e3d_cam[i]), int(g.e3d_obj[i]), g.e3d_meas.reshape(-1, 10)[i], np.full(9, g.e3d_weight[i])))

    def drop_nan(self):
        keep = []
        for e in self.edges:
            if e[0] == "bbox":
                r = self.residual(e)
                if np.isnan(r @ (e[4] * r)):
                    continue
            keep.append(e)
        self.edges = keep

    def finalize(self):
        act = []
        for e in self.edges:
            if e[0] == "odom":
                if self.fixed_cam[e[1]] and self.fixed_cam[e[2]]:
                    continue
            act.append(e)
        self.edges = act
        used_c, used_o = set(), set()
        for e in self.edges:
            if e[0] == "odom":
                used_c.update([e[1], e[2]])
            else:
                used_o.add(e[2])
                if e[1] >= 0:
                    used_c.add(e[1])
        self.idx_c, self.idx_o, n = {}, {}, 0
        for c in range(self.F):
            if c in used_c and not self.fixed_cam[c]:
                self.idx_c[c] = n
                n += 6
        for o in range(self.N):
            if o in used_o:
                self.idx_o[o] = n
                n += 9
        self.n = n

    def residual(self, e, cams=None, objs=None):
        cams = self.cams if cams is None else cams
        objs = self.objs if objs is None else objs
        K = self.g.K
        if e[0] == "bbox":
            return res_bbox(cams[e[1]], objs[e[2]][0], objs[e[2]][1], K, e[3])
        if e[0] == "e3d":
            return res_e3d(cams[e[1]], objs[e[2]][0], objs[e[2]][1], e[3])
        if e[0] == "grav":
            return res_grav(objs[e[2]][0], self.g.grav_normal)
        return res_odom(cams[e[1]], cams[e[2]], e[3])

    def chi2(self):
        c = 0.0
        for e in self.edges:
            r = self.residual(e)
            c += r @ (e[4] * r)
        return c

    def jacobians(self, e, delta):
        """central differences (base_binary_edge.hpp:147-197); returns list of (hessian index, J)"""
        out = []
        def num(kind, vid, dim):
            J = np.zeros((len(e[4]), dim))
            for d in range(dim):
                rs = []
                for sgn in (1, -1):
                    u = np.zeros(dim)
                    u[d] = sgn * delta
                    if kind == "cam":
                        cams = list(self.cams)
                        cams[vid] = cam_oplus(self.cams[vid], u)
                        rs.append(self.residual(e, cams=cams))
                    else:
                        objs = list(self.objs)
                        objs[vid] = obj_oplus(*self.objs[vid], u)
                        rs.append(self.residual(e, objs=objs))
                J[:, d] = (rs[0] - rs[1]) / (2 * delta)
            return J
        if e[0] == "odom":
            for v in (e[1], e[2]):
                if v in self.idx_c:
                    out.append((self.idx_c[v], num("cam", v, 6)))
        else:
            if e[1] >= 0 and e[1] in self.idx_c:
                out.append((self.idx_c[e[1]], num("cam", e[1], 6)))
            out.append((self.idx_o[e[2]], num("obj", e[2], 9)))
        return out

    def build(self, delta):
        H = np.zeros((self.n, self.n))
        b = np.zeros(self.n)
        for e in self.edges:
            r = self.residual(e)
            W = np.diag(e[4])
            js = self.jacobians(e, delta)
            for (i, Ji) in js:
                b[i:i + Ji.shape[1]] -= Ji.T @ W @ r
                for (k, Jk) in js:
                    H[i:i + Ji.shape[1], k:k + Jk.shape[1]] += Ji.T @ W @ Jk
        return H, b

    def apply(self, x):
        for c, i in self.idx_c.items():
            self.cams[c] = cam_oplus(self.cams[c], x[i:i + 6])
        for o, i in self.idx_o.items():
            self.objs[o] = obj_oplus(*self.objs[o], x[i:i + 9])


def optimize(graph, cams, objs, max_iters=10, max_trials=10, tau=1e-5, delta=1e-9, drop_nan=True):
    """g2o LM (optimization_algorithm_levenberg.cpp:61-164) with a dense solve of the whole system."""
    G = NpGraph(graph, cams, objs)
    if drop_nan:
        G.drop_nan()
    G.finalize()
    trace = []
    lam, ni, nbad = -1.0, 2.0, 0
    cur = 0.0
    it = 0
    ok = True
    while it < max_iters and ok:
        cur = G.chi2()
        ini = cur
        H, b = G.build(delta)
        if it == 0:
            lam, ni, nbad = tau * np.max(np.abs(np.diag(H))), 2.0, 0
        q, rho = 0, 0.0
        while True:
            bak = (list(G.cams), list(G.objs))
            A = H + lam * np.eye(G.n)
            try:
                np.linalg.cholesky(A)
                x = np.linalg.solve(A, b)
                good = True
            except np.linalg.LinAlgError:
                x = np.zeros(G.n)
                good = False
            G.apply(x)
            tmp = G.chi2() if good else np.finfo(float).max
            rho = (cur - tmp) / (x @ (lam * x + b) + 1e-3)
            if rho > 0 and np.isfinite(tmp):
                alpha = min(1. - (2 * rho - 1) ** 3, 2. / 3.)
                lam *= max(1. / 3., alpha)
                ni = 2.0
                cur = tmp
            else:
                lam *= ni
                ni *= 2
                G.cams, G.objs = bak
            q += 1
            if not (rho < 0 and q < max_trials):
                break
        trace.append((cur, lam, q))
        it += 1
        if q == max_trials or rho == 0:
            ok = False
        else:
            nbad = nbad + 1 if (ini - cur) * 1e3 < ini else 0
            if nbad >= 3:
                ok = False
    cams_out = np.array([T_to7(T) for T in G.cams])
    objs_out = np.array([obj_to10(T, s) for (T, s) in G.objs])
    return cams_out, objs_out, dict(iterations=it, trace=trace, chi2_final=cur, lambda_final=lam)
