"""Which side of the 1e-4 camera gap (VERDICT r5 'what's weak') is the checker's camera-first solve?  One LM trial on the 2,048-camera
chain of tests/test_gpu_slam.py: the checker's camera-first increment against a dense Cholesky solve of the SAME (H + lambda I) x = b
(the checker's own linearisation, po.build_system) with iterative refinement in long double residuals."""
import importlib, os, sys, time
import numpy as np, scipy.linalg as sl
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("object-oriented-slam_amd")
from oracle import pyoracle as po
from oracle import np_oracle as npo

args = dict(n_cams=int(sys.argv[1]) if len(sys.argv) > 1 else 2048, n_objs=150, n_bbox_target=8 * 2048, seed=43, slam=True)
g, c, o, _ = pkg.synth.make_graph(args["n_cams"], args["n_objs"], args["n_bbox_target"], seed=args["seed"], slam=True)
p1 = pkg.default_lm_params(numeric_delta=1e-6, max_iters=1)
t0 = time.time()
H, b, fidx, chi = po.build_system(g, c, o, delta=1e-6)
n = len(b)
print("system order", n, "built in %.1f s" % (time.time() - t0), "chi2", chi)
lam = p1.tau * np.abs(np.diag(H)).max()
A = H + lam * np.eye(n)
t0 = time.time()
cf = sl.cho_factor(A, lower=True)
x = sl.cho_solve(cf, b)
for it in range(4):
    r = (b.astype(np.longdouble) - np.array([np.dot(A[i].astype(np.longdouble), x.astype(np.longdouble)) for i in range(n)], dtype=np.longdouble))
    dx = sl.cho_solve(cf, r.astype(np.float64))
    x = (x.astype(np.longdouble) + dx).astype(np.float64)
    print("refine", it, "|r|/|b| %.2e |dx|/|x| %.2e" % (np.linalg.norm(r.astype(float)) / np.linalg.norm(b), np.linalg.norm(dx) / np.linalg.norm(x)))
print("dense solve %.1f s; lambda %.3e; cond estimate (diag ratio) %.2e" % (time.time() - t0, lam, np.diag(A).max() / np.diag(A).min()))
np.save("/tmp/x_refined.npy", x)
for solver, name in ((po.ORACLE_CAMFIRST, "camfirst"), (po.ORACLE_BLOCK, "block")):
    if solver == po.ORACLE_BLOCK and n > 8000 and os.environ.get("WITH_BLOCK") != "1":
        continue
    t0 = time.time()
    co, oo, ro = po.optimize(g, c, o, p1, solver=solver)
    # apply the refined x with the checker's own retractions
    nc = g.n_cams
    cr = np.array(c, dtype=float).reshape(-1, 7).copy(); orr = np.array(o, dtype=float).reshape(-1, 10).copy()
    for v in range(nc):
        if fidx[v] >= 0:
            cr[v] = po.cam_oplus(cr[v], x[fidx[v]:fidx[v] + 6])
    for v in range(g.n_objs):
        k = fidx[nc + v]
        if k >= 0:
            orr[v] = po.obj_oplus(orr[v], x[k:k + 9])
    e = max(np.linalg.norm(npo.se3_log(npo.T_inv(npo.T_from7(a)) @ npo.T_from7(bb))) for a, bb in zip(cr, co))
    step = max(np.linalg.norm(npo.se3_log(npo.T_inv(npo.T_from7(a)) @ npo.T_from7(bb))) for a, bb in zip(np.array(c).reshape(-1, 7), co))
    eo = np.abs(orr - oo).max()
    print("%s: %.0f s, trials %s, lambda0 %s; cameras vs refined dense solve %.3e (largest camera step %.3e), ellipsoids %.3e" % (
        name, time.time() - t0, ro["trace_trials"], ro.get("trace_lambda"), e, step, eo))
    np.savez("/tmp/chk_%s.npz" % name, cams=co, objs=oo, cr=cr, orr=orr)
