import importlib, sys, os, numpy as np
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd())
pkg = importlib.import_module("object-oriented-slam_amd")
g, c, o, _ = pkg.synth.make_graph(40, 5, 200, seed=2, slam=True)
# drop every ellipsoid edge: odometry only
z = np.zeros(0, np.int32)
g0 = pkg.Graph(g.K, g.n_cams, 0, g.cam_fixed, z, z, np.zeros((0, 4)), np.zeros(0), z, z, np.zeros((0, 10)), np.zeros(0), z, g.grav_normal, g.grav_weight, g.odom_i, g.odom_j, g.odom_meas.reshape(-1, 7))
cx = pkg.Context(0)
for solver in (0, 1, 2):
    try:
        cg, og, r = cx.optimize(g0, c, np.zeros((0, 10)), pkg.default_lm_params(jacobian_mode=1, linear_solver=solver))
        print("solver", solver, "ok: iterations", r["iterations"], "chi2", r["chi2_initial"], "->", r["chi2_final"], "used", cx.lm_solver_used())
    except Exception as e:
        print("solver", solver, "FAILED:", str(e)[:300])
