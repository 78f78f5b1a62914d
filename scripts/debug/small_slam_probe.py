"""Where does a SLAM-mode LM iteration go at streaming sizes (<= 119 free cameras, 20 ellipsoids)?  Host-call times of the step API."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("object-oriented-slam_amd")
import os
SIZES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(120, 20, 2400), (500, 50, 5000)]
for (F, N, E) in SIZES:
    g, c, o, _ = pkg.synth.make_graph(F, N, E, seed=3, slam=True)
    ctx = pkg.Context(0)
    ctx.upload_graph(g); ctx.upload_states(c, o); ctx.snapshot_states()
    for solver in (1, 2):
        p = pkg.default_lm_params(jacobian_mode=1, linear_solver=solver)
        ctx.restore_states(); ctx.optimize_resident(p)
        ctx.restore_states()
        t0 = time.perf_counter(); rep = ctx.optimize_resident(p); ctx.synchronize(); dt = time.perf_counter() - t0
        ctx.restore_states()
        ctx.lm_begin(p)
        tl = tt = tc = 0.0; n = 20
        lin = ctx.lm_linearize()
        lam = 1e-5 * lin.max_diag
        for _ in range(n):
            t0 = time.perf_counter(); ctx.lm_linearize(); t1 = time.perf_counter(); ctx.lm_try_step(lam); t2 = time.perf_counter(); ctx.lm_commit(False); t3 = time.perf_counter()
            tl += t1 - t0; tt += t2 - t1; tc += t3 - t2
        print("F %d N %d solver %d: optimize %.3f ms for %d it / %d trials = %.3f ms per it | host call us: linearize %.1f try_step %.1f commit %.1f" % (
            F, N, solver, 1e3 * dt, rep["iterations"], rep["total_trials"], 1e3 * dt / max(rep["iterations"], 1), 1e6 * tl / n, 1e6 * tt / n, 1e6 * tc / n))
    ctx.close()
