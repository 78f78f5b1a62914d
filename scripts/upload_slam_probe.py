#!/usr/bin/env python
"""SLAM-mode host-call cost of the drop-in path: esl_graph_upload alone, and esl_optimize (upload + states + LM + download) against
esl_optimize_resident on the same graph and start states.  ESL_UPLOAD_HOST_TIMING=1 adds the host stages."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("object-oriented-slam_amd")
ctx = pkg.Context(0)
names = sys.argv[1:] or ["C3", "C4"]
for name in names:
    for slam in (False, True):
        g, c, o, _ = pkg.synth.make_config(name, seed=0, slam=slam)
        ctx.upload_graph(g)
        t = []
        for _ in range(5):
            ctx.synchronize(); t0 = time.perf_counter(); ctx.upload_graph(g); ctx.synchronize(); t.append(1e3 * (time.perf_counter() - t0))
        print("%s %s: esl_graph_upload %.2f ms (min of 5; %s)" % (name, "SLAM" if slam else "mapping", min(t), " ".join("%.2f" % x for x in t)), flush=True)
    # one LM iteration, same start: the whole drop-in call against the resident one
    g, c, o, _ = pkg.synth.make_config(name, seed=0, slam=True)
    p = pkg.abi.default_lm_params(); p.max_iters = 1
    full, res = [], []
    for rep in range(4):
        ctx.synchronize(); t0 = time.perf_counter(); _, _, r1 = ctx.optimize(g, c, o, p); full.append(1e3 * (time.perf_counter() - t0))
    for rep in range(4):
        ctx.upload_states(c, o)
        ctx.synchronize(); t0 = time.perf_counter(); r2 = ctx.optimize_resident(p); ctx.synchronize(); res.append(1e3 * (time.perf_counter() - t0))
    print("%s SLAM, 1 LM iteration: esl_optimize %.2f ms (%s) | esl_optimize_resident %.2f ms (%s) | difference %.2f ms | chi2 %.9g vs %.9g, trials %d vs %d"
          % (name, min(full), " ".join("%.1f" % x for x in full), min(res), " ".join("%.1f" % x for x in res), min(full) - min(res),
             r1["chi2_final"], r2["chi2_final"], r1["total_trials"], r2["total_trials"]), flush=True)
ctx.close()
