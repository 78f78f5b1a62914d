#!/usr/bin/env python
"""esl_graph_upload of the C4 / C3 graphs in SLAM mode (host-side list building included): ms per call."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("object-oriented-slam_amd")
ctx = pkg.Context(0)
for name in ("C3", "C4"):
    for slam in (False, True):
        g, c, o, _ = pkg.synth.make_config(name, seed=0, slam=slam)
        ctx.upload_graph(g)
        t = []
        for _ in range(5):
            ctx.synchronize(); t0 = time.perf_counter(); ctx.upload_graph(g); ctx.synchronize(); t.append(1e3 * (time.perf_counter() - t0))
        print("%s %s: esl_graph_upload %.2f ms (min of 5; %s)" % (name, "SLAM" if slam else "mapping", min(t), " ".join("%.2f" % x for x in t)))
ctx.close()
