#!/usr/bin/env python
"""Where esl_graph_upload's host time goes at C4 (run with ESL_UPLOAD_HOST_TIMING=1)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("object-oriented-slam_amd")
ctx = pkg.Context(0)
g, c, o, _ = pkg.synth.make_config("C4", seed=0)
for _ in range(6):
    t0 = time.perf_counter()
    ctx.upload_graph(g)
    print(f"upload {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
