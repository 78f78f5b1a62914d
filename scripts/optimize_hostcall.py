#!/usr/bin/env python
"""PCIe-inclusive cost of the drop-in call: esl_optimize with host buffers (validate + sort + stage + upload the graph,
upload the states, run the LM, download the states), C4 and the streaming-size graph."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("object-oriented-slam_amd")
ctx = pkg.Context(0)
p = pkg.default_lm_params(jacobian_mode=1)
for name, (g, c, o, _) in (("C4", pkg.synth.make_config("C4", seed=0)), ("C3", pkg.synth.make_config("C3", seed=0)),
                           ("60 cams / 20 ellipsoids / 1.2k edges", pkg.synth.make_graph(60, 20, 1200, seed=1))):
    for _ in range(2):
        ctx.optimize(g, c, o, p)
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        _, _, rep = ctx.optimize(g, c, o, p)
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.upload_graph(g)
    du = (time.perf_counter() - t0) / n
    print(f"{name}: esl_optimize (host buffers) {dt * 1e3:.3f} ms per call, of which esl_graph_upload {du * 1e3:.3f} ms; "
          f"{rep['iterations']} LM iterations -> {rep['iterations'] / dt:.0f} it/s PCIe- and upload-inclusive", flush=True)
