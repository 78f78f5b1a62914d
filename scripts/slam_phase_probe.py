#!/usr/bin/env python
"""Per-phase device time of one SLAM-mode optimize() (Schur build / Cholesky / linearise), C3 or C4, without bench.py's extras."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("object-oriented-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
ctx = pkg.Context(0)
g, c, o, _ = pkg.synth.make_config(name, seed=0, slam=True)
p = pkg.default_lm_params(jacobian_mode=1)
ctx.upload_graph(g); ctx.upload_states(c, o); ctx.snapshot_states()
ctx.profile_enable(2)
ctx.synchronize()
t0 = time.perf_counter()
rep = ctx.optimize_resident(p)
ctx.synchronize()
dt = time.perf_counter() - t0
prof = ctx.profile_get()
print(f"{name} SLAM: {rep['iterations']} iterations in {dt:.3f} s -> {rep['iterations'] / dt:.3f} it/s; per phase:",
      {k: (v['count'], round(v['total_ms'] / max(v['count'], 1), 3)) for k, v in prof.items()}, flush=True)
