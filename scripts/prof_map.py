#!/usr/bin/env python
"""Optimiser-only workload for rocprofv3: C4 mapping graph, `reps` LM runs from the same start."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("object-oriented-slam_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
jac = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g, c, o, _ = pkg.synth.make_config(cfg, seed=0)
ctx = pkg.Context(0)
ctx.upload_graph(g); ctx.upload_states(c, o); ctx.snapshot_states()
p = pkg.default_lm_params(jacobian_mode=jac)
for _ in range(3):
    ctx.restore_states(); ctx.optimize_resident(p)
ctx.synchronize()
t0 = time.perf_counter(); its = 0
for _ in range(reps):
    ctx.restore_states(); its += ctx.optimize_resident(p)["iterations"]
ctx.synchronize()
dt = time.perf_counter() - t0
print(f"{cfg} jac={jac}: {its} LM iterations in {dt*1e3:.2f} ms -> {its/dt:.0f} it/s, {dt/its*1e6:.1f} us/iteration")
