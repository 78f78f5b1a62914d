#!/usr/bin/env python
"""Quick C4 mapping-mode timing of esl_optimize_resident (device-resident graph, states restored on the device):
LM iterations/s over N runs + the phase marks of the fused kernel (ESL_LM_TIMING).  Used to compare library variants:
  ESL_HIP_LIB=object-oriented-slam_amd/csrc/variants/C.so python scripts/map_quick.py"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("object-oriented-slam_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
g, c, o, _ = pkg.synth.make_config(cfg, seed=0)
ctx = pkg.Context(0)
ctx.upload_graph(g); ctx.upload_states(c, o); ctx.snapshot_states()
p = pkg.default_lm_params(jacobian_mode=1)
for _ in range(5):
    ctx.restore_states(); rep = ctx.optimize_resident(p)
ctx.synchronize()
best = 1e9
for rnd in range(5):
    t0 = time.perf_counter(); its = 0
    for _ in range(20):
        ctx.restore_states(); rep = ctx.optimize_resident(p); its += rep["iterations"]
    ctx.synchronize()
    dt = time.perf_counter() - t0
    best = min(best, dt / 20)
print("%s lib=%s: %.1f us per optimize (%d iterations, %d trials) -> %.0f LM it/s; chi2 %.6e" % (
    cfg, os.path.basename(pkg.lib.LIB_PATH), best * 1e6, rep["iterations"], rep["total_trials"], rep["iterations"] / best, rep["chi2_final"]), flush=True)
os.environ["ESL_LM_TIMING"] = "1"
ctx.restore_states(); ctx.optimize_resident(p)
ctx.close()
