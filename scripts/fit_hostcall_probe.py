#!/usr/bin/env python
"""Why does the C2 fit's host call take 0.7 ms in some processes and 2.3 ms in others?  Times it before / after the
mapping-mode and SLAM-mode optimiser ran in the same context, with and without graph replay."""
import importlib, os, sys, time
if os.environ.get('PROBE_TORCH_FIRST'):
    import torch
    torch.cuda.synchronize()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("object-oriented-slam_amd")
ctx = pkg.Context(0)
sc = pkg.synth.make_depth_scene(width=1280, height=960, n_objs=1, seed=11, size=(0.23, 0.28))
P = pkg.lib.default_fit_params(stride=1)
args = (sc["depth"], sc["bboxes"][:1], [28], sc["Twc"], sc["intr"], sc["ground"], P)
def t_fit(tag, n=20):
    for _ in range(3): ctx.fit_frame(*args)
    t0 = time.perf_counter()
    for _ in range(n): ctx.fit_frame(*args)
    print(f"{tag:40s} {1e3 * (time.perf_counter() - t0) / n:.3f} ms per call", flush=True)
t_fit("fresh context")
g, c, o, _ = pkg.synth.make_config("C3", seed=0)
ctx.upload_graph(g); ctx.upload_states(c, o); ctx.optimize_resident(pkg.default_lm_params(jacobian_mode=1))
t_fit("after C3 mapping optimize")
g4, c4, o4, _ = pkg.synth.make_config("C4", seed=0)
ctx.upload_graph(g4); ctx.upload_states(c4, o4); ctx.optimize_resident(pkg.default_lm_params(jacobian_mode=1))
t_fit("after C4 mapping optimize")
ctx.profile_enable(1); ctx.optimize_resident(pkg.default_lm_params(jacobian_mode=1)); ctx.profile_enable(0)
t_fit("after profiled C4 mapping optimize")
os.environ["ESL_FIT_NO_GRAPH"] = "1"
t_fit("same, ESL_FIT_NO_GRAPH=1")
del os.environ["ESL_FIT_NO_GRAPH"]
gs, cs, os_, _ = pkg.synth.make_config("C3", seed=0, slam=True) if "slam" in pkg.synth.make_config.__code__.co_varnames else (None,) * 4
if gs is not None:
    ctx.upload_graph(gs); ctx.upload_states(cs, os_); ctx.optimize_resident(pkg.default_lm_params(jacobian_mode=1))
    t_fit("after C3 SLAM optimize")
ctx.upload_graph(g4); ctx.upload_states(c4, o4); ctx.snapshot_states()
p = pkg.default_lm_params(jacobian_mode=1)
for _ in range(300):
    ctx.restore_states(); ctx.optimize_resident(p)
ctx.synchronize()
t_fit("after 300 C4 mapping optimizes")
ctx.profile_enable(2)
for _ in range(3):
    ctx.restore_states(); ctx.optimize_resident(p)
prof = ctx.profile_get()
ctx.profile_enable(False)
t_fit("after a level-2 profiled pass")
