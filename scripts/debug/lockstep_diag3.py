import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("object-oriented-slam_amd")
from oracle import pyoracle as po
import test_gpu_streaming as t
np.set_printoptions(precision=9, linewidth=220)
g, c, o, _ = pkg.synth.make_graph(60, 20, 1200, seed=3)
ctx = pkg.Context(0)
p = pkg.default_lm_params(jacobian_mode=1)
objs = o.copy()
for f in range(0, 12):
    gf = t.graph_upto(pkg, g, f)
    before = objs.copy()
    _, objs, _ = ctx.optimize(gf, c[:f + 1], objs, p)
np.savez(os.path.join(ROOT, "gpurun_out", "r4b", "before11.npz"), before=before, after=objs)
f, k = 11, 15
for grav in (False, True):
    sub = gf.subset_objects([k])
    gs = sub if grav else t.without_gravity(pkg, sub)
    st = before[k:k + 1]
    for jac in (0, 1):
        ctx.upload_graph(gs); ctx.upload_states(c[:f + 1], st)
        ctx.lm_begin(pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6))
        lin = ctx.lm_linearize()
        bg = ctx.lm_download(1, 9)
        H, b, fidx, chi = po.build_system(gs, c[:f + 1], st, delta=1e-6)
        print("single-ellipsoid graph, gravity %s, jac %d: chi2 gpu %.9e checker %.9e | b gpu %s | b checker %s" % (grav, jac, lin.chi2, chi, bg, b))
print("state", before[k]); print("meas", sub.e3d_meas.reshape(-1, 10)[0], "w", sub.e3d_weight[0], "cam", c[sub.e3d_cam[0]])
print("checker res_e3d", po.res_e3d(c[sub.e3d_cam[0]], before[k], sub.e3d_meas.reshape(-1, 10)[0]))
ctx.close()
