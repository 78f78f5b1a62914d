"""debug aid (round 4): ellipsoid with a single 3-D edge in a gravity-free graph -- first linearisation after upload"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("object-oriented-slam_amd")
from oracle import pyoracle as po
import test_gpu_streaming as t
np.set_printoptions(precision=6, linewidth=220)
n_frames = 60
g, c, o, _ = pkg.synth.make_graph(n_frames, 20, 20 * n_frames, seed=3)
ctx = pkg.Context(0)
p = pkg.default_lm_params(jacobian_mode=1)
objs = o.copy()
for f in range(0, 12):
    gf = t.graph_upto(pkg, g, f)
    before = objs.copy()
    _, objs, _ = ctx.optimize(gf, c[:f + 1], objs, p)
f = 11
gn = t.without_gravity(pkg, gf)
N, F = gn.n_objs, gn.n_cams
k = 15
def show(tag, cx, jac):
    cx.upload_graph(gn); cx.upload_states(c[:f + 1], before)
    cx.lm_begin(pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6))
    _, cur = cx.download_states()
    H, b, fidx, chi = po.build_system(gn, c[:f + 1], cur, delta=1e-6)
    i = fidx[F + k]
    for rep in range(2):
        lin = cx.lm_linearize()
        Hg = cx.lm_download(0, N * 45).reshape(N, 45); bg = cx.lm_download(1, N * 9).reshape(N, 9)
        print("%s jac %d linearize #%d: chi2 gpu %.9e checker %.9e diff %.3e | state == before: %s" % (tag, jac, rep, lin.chi2, chi, lin.chi2 - chi, np.array_equal(cur, before)))
        print("     b gpu    ", bg[k]); print("     b checker", b[i:i + 9])
        print("     diag H gpu    ", np.diag(t.unpack45(Hg[k]))); print("     diag H checker", np.diag(H[i:i + 9, i:i + 9]))
show("shared context", ctx, 0)
show("shared context", ctx, 1)
fresh = pkg.Context(0)
show("fresh context", fresh, 0)
show("fresh context", fresh, 1)
# the same ellipsoid WITH one gravity edge somewhere else in the graph (ellipsoid 0 only)
gg = pkg.Graph(gn.K, gn.n_cams, gn.n_objs, None, gn.bbox_cam, gn.bbox_obj, gn.bbox_meas, gn.bbox_weight, gn.e3d_cam, gn.e3d_obj, gn.e3d_meas, gn.e3d_weight, (0,), gf.grav_normal, gf.grav_weight)
gn = gg
show("gravity on ellipsoid 0 only, fresh context", fresh, 0)
sub = gg.subset_objects([k])
print("edges of ellipsoid", k, ":", len(sub.bbox_cam), "bbox,", len(sub.e3d_cam), "3-D; 3-D chunks in the graph:", len(np.unique(gg.e3d_obj)), "ellipsoids with 3-D edges;", "e3d edges", len(gg.e3d_cam))
print("e3d_obj order:", gg.e3d_obj[:40])
ctx.close(); fresh.close()
