import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("object-oriented-slam_amd")
abi = importlib.import_module("object-oriented-slam_amd.abi")
import test_plane as tp
from oracle import pyoracle as po
intr, frames = tp.fixture()
depth = frames[0][0]
which = sys.argv[1]
ctx = pkg.Context(0)
if which == "g0":
    print(ctx.extract_ground_plane(depth, intr, abi.default_plane_params(refine=0)))
elif which == "p0":
    r = ctx.extract_planes(depth, intr, abi.default_plane_params(refine=0)); print(r["n_planes"], r["sizes"])
elif which == "p1":
    r = ctx.extract_planes(depth, intr); print(r["n_planes"], r["sizes"])
elif which == "time":
    for d in (frames[0][0], frames[1][0]):
        ctx.extract_ground_plane(d, intr)
        t0 = time.perf_counter()
        for _ in range(20): ctx.extract_ground_plane(d, intr)
        print("ground plane host call %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
