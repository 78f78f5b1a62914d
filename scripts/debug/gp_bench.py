import importlib, sys, os, json
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("object-oriented-slam_amd")
ctx = pkg.Context(0)
r = bench.ground_plane_bench(pkg, ctx, with_cpu=False)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k not in ("note",)})
