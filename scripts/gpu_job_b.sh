python -m pytest tests/test_gpu_streaming.py tests/test_tangency.py tests/test_visibility.py tests/test_gpu_device_lm.py tests/test_gpu_optimizer.py -m gpu -q -x 2>&1 | tail -30
python - <<'PY'
import importlib, time, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
pkg = importlib.import_module("object-oriented-slam_amd")
sys.argv = ["bench"]
import bench
ctx = pkg.Context(0)
print("streaming", bench.streaming_bench(pkg, ctx))
sc = pkg.synth.make_depth_scene(width=1280, height=960, n_objs=1, seed=11, size=(0.23, 0.28))
P = pkg.lib.default_fit_params(stride=1)
args = (sc["depth"], sc["bboxes"][:1], [28], sc["Twc"], sc["intr"], sc["ground"], P)
for env in ("graph", "nograph"):
    if env == "nograph": os.environ["ESL_FIT_NO_GRAPH"] = "1"
    for _ in range(3): ctx.fit_frame(*args)
    t0 = time.perf_counter()
    for _ in range(20): ctx.fit_frame(*args)
    print("C2 fit host call", env, "%.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
PY
python scripts/map_quick.py C4 2>&1 | grep -v amdgpu; python scripts/map_quick.py C3 2>&1 | grep -v amdgpu
