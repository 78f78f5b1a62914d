# round 5: the refinement pass on the GPU (plane tests, the harness, the adapters)
R=gpurun_out/r5c; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_plane.py tests/test_harness.py tests/test_adapter_link.py -m gpu -q -s > $R/gputest.log 2>&1; echo rc=$? >> $R/gputest.log
grep -n "passed\|failed" $R/gputest.log | tail -3
grep -n "^E  " $R/gputest.log | head -30
python - <<'PY'
import importlib, time, sys, numpy as np
sys.path.insert(0, "tests")
pkg = importlib.import_module("object-oriented-slam_amd")
import test_plane as tp
ctx = pkg.Context(0)
intr, frames = tp.fixture()
for rf in (0, 1):
    p = pkg.abi.default_plane_params(refine=rf)
    for _ in range(3): ctx.extract_ground_plane(frames[0][0], intr, p)
    t0 = time.perf_counter()
    for _ in range(20): r = ctx.extract_ground_plane(frames[0][0], intr, p)
    print("ground plane 640x480 clip frame, refine=%d: %.2f ms per host call, %d planes, %d pixels" % (rf, 1e3 * (time.perf_counter() - t0) / 20, r["n_planes"], r["n_pixels"]))
PY
