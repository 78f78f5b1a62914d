# round 3, GPU batch 4: camera-first chain kernels reworked -- SLAM parity tests, C3 / C4 timings, kernel traces
R=gpurun_out/r3i; mkdir -p $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_fullsize.py::test_c4_slam_schur_solve_full_size > $R/tests_a.log 2>&1; echo rc=$? >> $R/tests_a.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k slam > $R/tests_c4.log 2>&1; echo rc=$? >> $R/tests_c4.log
timeout 300 python bench.py --config C3 --solver ellipsoid --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $R/c3_ellipsoid.json 2> $R/c3_ellipsoid.err
timeout 600 python bench.py --config C4 --solver ellipsoid --no-extras --no-cpu-baseline --steps 2 --warmup 1 > $R/c4_ellipsoid.json 2> $R/c4_ellipsoid.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/prof_c3 -- python $GRAFT_REPO_ROOT/bench.py --config C3 --solver ellipsoid --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$R/prof_c3.log 2>&1)
python profiles/summarize_rocpd.py $R/prof_c3/*/*_results.db > $R/c3_cf_kernel_stats.md 2>> $R/prof_c3.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/prof_c4 -- python $GRAFT_REPO_ROOT/bench.py --config C4 --solver ellipsoid --no-extras --no-cpu-baseline --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/$R/prof_c4.log 2>&1)
python profiles/summarize_rocpd.py $R/prof_c4/*/*_results.db > $R/c4_cf_kernel_stats.md 2>> $R/prof_c4.log
rm -rf $R/prof_c3 $R/prof_c4
python - <<'PY'
import json
for f in ["c3_ellipsoid","c4_ellipsoid"]:
    try:
        d=json.loads(open("gpurun_out/r3i/%s.json"%f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, "%.3f it/s"%d["value"], "%.3f ms/step"%d["ms_per_step"], "|", "%.2f TF"%r["achieved"], "frac %.3f"%r["frac"], "avg %.3f ms"%r["avg_launch_ms"], "solve/trial %.3f ms"%r["linear_solve_ms_per_trial"])
        print("    ", {k:(round(v["total_ms"]/max(v["count"],1),4), v["count"]) for k,v in d["kernel_ms"].items()})
    except Exception as e:
        print(f,"FAILED",e)
PY
head -24 $R/c3_cf_kernel_stats.md; head -24 $R/c4_cf_kernel_stats.md
grep -n "passed\|failed\|FAILED\|variant\|C4 SLAM" $R/tests_a.log $R/tests_c4.log | cut -c1-250
