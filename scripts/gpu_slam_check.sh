# quick GPU check of SLAM mode's camera-first elimination (gpurun -- bash scripts/gpu_slam_check.sh <tag>): the parity tests of the
# sparse / dissected / replicated forms, the C4 full-size test, a short C4 bench and its kernel trace -> gpurun_out/<tag>/
R=gpurun_out/${1:-r3j}; mkdir -p $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_slam.py tests/test_gpu_sharded.py -m gpu -q -s -x -k "sparse_interior or nested_dissection or applicab or replicated or distributed or sharded" > $R/tests_sparse.log 2>&1; echo rc=$? >> $R/tests_sparse.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k slam > $R/tests_c4.log 2>&1; echo rc=$? >> $R/tests_c4.log
timeout 600 python bench.py --config C4 --solver ellipsoid --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $R/c4_ellipsoid.json 2> $R/c4_ellipsoid.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/prof_c4 -- python $GRAFT_REPO_ROOT/bench.py --config C4 --solver ellipsoid --no-extras --no-cpu-baseline --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/$R/prof_c4.log 2>&1)
python profiles/summarize_rocpd.py $R/prof_c4/*/*_results.db > $R/c4_cf_kernel_stats.md 2>> $R/prof_c4.log
rm -rf $R/prof_c4
python - $R <<'PY'
import json, sys
for f in ["c4_ellipsoid"]:
    try:
        d=json.loads(open("%s/%s.json"%(sys.argv[1], f)).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, "%.3f it/s"%d["value"], "%.3f ms/step"%d["ms_per_step"], "|", "%.2f TF"%r["achieved"], "frac %.3f"%r["frac"], "avg %.3f ms"%r["avg_launch_ms"], "solve/trial %.3f ms"%r["linear_solve_ms_per_trial"])
        print("    ", {k:(round(v["total_ms"]/max(v["count"],1),4), v["count"]) for k,v in d["kernel_ms"].items()})
    except Exception as e:
        print(f,"FAILED",e)
PY
head -30 $R/c4_cf_kernel_stats.md
grep -n "passed\|failed\|FAILED\|Error\|vs dense\|C4 SLAM\|self test" $R/tests_sparse.log $R/tests_c4.log | cut -c1-250
tail -5 $R/c4_ellipsoid.err | cut -c1-300
