# round 3, GPU batch 2: the camera-first elimination -- parity tests, then C3 / C4 SLAM timings of both eliminations
R=gpurun_out/r3b; mkdir -p $R
timeout 900 python -m pytest tests/test_gpu_slam.py tests/test_golden_vectors.py tests/test_harness.py tests/test_adapter_link.py tests/test_gpu_device_lm.py tests/test_gpu_streaming.py tests/test_gpu_sharded.py tests/test_tangency.py -m gpu -q -s --durations=10 > $R/tests_a.log 2>&1; echo rc=$? >> $R/tests_a.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k slam > $R/tests_c4.log 2>&1; echo rc=$? >> $R/tests_c4.log
for sv in ellipsoid camera; do
  timeout 300 python bench.py --config C3 --solver $sv --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $R/c3_$sv.json 2> $R/c3_$sv.err
  timeout 600 python bench.py --config C4 --solver $sv --no-extras --no-cpu-baseline --steps 2 --warmup 1 > $R/c4_$sv.json 2> $R/c4_$sv.err
done
python - <<'PY'
import json
for f in ["c3_ellipsoid","c3_camera","c4_ellipsoid","c4_camera"]:
    try:
        d=json.loads(open("gpurun_out/r3b/%s.json"%f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, "%.3f it/s"%d["value"], "%.3f ms/step"%d["ms_per_step"], "trials/step", d["config"]["lm_trials_per_step"], "|", r["kernel"][:50], "%.2f TF"%r["achieved"], "frac %.3f"%r["frac"], "avg %.3f ms"%r["avg_launch_ms"], "solve/trial %.3f ms"%r["linear_solve_ms_per_trial"])
        print("    ", {k:(round(v["total_ms"]/max(v["count"],1),4), v["count"]) for k,v in d["kernel_ms"].items()})
    except Exception as e:
        print(f,"FAILED",e)
PY
tail -3 $R/tests_a.log $R/tests_c4.log
