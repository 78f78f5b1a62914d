#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_slam.py tests/test_gpu_sharded.py tests/test_golden_vectors.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -8 > gpurun_out/r2_job_j_tests.log
for v in 0 1; do
  if [ $v = 1 ]; then export ESL_SCHUR_ATOMIC=1; fi
  timeout 900 python bench.py --no-cpu-baseline --mode slam --config C3 --steps 3 --warmup 1 > gpurun_out/r2_job_j_c3_$v.json 2> /dev/null
  timeout 900 python bench.py --no-cpu-baseline --mode slam --config C4 --steps 1 --warmup 0 > gpurun_out/r2_job_j_c4_$v.json 2> /dev/null
done
python - <<'PY'
import json
for f in ("c3_0","c3_1","c4_0","c4_1"):
    try:
        d=json.loads(open(f"gpurun_out/r2_job_j_{f}.json").read().strip().splitlines()[-1])
        k=d["kernel_ms"]; print(f, round(d["value"],3), "schur_build", round(k["schur_build"]["total_ms"]/k["schur_build"]["count"],3), "ms; chol", round(k["cholesky_solve"]["total_ms"]/k["cholesky_solve"]["count"],2), "chi2", d.get("chi2"))
    except Exception as e: print(f, "FAILED", e)
PY
cat gpurun_out/r2_job_j_tests.log
