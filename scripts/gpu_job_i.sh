#!/bin/bash
mkdir -p gpurun_out
(hipcc -w -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && timeout 300 /tmp/mfma_peak) > gpurun_out/r2_fp64_ceilings.txt 2>&1
timeout 900 python bench.py --no-cpu-baseline --mode slam --config C4 --steps 1 --warmup 0 > gpurun_out/r2_job_i_c4slam.json 2> gpurun_out/r2_job_i_c4slam.err
timeout 900 python bench.py --no-cpu-baseline --mode slam --config C3 --steps 3 --warmup 1 > gpurun_out/r2_job_i_c3slam.json 2> gpurun_out/r2_job_i_c3slam.err
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r2_job_i_tests.log
cat gpurun_out/r2_fp64_ceilings.txt | head -8
python - <<'PY'
import json
for f in ("c4slam","c3slam"):
    try:
        d=json.loads(open(f"gpurun_out/r2_job_i_{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d.get("kernel_ms"))
    except Exception as e: print(f, "FAILED", e)
PY
cat gpurun_out/r2_job_i_tests.log
