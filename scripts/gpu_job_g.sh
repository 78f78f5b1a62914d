#!/bin/bash
mkdir -p gpurun_out
(hipcc -w -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && timeout 300 /tmp/mfma_peak) > gpurun_out/r2_fp64_ceilings.txt 2>&1
ESL_CHOL_TIMING=1 python scripts/chol_bench.py 2994 2>&1 | tail -2 > gpurun_out/r2_job_g_chol.log
python scripts/chol_bench.py 1000 2994 6000 8192 12000 16384 32768 >> gpurun_out/r2_job_g_chol.log 2>&1
python -m pytest tests/test_gpu_slam.py -m gpu -q 2>&1 | tail -3 >> gpurun_out/r2_job_g_chol.log
cat gpurun_out/r2_fp64_ceilings.txt; tail -12 gpurun_out/r2_job_g_chol.log
