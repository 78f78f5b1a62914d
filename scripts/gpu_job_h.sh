#!/bin/bash
mkdir -p gpurun_out
python scripts/chol_bench.py 2994 8192 16384 32768 > gpurun_out/r2_job_h_chol.log 2>&1
python -m pytest tests/test_gpu_slam.py -m gpu -q 2>&1 | tail -3 >> gpurun_out/r2_job_h_chol.log
bash scripts/gpu_pmc_chol.sh r2h 16384 >> gpurun_out/r2_job_h_chol.log 2>&1
tail -25 gpurun_out/r2_job_h_chol.log
