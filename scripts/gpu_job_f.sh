#!/bin/bash
mkdir -p gpurun_out
ESL_CHOL_TIMING=1 python scripts/chol_bench.py 2994 > gpurun_out/r2_job_f_chol.log 2>&1
python scripts/chol_bench.py 1000 2994 6000 8192 12000 16384 32768 >> gpurun_out/r2_job_f_chol.log 2>&1
python -m pytest tests/test_gpu_slam.py -m gpu -q 2>&1 | tail -5 >> gpurun_out/r2_job_f_chol.log
tail -30 gpurun_out/r2_job_f_chol.log
