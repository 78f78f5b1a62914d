#!/bin/bash
# round 2: ground-plane kernels + rsqrt diagonal block of the Cholesky
mkdir -p gpurun_out
python -m pytest tests/test_plane.py tests/test_gpu_slam.py tests/test_harness.py tests/test_gpu_streaming.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r2_job_d_tests.log
ESL_CHOL_TIMING=1 python scripts/chol_bench.py 2994 8192 > gpurun_out/r2_job_d_chol.log 2>&1
tail -15 gpurun_out/r2_job_d_tests.log; tail -40 gpurun_out/r2_job_d_chol.log
