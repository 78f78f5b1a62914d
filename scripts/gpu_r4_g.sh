#!/bin/bash
mkdir -p gpurun_out/r4g
timeout 200 python scripts/fit_timing.py > gpurun_out/r4g/fit_timing.txt 2>&1
timeout 600 python -m pytest tests/test_fit.py tests/test_fit_independent.py -m gpu -q -p no:cacheprovider --timeout 300 > gpurun_out/r4g/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4g/gputest.log
timeout 300 bash scripts/gpu_prof_fit.sh r4g > gpurun_out/r4g/fit_kernels.txt 2>&1
tail -12 gpurun_out/r4g/fit_timing.txt; tail -6 gpurun_out/r4g/gputest.log; tail -45 gpurun_out/r4g/fit_kernels.txt
