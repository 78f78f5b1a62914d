#!/bin/bash
mkdir -p gpurun_out/r4e
export ESL_UPLOAD_HOST_TIMING=1
timeout 300 python scripts/upload_slam_probe.py C4 > gpurun_out/r4e/upload_probe.txt 2>&1; echo "probe rc=$?" >> gpurun_out/r4e/upload_probe.txt
unset ESL_UPLOAD_HOST_TIMING
timeout 900 python -m pytest tests/test_gpu_slam.py tests/test_gpu_fullsize.py tests/test_gpu_streaming.py -m gpu -q -p no:cacheprovider --timeout 600 > gpurun_out/r4e/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4e/gputest.log
tail -12 gpurun_out/r4e/upload_probe.txt; tail -8 gpurun_out/r4e/gputest.log
