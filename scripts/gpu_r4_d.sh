#!/bin/bash
# round 4, call d: SLAM-mode blobs (upload probe), then the whole GPU suite (half-turn fix, streaming lock-step, persistent Cholesky default)
mkdir -p gpurun_out/r4d
export ESL_UPLOAD_HOST_TIMING=1
timeout 300 python scripts/upload_slam_probe.py > gpurun_out/r4d/upload_probe.txt 2>&1; echo "probe rc=$?" >> gpurun_out/r4d/upload_probe.txt
unset ESL_UPLOAD_HOST_TIMING
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > gpurun_out/r4d/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4d/gputest.log
tail -5 gpurun_out/r4d/upload_probe.txt; tail -15 gpurun_out/r4d/gputest.log
