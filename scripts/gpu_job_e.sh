#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_plane.py tests/test_adapter_link.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r2_job_e_tests.log
tail -25 gpurun_out/r2_job_e_tests.log
