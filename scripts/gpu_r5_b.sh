# round 5, second GPU call: SLAM-mode append == rebuild, the mid-size full-run SLAM parity with its retuned tolerances, the streaming file
R=gpurun_out/r5b; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_streaming.py -m gpu -q -s > $R/gputest.log 2>&1; echo rc=$? >> $R/gputest.log
grep -n "passed\|failed" $R/gputest.log | tail -3
grep -n "frame [0-9]*:\|SLAM-mode append\|2,000 cams\|   the GPU\|   checker\|Jacobians vs that\|^E  " $R/gputest.log | head -40
