R=gpurun_out/r4b; mkdir -p $R
timeout 1200 python -m pytest tests/test_gpu_streaming.py -m gpu -q -s --durations=5 > $R/gputest3.log 2>&1; echo rc=$? >> $R/gputest3.log
grep -n "passed\|failed\|Error\|error" $R/gputest3.log | tail -12; grep "streaming, all\|lock-step beyond" $R/gputest3.log | cut -c1-3000
