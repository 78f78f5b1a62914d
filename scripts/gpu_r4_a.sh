# round 4, first GPU call: the whole -m gpu suite (one-launch back-substitution in every SLAM solve, the new parity tests),
# Cholesky micro-benchmark with the back-substitution A/B, streaming parity probe
R=gpurun_out/r4a; mkdir -p $R
timeout 900 python -m pytest tests -m gpu -q -s -x --durations=10 > $R/gputest.log 2>&1; echo rc=$? >> $R/gputest.log
timeout 300 python scripts/chol_bench.py 450 2994 8192 18000 > $R/chol_new.txt 2>&1
ESL_CHOL_BACKSUB_LAUNCHES=1 timeout 300 python scripts/chol_bench.py 450 2994 8192 18000 > $R/chol_old.txt 2>&1
timeout 600 python scripts/streaming_parity_probe.py > $R/stream_probe.txt 2>&1
tail -5 $R/gputest.log; cat $R/chol_new.txt $R/chol_old.txt; tail -3 $R/stream_probe.txt
