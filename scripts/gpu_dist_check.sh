# quick GPU check of the distributed factorisation on ONE GPU (host transport + a 1-rank RCCL communicator): the sharded / replicated
# tests with the panel messages on their own stream and on the compute stream, then the forced-dist bench both ways
export TMPDIR=/tmp
mkdir -p gpurun_out/r3y
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q -s > gpurun_out/r3y/sharded_overlap.log 2>&1; echo rc=$? >> gpurun_out/r3y/sharded_overlap.log
ESL_CHOL_DIST_OVERLAP=0 timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q > gpurun_out/r3y/sharded_seq.log 2>&1; echo rc=$? >> gpurun_out/r3y/sharded_seq.log
grep -n "passed\|failed\|rc=\|replicated graph\|Error" gpurun_out/r3y/sharded_overlap.log gpurun_out/r3y/sharded_seq.log | cut -c1-220
for ov in 1 0; do
export ESL_CHOL_DIST_OVERLAP=$ov
ESL_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forced dist, 1 rank, overlap', os.environ['ESL_CHOL_DIST_OVERLAP'], round(d['value'],3), 'it/s chi2', d['chi2']['final'])"
done
