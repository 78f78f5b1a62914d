#!/bin/bash
# round 4, last call: the GPU suite and the default bench line at HEAD, the Cholesky micro-benchmark with the fused chain
R=gpurun_out/r4f2
mkdir -p $R
timeout 1200 python -m pytest tests -m gpu -q -s --durations=10 -p no:cacheprovider > $R/gputest.log 2>&1; echo rc=$? >> $R/gputest.log
timeout 900 python bench.py > $R/bench_default.json 2> $R/bench_default.err; echo rc=$? >> $R/bench_default.err
{ echo "== default selection (one launch from n = 4,096 to 30,000, launch per step elsewhere; fused chain)"; timeout 300 python scripts/chol_bench.py 450 2994 4096 6000 8192 12000 18000 24000 32768
  echo "== ESL_CHOL_FUSE=0 (strips under the block and the next block's update as worker tasks)"; ESL_CHOL_FUSE=0 timeout 300 python scripts/chol_bench.py 4096 8192 12000 18000 24000
  echo "== ESL_CHOL_PERSISTENT=0 (launch per step everywhere)"; ESL_CHOL_PERSISTENT=0 timeout 300 python scripts/chol_bench.py 2994 4096 6000 8192 12000 18000 24000
  echo "== ESL_CHOL_TIMING=1, n = 18000: where the persistent kernel's time goes"; ESL_CHOL_TIMING=1 timeout 120 python scripts/chol_bench.py 18000 2>&1 | grep -v "k_chol_potrf2, last"; } > $R/cholesky_microbench.txt 2>&1
python - $R <<'PY'
import json, sys
R = sys.argv[1]
d = json.loads(open(f"{R}/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("bench_default", round(d["value"], 3), "it/s", round(d["ms_per_step"], 3), "ms/step | frac", round(r["frac"], 4), "chol ms", round(r.get("cholesky_order_9N_ms_per_trial", 0), 2))
for k in ("slam_reduced_camera", "mapping", "slam_c3"):
    if k in d: print("   ", k, round(d[k]["value"], 3), "it/s")
print("    fit", {k: (round(v["ms_per_frame_kernel"], 3), round(v["ms_per_frame_host_call"], 3)) for k, v in d["fit"].items()}, "stream", round(d["streaming_c5"]["ms_per_frame"], 3))
PY
grep -n "passed\|failed" $R/gputest.log | tail -2; tail -30 $R/cholesky_microbench.txt
