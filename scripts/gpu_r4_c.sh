run() { ESL_CHOL_DEBUG=$3 ESL_CHOL_TIMEOUT_MS=500 ESL_CHOL_PERSISTENT=$2 timeout 90 python scripts/chol_bench.py $1 2>&1 | grep "^n=\|Error" | sed "s/^/persistent=$2 debug=$3 /"; }
run 450 1 0; run 2994 1 0; run 8192 1 0; run 18000 1 0; run 18000 1 6; run 18000 1 1; run 12000 1 0; run 32768 1 0; run 130 1 0; run 1153 1 0
