#!/bin/bash
mkdir -p gpurun_out/r4h
{
for W in 3 4 6 8; do
  echo "== persistent, W=$W"
  ESL_CHOL_PERSISTENT=1 ESL_CHOL_W=$W timeout 120 python scripts/chol_bench.py 12000 18000 24000
done
echo "== launch path (ESL_CHOL_PERSISTENT=0)"
ESL_CHOL_PERSISTENT=0 timeout 120 python scripts/chol_bench.py 12000 18000 24000
} > gpurun_out/r4h/chol_w.txt 2>&1
cat gpurun_out/r4h/chol_w.txt
