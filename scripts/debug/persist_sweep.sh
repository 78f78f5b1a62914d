for m in 2 3 4; do for w in 4 6; do echo "MERGE=$m W=$w"; ESL_CHOL_MERGE=$m ESL_CHOL_W=$w timeout 120 python scripts/chol_bench.py 18000 24000; done; done
echo "default timing"; ESL_CHOL_TIMING=1 timeout 120 python scripts/chol_bench.py 18000 2>&1 | grep -v "k_chol_potrf2, last" | tail -6
