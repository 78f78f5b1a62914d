for sp in 1 2 3 4; do echo "SPECIAL=$sp"; ESL_CHOL_SPECIAL=$sp timeout 120 python scripts/chol_bench.py 18000; done
for f in 64 96 128 192 256; do echo "FILLER=$f"; ESL_CHOL_FILLER=$f timeout 120 python scripts/chol_bench.py 18000; done
