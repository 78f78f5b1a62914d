# A/B of libesl_hip.so variants on the dense solver: bash scripts/debug/chol_ab.sh "<sizes>" variant... (default = the in-tree build), two alternating rounds
SIZES=$1; shift
for round in 1 2; do
  for v in "$@"; do
    if [ $v != default ]; then export ESL_HIP_LIB=$(pwd)/object-oriented-slam_amd/csrc/variants/$v.so; else unset ESL_HIP_LIB; fi
    echo "== $v (round $round)"; timeout 300 python scripts/chol_bench.py $SIZES
  done
done
