for i in 1 2; do timeout 100 scripts/debug/bin/upd_v_probe 18000 3744 1 0 | grep "^lds\|^v\|differ"; done
bash scripts/debug/chol_ab.sh "8192 18000 24000 32768" prev default
for v in prev default; do
  if [ $v != default ]; then export ESL_HIP_LIB=$(pwd)/object-oriented-slam_amd/csrc/variants/$v.so; else unset ESL_HIP_LIB; fi
  echo "bench $v"; timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 1 --extras /tmp/x$v.json > /tmp/l$v.json 2>/dev/null
  python - $v <<PY
import json,sys
d=json.load(open("/tmp/x%s.json"%sys.argv[1]))
print(round(d["value"],3),"it/s", {k:round(v["total_ms"]/v["count"],3) for k,v in d["kernel_ms"].items()})
PY
done
