# A/B of two builds of libesl_hip.so on the C4 SLAM trial: bash scripts/debug/ab.sh <variant .so under csrc/variants> (the other side is the default build)
for v in $1 default; do
  echo VARIANT=$v
  if [ $v != default ]; then export ESL_HIP_LIB=$(pwd)/object-oriented-slam_amd/csrc/variants/$v.so; else unset ESL_HIP_LIB; fi
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --extras /tmp/x$v.json > /tmp/l$v.json 2>/dev/null
  python - $v <<'PY'
import json,sys
d=json.load(open('/tmp/x%s.json'%sys.argv[1]))
print(round(d['value'],3),'it/s', {k:round(v['total_ms']/v['count'],3) for k,v in d['kernel_ms'].items()})
PY
done
