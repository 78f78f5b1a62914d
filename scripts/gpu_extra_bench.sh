# the non-default configurations quoted in DESIGN.md §5
mkdir -p gpurun_out/r1x
run() { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/r1x/$name.json 2> gpurun_out/r1x/$name.err; }
run c4_mapping_numeric --jacobian numeric --steps 5 --warmup 1
run c3_mapping --config C3
run c4_slam --mode slam --config C4 --steps 1 --warmup 0
python - <<'PY'
import json
for f in ["c4_mapping_numeric", "c3_mapping", "c4_slam"]:
    try:
        d = json.loads(open(f"gpurun_out/r1x/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, round(d["value"], 3), "it/s", round(d["ms_per_step"], 3), "ms/step", r["kernel"], round(r["achieved"], 2), r["unit"], round(r["frac"], 4), {k: (v["count"], round(v["total_ms"], 2)) for k, v in d["kernel_ms"].items()})
    except Exception as e:
        print(f, "FAILED", e, open(f"gpurun_out/r1x/{f}.err").read()[-500:])
PY
