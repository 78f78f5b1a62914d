# round 3, GPU batch 3: cut-chain debug, default bench line, rocprof of the C3 camera-first run
R=gpurun_out/r3c; mkdir -p $R
export TMPDIR=/tmp
timeout 300 python scripts/debug/cut_graph.py > $R/cut_graph.txt 2>&1
timeout 900 python bench.py --steps 2 --warmup 1 > $R/bench_default.json 2> $R/bench_default.err; echo rc=$? >> $R/bench_default.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/prof_c3 -- python $GRAFT_REPO_ROOT/bench.py --config C3 --solver ellipsoid --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$R/prof_c3.log 2>&1)
python profiles/summarize_rocpd.py $R/prof_c3/*/*_results.db > $R/c3_cf_kernel_stats.md 2>> $R/prof_c3.log
head -40 $R/c3_cf_kernel_stats.md
cat $R/cut_graph.txt
tail -c 1500 $R/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c/bench_default.json").read().strip().splitlines()[-1])
def short(x, depth=0):
    if isinstance(x, dict):
        return {k: short(v, depth+1) for k, v in x.items() if k not in ("note","sample","kernel_ms","sampling","measured_ceiling","host","split_s","workload")} if depth < 3 else "..."
    if isinstance(x, float): return round(x, 5)
    if isinstance(x, str): return x[:60]
    return x
print(json.dumps(short(d), indent=1)[:6000])
PY
