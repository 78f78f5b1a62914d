# Round-end measurement set (run on the GPU box through gpurun); everything lands under gpurun_out/ -- only that directory
# travels back -- and scripts/collect_profiles.sh copies the judged summaries into profiles/ afterwards.
#   default bench (with CPU baseline) + rocprofv3 kernel stats of the same command, PMC traffic of the LM kernels,
#   optimiser-only kernel stats, fit pipeline kernel times, dense-Cholesky micro-benchmark, FP64 ceilings, other configs
TAG=${1:-r2}
R=gpurun_out/$TAG
mkdir -p $R
export TMPDIR=/tmp
bash scripts/gpu_pmc.sh $TAG > $R/pmc.txt 2>&1
timeout 600 python bench.py > $R/bench_default.json 2> $R/bench_default.err
ESL_BENCH_NO_PROFILE=1 timeout 300 python bench.py --no-cpu-baseline > $R/bench_noprofile.json 2> /dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$R/prof_bench -- python /root/repo/bench.py --no-cpu-baseline > /root/repo/$R/prof_bench.log 2>&1)
python profiles/summarize_rocpd.py $R/prof_bench/*/*_results.db > $R/bench_default_kernel_stats.md
bash scripts/gpu_prof_map.sh $TAG > $R/prof_map.txt 2>&1
python profiles/summarize_rocpd.py gpurun_out/prof_map_$TAG/*/*_results.db > $R/mapping_c4_kernel_stats.md
bash scripts/gpu_prof_fit.sh $TAG > $R/fit_kernel_times.txt 2>&1
bash scripts/gpu_pmc_lm.sh $TAG > $R/pmc_sq_lm_kernels.md 2>&1
timeout 300 python scripts/chol_bench.py 2994 8192 16384 24576 32768 > $R/cholesky_microbench.txt 2>&1
bash scripts/gpu_prof_chol.sh $TAG 32768 > $R/prof_chol.txt 2>&1
python profiles/summarize_rocpd.py gpurun_out/prof_chol_$TAG/*/*_results.db > $R/cholesky_n32768_kernel_stats.md
(hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && timeout 120 /tmp/mfma_peak) > $R/fp64_ceilings.txt 2>&1
run() { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > $R/$name.json 2> $R/$name.err; }
run c3_slam --mode slam --config C3 --steps 3 --warmup 1
run c4_mapping_numeric --no-slam --jacobian numeric --steps 5 --warmup 1
run c3_mapping --no-slam --config C3
run c4_slam --mode slam --config C4 --steps 1 --warmup 0
python - $R <<'PY'
import json, sys
R = sys.argv[1]
for f in ["bench_default", "bench_noprofile", "c3_slam", "c4_mapping_numeric", "c3_mapping", "c4_slam"]:
    try:
        d = json.loads(open(f"{R}/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, round(d["value"], 3), "it/s", round(d["ms_per_step"], 4), "ms/step |", r["kernel"][:40], round(r["achieved"], 2), r["unit"], "frac", round(r["frac"], 4),
              "| avg launch ms", round(r.get("avg_launch_ms", 0), 5), "| traffic", r.get("traffic"))
        if r.get("valu_issue_floor"): print("   valu floor", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r["valu_issue_floor"].items() if k != "note"})
        if "cpu_baseline" in d: print("   cpu", round(d["cpu_baseline"]["value"], 4), "x", round(d.get("speedup_vs_cpu_port", 0)))
        if "fit" in d: print("   fit", {k: (round(v["ms_per_frame_kernel"], 3), round(v["ms_per_frame_host_call"], 3), v["samples"], round(v.get("cpu_port_ms_per_frame", 0), 2)) for k, v in d["fit"].items()},
                             "stream", round(d["streaming_c5"]["ms_per_frame"], 3))
    except Exception as e:
        print(f, "FAILED", e)
PY
cat $R/pmc.txt | tail -5
cat $R/cholesky_microbench.txt
cat $R/fp64_ceilings.txt
tail -12 $R/fit_kernel_times.txt
head -9 $R/mapping_c4_kernel_stats.md
