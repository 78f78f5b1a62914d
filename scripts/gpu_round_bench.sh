# round-end measurement set: default bench (with CPU baseline), kernel stats of the same command, PMC traffic, SLAM benches
mkdir -p gpurun_out/r1g
export TMPDIR=/tmp
bash scripts/gpu_pmc.sh r1g > gpurun_out/r1g/pmc.txt 2>&1
mkdir -p profiles && cp gpurun_out/pmc_r1g/traffic.json profiles/r1_pmc_traffic_device_lm.json
timeout 400 python bench.py > gpurun_out/r1g/bench_default.json 2> gpurun_out/r1g/bench_default.err
ESL_BENCH_NO_PROFILE=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r1g/bench_noprofile.json 2> /dev/null
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r1g/prof_bench -- python /root/repo/bench.py --no-cpu-baseline > /root/repo/gpurun_out/r1g/prof_bench.log 2>&1)
python profiles/summarize_rocpd.py gpurun_out/r1g/prof_bench/*/*_results.db > gpurun_out/r1g/prof_bench_summary.md
timeout 300 python bench.py --mode slam --config C3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1g/bench_c3_slam.json 2> gpurun_out/r1g/bench_c3_slam.err
cat gpurun_out/r1g/pmc.txt | tail -8
python - <<'PY'
import json
for f in ["bench_default", "bench_noprofile", "bench_c3_slam"]:
    try:
        d = json.loads(open(f"gpurun_out/r1g/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), round(d["ms_per_step"], 4), json.dumps(d["roofline"])[:900])
        if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"]["value"], d.get("speedup_vs_cpu_port"))
        if "fit" in d: print("  fit", {k: (round(v["ms_per_frame_kernel"], 3), v["samples"]) for k, v in d["fit"].items()}, "stream", round(d["streaming_c5"]["ms_per_frame"], 3))
    except Exception as e:
        print(f, "FAILED", e)
PY
head -12 gpurun_out/r1g/prof_bench_summary.md
