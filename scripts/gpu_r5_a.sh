# round 5, first GPU call: the whole -m gpu suite (new: replicated-mode contract, mid-size full-run SLAM parity, both N>1 designs in
# one bench line), the default bench line with the DRIVER's command, the Cholesky micro-benchmark of this box
R=gpurun_out/r5a; mkdir -p $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 > $R/gputest.log 2>&1; echo rc=$? >> $R/gputest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/bench_default.json 2> $R/bench_default.err; echo rc=$? >> $R/bench_default.err
{ timeout 300 python scripts/chol_bench.py 450 2994 8192 18000; ESL_CHOL_TIMING=1 timeout 120 python scripts/chol_bench.py 18000 2>&1 | grep -v "k_chol_potrf2, last"; } > $R/cholesky_microbench.txt 2>&1
grep -n "passed\|failed\|error" $R/gputest.log | tail -5
grep -v "^\.\+" $R/gputest.log | grep -i "mid\|2,000 cams\|both designs\|contract\|FAILED\|Error" | head -30
python - $R <<'PY'
import json, sys
R = sys.argv[1]
d = json.loads(open(f"{R}/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"], 3), "it/s", round(d["ms_per_step"], 2), "ms/step")
print("roofline:", r["kernel"][:70], "| achieved", round(r["achieved"], 2), "frac", round(r["frac"], 4), "avg launch ms", round(r["avg_launch_ms"], 3), "traffic", r["traffic"])
print("selection:", r.get("selection"))
s = r["secondary"]; print("secondary:", s["kernel"][:60], round(s["achieved"], 2), round(s["frac"], 4), round(s["avg_launch_ms"], 3))
print("trial_ms", r["trial_ms"], "trial_frac", r["trial_frac"])
print("kernel_ms per trial:", {k: round(v["total_ms"] / max(d["config"]["lm_trials_per_step"] * d["steps"], 1), 3) for k, v in d["kernel_ms"].items()})
print("cpu_baseline", d["cpu_baseline"]["value"], d.get("speedup_vs_cpu_baseline"))
for k in ("slam_reduced_camera", "mapping", "slam_c3"):
    if k in d: print("   ", k, round(d[k]["value"], 3), "it/s")
print("fit", {k: (round(v["ms_per_frame_kernel"], 3), round(v["ms_per_frame_host_call"], 3)) for k, v in d["fit"].items()}, "stream", round(d["streaming_c5"]["ms_per_frame"], 3))
PY
cat $R/cholesky_microbench.txt
tail -3 $R/bench_default.err
