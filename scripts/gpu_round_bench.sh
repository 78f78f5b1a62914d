# Round-end measurement set (run on the GPU box through gpurun); everything lands under gpurun_out/<tag>/ -- only gpurun_out/
# travels back -- and scripts/collect_profiles.sh copies the judged summaries into profiles/ afterwards.
#   the -m gpu test-suite, the default bench line (C4 SLAM timed + all records, with CPU baselines) and rocprofv3 kernel stats of
#   the SAME command, PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes) of one C4 SLAM optimize and of the mapping-mode LM,
#   C3 SLAM kernel stats, mapping kernel stats, fit kernel times, Cholesky micro-benchmark, other configs
TAG=${1:-r6}
R=gpurun_out/$TAG
mkdir -p $R
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests -m gpu -q -s --durations=15 > $R/gputest.log 2>&1; echo rc=$? >> $R/gputest.log
# the DRIVER's command (BENCH_rNN.json: `python3 bench.py --gpus 1 --steps 20 --warmup 5`), and rocprofv3 kernel stats of the same command
# (round 6: stdout is the driver's line alone, < 4 KB; the full record -- what these scripts summarise -- is the --extras side file)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --extras $R/bench_default.json > $R/bench_default_line.json 2> $R/bench_default.err; echo rc=$? >> $R/bench_default.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$R/prof_bench -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --extras $ROOT/$R/prof_bench_extras.json > $ROOT/$R/prof_bench.json 2> $ROOT/$R/prof_bench.log)
python profiles/summarize_rocpd.py $(ls -t $R/prof_bench/*/*_results.db | head -1) > $R/bench_default_kernel_stats.md
rm -rf $R/prof_bench
# HBM traffic of ONE C4 SLAM optimize (5 trials, camera-first elimination): two separate --pmc passes, kernel trace only
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $ROOT/$R/pmc_slam_$c -- python $ROOT/bench.py --no-extras --no-cpu-baseline --steps 1 --warmup 0 --extras /tmp/pmc_extras.json > $ROOT/$R/pmc_slam_$c.log 2>&1)
done
python scripts/pmc_summary.py $R/pmc_slam_FETCH_SIZE $R/pmc_slam_WRITE_SIZE $R/pmc_traffic_c4_slam.json k_chol_update k_chol_persist k_chol_backsub k_cf_ k_slam_linearize k_chol_potrf k_chol_panel > $R/pmc_slam.txt 2>&1
rm -rf $R/pmc_slam_FETCH_SIZE $R/pmc_slam_WRITE_SIZE
# ... and of the mapping-mode LM kernels
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $ROOT/$R/pmc_map_$c -- python $ROOT/scripts/prof_map.py C4 5 > $ROOT/$R/pmc_map_$c.log 2>&1)
done
python scripts/pmc_summary.py $R/pmc_map_FETCH_SIZE $R/pmc_map_WRITE_SIZE $R/pmc_traffic_device_lm.json k_chunk k_lm_step k_obj > $R/pmc_map.txt 2>&1
rm -rf $R/pmc_map_FETCH_SIZE $R/pmc_map_WRITE_SIZE
# kernel stats: C3 SLAM (both eliminations), C4 mapping
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$R/prof_c3 -- python $ROOT/bench.py --config C3 --solver ellipsoid --no-extras --no-cpu-baseline --steps 5 --warmup 2 --extras /tmp/c3_extras.json > $ROOT/$R/prof_c3.log 2>&1)
python profiles/summarize_rocpd.py $(ls -t $R/prof_c3/*/*_results.db | head -1) > $R/c3_slam_camera_first_kernel_stats.md; rm -rf $R/prof_c3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$R/prof_map -- python $ROOT/scripts/prof_map.py C4 20 > $ROOT/$R/prof_map.log 2>&1)
python profiles/summarize_rocpd.py $(ls -t $R/prof_map/*/*_results.db | head -1) > $R/mapping_c4_kernel_stats.md; rm -rf $R/prof_map
{ echo "== default selection (one launch from n = 4,096 to 30,000, launch per step elsewhere)"; timeout 300 python scripts/chol_bench.py 450 2994 4096 6000 8192 12000 18000 24000 32768
  echo "== ESL_CHOL_PERSISTENT=0 (launch per step everywhere)"; ESL_CHOL_PERSISTENT=0 timeout 300 python scripts/chol_bench.py 2994 4096 6000 8192 12000 18000 24000
  echo "== ESL_CHOL_TIMING=1, n = 18000: where the persistent kernel's time goes"; ESL_CHOL_TIMING=1 timeout 120 python scripts/chol_bench.py 18000 2>&1 | grep -v "k_chol_potrf2, last"; } > $R/cholesky_microbench.txt 2>&1
ESL_UPLOAD_HOST_TIMING=1 timeout 300 python scripts/upload_slam_probe.py C4 2>&1 | tail -12 > $R/slam_upload_probe.txt
timeout 200 python scripts/fit_timing.py > $R/fit_stage_timing.txt 2>&1
timeout 300 bash scripts/gpu_prof_fit.sh $TAG > $R/fit_kernel_times.txt 2>&1; rm -rf gpurun_out/prof_fit_$TAG
run() { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline --no-extras --extras $R/$name.json "$@" > $R/${name}_line.json 2> $R/$name.err; }
run c3_slam_camera_first --config C3 --solver ellipsoid --steps 5 --warmup 2
run c3_slam_reduced_camera --config C3 --solver camera --steps 5 --warmup 2
run c4_slam_reduced_camera --config C4 --solver camera --steps 2 --warmup 1
run c4_mapping --mode mapping
run c4_mapping_numeric --mode mapping --jacobian numeric --steps 5 --warmup 1
run c3_mapping --mode mapping --config C3
python - $R <<'PY'
import json, sys
R = sys.argv[1]
for f in ["bench_default", "c3_slam_camera_first", "c3_slam_reduced_camera", "c4_slam_reduced_camera", "c4_mapping", "c4_mapping_numeric", "c3_mapping"]:
    try:
        d = json.loads(open(f"{R}/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, round(d["value"], 3), "it/s", round(d["ms_per_step"], 4), "ms/step |", r["kernel"][:48], round(r["achieved"], 2), r["unit"], "frac", round(r["frac"], 4),
              "| avg launch ms", round(r.get("avg_launch_ms", 0), 5))
        for k in ("slam_reduced_camera", "mapping", "slam_c3"):
            if k in d: print("   ", k, round(d[k]["value"], 3), "it/s")
        if "cpu_baseline" in d: print("    cpu", d["cpu_baseline"]["value"], "x", round(d.get("speedup_vs_cpu_baseline", d.get("speedup_vs_cpu_port", 0))))
        if "secondary" in r: print("    secondary:", r["secondary"]["kernel"][:40], round(r["secondary"]["achieved"], 2), "frac", round(r["secondary"]["frac"], 4), "| trial_frac", r.get("trial_frac"), "| traffic", r.get("traffic"))
        if "fit" in d: print("    fit", {k: (round(v["ms_per_frame_kernel"], 3), round(v["ms_per_frame_host_call"], 3)) for k, v in d["fit"].items()},
                             "stream", round(d["streaming_c5"]["ms_per_frame"], 3), "ground plane", round(d["ground_plane"]["ms_per_frame_host_call"], 3))
    except Exception as e:
        print(f, "FAILED", e)
PY
grep -n "passed\|failed" $R/gputest.log | tail -3
cat $R/pmc_slam.txt | tail -12
cat $R/pmc_map.txt | tail -5
cat $R/cholesky_microbench.txt
head -14 $R/bench_default_kernel_stats.md
