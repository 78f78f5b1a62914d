# after `gpurun -- bash scripts/gpu_round_bench.sh <tag>`: copy the judged summaries from gpurun_out/<tag>/ (scratch) into profiles/
# (tracked) under the round's name:   bash scripts/collect_profiles.sh <tag> [round-name, default r3]
TAG=${1:-r6}
RN=${2:-r6}
R=gpurun_out/$TAG
cp $R/bench_default.json profiles/${RN}_bench_default.json
cp $R/bench_default_line.json profiles/${RN}_bench_default_line.json
cp $R/bench_default_kernel_stats.md profiles/${RN}_bench_default_kernel_stats.md
for n in c3_slam_camera_first c3_slam_reduced_camera c4_slam_reduced_camera c4_mapping c4_mapping_numeric c3_mapping; do [ -s $R/$n.json ] && cp $R/$n.json profiles/${RN}_bench_$n.json; done
cp $R/pmc_traffic_c4_slam.json profiles/${RN}_pmc_traffic_c4_slam.json
cp $R/pmc_traffic_device_lm.json profiles/${RN}_pmc_traffic_device_lm.json
cp $R/c3_slam_camera_first_kernel_stats.md profiles/${RN}_c3_slam_camera_first_kernel_stats.md
cp $R/mapping_c4_kernel_stats.md profiles/${RN}_mapping_c4_kernel_stats.md
cp $R/cholesky_microbench.txt profiles/${RN}_cholesky_microbench.txt
for n in slam_upload_probe fit_stage_timing fit_kernel_times; do [ -s $R/$n.txt ] && cp $R/$n.txt profiles/${RN}_$n.txt; done
grep -v "^$" $R/gputest.log | grep -v "^\.\+$" | grep -v "^streaming frame [0-9]*: lock-step" | tail -150 > profiles/${RN}_gputest_tail.txt
ls -la profiles
