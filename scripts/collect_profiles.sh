# after `gpurun -- bash scripts/gpu_round_bench.sh`: copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked)
R=gpurun_out/r1
cp $R/bench_default.json profiles/r1_bench_c4_mapping.json
for n in c3_slam c4_mapping_numeric c3_mapping c4_slam; do cp $R/$n.json profiles/r1_bench_$n.json; done
cp gpurun_out/pmc_r1/traffic.json profiles/r1_pmc_traffic_device_lm.json
python profiles/summarize_rocpd.py $R/prof_bench/*/*_results.db > profiles/r1_bench_default_kernel_stats.md
python profiles/summarize_rocpd.py gpurun_out/prof_map_r1/*/*_results.db > profiles/r1_mapping_c4_kernel_stats.md
[ -f $R/fit_kernel_times.txt ] && cp $R/fit_kernel_times.txt profiles/r1_fit_kernel_times.txt
[ -f $R/cholesky_microbench.txt ] && cp $R/cholesky_microbench.txt profiles/r1_cholesky_microbench.txt
[ -f $R/fp64_ceilings.txt ] && cp $R/fp64_ceilings.txt profiles/r1_fp64_ceilings_raw.txt
ls -la profiles
