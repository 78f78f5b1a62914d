# after `gpurun -- bash scripts/gpu_round_bench.sh`: copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked)
TAG=${1:-r2}
R=gpurun_out/$TAG
cp $R/bench_default.json profiles/${TAG}_bench_c4_mapping.json
for n in c3_slam c4_mapping_numeric c3_mapping c4_slam; do cp $R/$n.json profiles/${TAG}_bench_$n.json; done
cp gpurun_out/pmc_$TAG/traffic.json profiles/${TAG}_pmc_traffic_device_lm.json
newest() { ls -t $1 2>/dev/null | head -1; }   # gpurun merges into gpurun_out/: older runs' databases stay next to the new one
python profiles/summarize_rocpd.py $(newest "$R/prof_bench/*/*_results.db") > profiles/${TAG}_bench_default_kernel_stats.md
python profiles/summarize_rocpd.py $(newest "gpurun_out/prof_map_$TAG/*/*_results.db") > profiles/${TAG}_mapping_c4_kernel_stats.md
[ -f $R/fit_kernel_times.txt ] && cp $R/fit_kernel_times.txt profiles/${TAG}_fit_kernel_times.txt
[ -f $R/cholesky_microbench.txt ] && cp $R/cholesky_microbench.txt profiles/${TAG}_cholesky_microbench.txt
[ -f $R/fp64_ceilings.txt ] && cp $R/fp64_ceilings.txt profiles/${TAG}_fp64_ceilings_raw.txt
[ -f $R/cholesky_n32768_kernel_stats.md ] && cp $R/cholesky_n32768_kernel_stats.md profiles/${TAG}_cholesky_n32768_kernel_stats.md
[ -f $R/fp64_ceilings.txt ] && cp $R/fp64_ceilings.txt profiles/${TAG}_fp64_ceilings_raw.txt
[ -f $R/pmc_sq_lm_kernels.md ] && cp $R/pmc_sq_lm_kernels.md profiles/${TAG}_pmc_sq_lm_kernels.md
ls -la profiles
