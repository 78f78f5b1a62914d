# HBM traffic of the mapping-mode kernels: two separate --pmc passes (kernel trace only), then the JSON summary
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_$1
rm -rf $OUT; mkdir -p $OUT/fetch $OUT/write
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python /root/repo/scripts/prof_map.py C4 5 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python /root/repo/scripts/prof_map.py C4 5 > $OUT/write.log 2>&1
cd /root/repo
python scripts/pmc_summary.py $OUT/fetch $OUT/write $OUT/traffic.json k_chunk k_lm_step k_obj
