# per-kernel durations of one dense factorisation (rocprofv3 kernel trace): n from $2
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_chol_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -- python /root/repo/scripts/chol_bench.py ${2:-32768} > $OUT/log.txt 2>&1
cd /root/repo
tail -1 $OUT/log.txt
python profiles/summarize_rocpd.py $OUT/*/*_results.db | head -30
