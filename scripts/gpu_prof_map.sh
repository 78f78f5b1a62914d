# per-kernel durations of the mapping-mode optimiser (rocprofv3 kernel trace), printed as markdown
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_map_$1
rm -rf $OUT; mkdir -p $OUT
python scripts/prof_map.py C4 20
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -- python /root/repo/scripts/prof_map.py C4 20 > $OUT/log.txt 2>&1
cd /root/repo
tail -1 $OUT/log.txt
python profiles/summarize_rocpd.py $OUT/*/*_results.db | tee $OUT/summary.md
