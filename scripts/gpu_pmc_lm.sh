# SQ counters of the mapping-mode LM kernels: where the wave cycles go (VALU issue vs waiting)
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_lm_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $OUT -- python /root/repo/scripts/prof_map.py C4 5 > $OUT/log.txt 2>&1
cd /root/repo
python scripts/pmc_lm_summary.py $OUT | tee $OUT/summary.md
