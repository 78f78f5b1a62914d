# per-kernel durations of the fit pipeline (rocprofv3 kernel trace)
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_fit_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && ESL_FIT_NO_TIMING=1 rocprofv3 --kernel-trace --stats -d $OUT -- python /root/repo/scripts/fit_timing.py > $OUT/log.txt 2>&1
cd /root/repo
python - <<PY
import sqlite3, glob, statistics
db = sqlite3.connect(glob.glob("$OUT/*/*_results.db")[0])
rows = db.execute("select name, (end-start)/1000.0, grid_x, start from kernels order by start").fetchall()
for n, t, g, _ in rows:
    if "k_fit" in n: print(f"{n.split('(')[0]:28s} grid={g:7d} {t:8.1f} us")
PY
