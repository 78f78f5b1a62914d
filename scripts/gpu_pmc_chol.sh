# MFMA-pipe / stall counters of the dense Cholesky kernels (one SQ pass + GRBM), n from $2
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_chol_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT -- python /root/repo/scripts/chol_bench.py ${2:-16384} > $OUT/log.txt 2>&1
cd /root/repo
tail -2 $OUT/log.txt
python - <<PY
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob("$OUT/*/*_results.db")[0])
rows = db.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by kernel_name, dispatch_id, counter_name").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen=set()
for k, d, c, v in rows:
    k = k.split("(")[0]
    agg[k][c] += v
    if (k,d) not in seen: seen.add((k,d)); cnt[k]+=1
for k, cs in agg.items():
    if "chol" not in k: continue
    gui = cs.get("GRBM_GUI_ACTIVE", 0)
    print(f"{k:28s} n={cnt[k]:4d} " + " ".join(f"{c}={v:.3g}" for c, v in sorted(cs.items())))
    if gui: print(f"   MFMA busy / (GUI_ACTIVE x 1024 SIMDs) = {cs.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(gui*1024):.3f}   (x256 CUs: {cs.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(gui*256):.3f})")
PY
