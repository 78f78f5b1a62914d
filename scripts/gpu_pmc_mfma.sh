# MFMA-pipe utilisation of one C4 SLAM optimize (5 trials): SQ_VALU_MFMA_BUSY_CYCLES against the kernels' durations (PMC pass with
# --kernel-trace only).  gpurun -- bash scripts/gpu_pmc_mfma.sh <tag>   -> gpurun_out/<tag>/pmc_mfma_c4_slam.{json,txt}
TAG=${1:-r3}
R=gpurun_out/$TAG; mkdir -p $R
export TMPDIR=/tmp
ROOT=$(pwd)
rocprofv3 -L 2>/dev/null | grep -i "mfma\|SQ_BUSY_CYCLES\|GRBM_GUI_ACTIVE\|SQ_BUSY_CU" | head -20 > $R/pmc_mfma_counters_available.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $ROOT/$R/pmc_mfma -- python $ROOT/bench.py --no-extras --no-cpu-baseline --steps 1 --warmup 0 > $ROOT/$R/pmc_mfma.log 2>&1)
python - $R <<'PY'
import glob, json, sqlite3, statistics, sys
R = sys.argv[1]
out = {}
for f in glob.glob(R + "/pmc_mfma/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = db.execute("select kernel_name, dispatch_id, counter_name, sum(value), max(grid_size) from counters_collection group by kernel_name, dispatch_id, counter_name").fetchall()
    dur = {}
    try:
        for name, did, st, en in db.execute("select name, dispatch_id, start, end from kernels"):
            dur[did] = (en - st)
    except Exception as e:
        print("no kernel durations:", e)
    per = {}
    for name, did, cn, val, grid in rows:
        per.setdefault((name.split("(")[0], did), {"grid": grid})[cn] = val
    for (name, did), v in per.items():
        out.setdefault(name, []).append(dict(v, dur_ns=dur.get(did)))
res = {}
for name, lst in out.items():
    if not any(k in name for k in ("k_chol_update", "k_cf_seg_syrk", "k_cf_T_gather", "k_chol_panel", "k_chol_potrf")):
        continue
    lst.sort(key=lambda d: -d.get("grid", 0))
    big = [d for d in lst if d.get("grid", 0) == lst[0].get("grid", 0)]
    m = statistics.median(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for d in big)
    b = statistics.median(d.get("SQ_BUSY_CYCLES", 0) for d in big)
    res[name] = {"dispatches_of_the_largest_grid": len(big), "grid_threads": lst[0].get("grid"), "SQ_VALU_MFMA_BUSY_CYCLES": m, "SQ_BUSY_CYCLES": b,
                 "all_dispatches": len(lst), "sum_MFMA_BUSY_all": sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for d in lst), "sum_SQ_BUSY_all": sum(d.get("SQ_BUSY_CYCLES", 0) for d in lst)}
json.dump(res, open(R + "/pmc_mfma_c4_slam.json", "w"), indent=1)
for k, v in res.items():
    print(k, v)
PY
rm -rf $R/pmc_mfma
cat $R/pmc_mfma_counters_available.txt; tail -3 $R/pmc_mfma.log | cut -c1-300
