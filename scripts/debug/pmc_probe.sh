# PMC passes over one run of a probe binary: bash scripts/debug/pmc_probe.sh <out-dir> <binary> [args...]
OUT=$1; shift
mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
i=0
for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $ROOT/$OUT/p$i --output-format csv -- $ROOT/"$@" > $ROOT/$OUT/p$i.log 2>&1)
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$set" <<'PY' | tee -a $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-40:]
    acc[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
byk = collections.defaultdict(list)
for (k, d), v in acc.items(): byk[k].append(v)
for k, lst in byk.items():
    names = sorted(lst[0])
    print(k, "dispatches", len(lst), " ".join(f"{n}={sorted(v[n] for v in lst)[len(lst)//2]:.4g}" for n in names))
PY
done
