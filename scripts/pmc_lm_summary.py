#!/usr/bin/env python
"""Where the wave cycles of the LM kernels go (rocprofv3 --pmc SQ_* pass, rocpd sqlite): python scripts/pmc_lm_summary.py <dir>"""
import collections
import glob
import sqlite3
import sys

db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
rows = db.execute("select kernel_name, dispatch_id, counter_name, sum(value), max(grid_size) from counters_collection "
                  "group by kernel_name, dispatch_id, counter_name").fetchall()
per = collections.defaultdict(dict)
for k, d, c, v, g in rows:
    per[(k.split("(")[0], d, g)][c] = v
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for (k, d, g), cs in per.items():
    if cs.get("SQ_INSTS_VALU", 0) < 2e4:      # launches queued behind a finished LM run exit at once
        continue
    for c, v in cs.items():
        agg[(k, g)][c] += v
    cnt[(k, g)] += 1
print("SQ counters per kernel (live launches only).  SQ_WAVE_CYCLES and the ACTIVE / WAIT counters are per-wave sums in the same")
print("unit, so their ratios are fractions of a wave's residency; with w waves per SIMD the VALU pipe is busy ~ w x 'VALU active'.\n")
print("| kernel | grid | live launches | VALU wave-instr / launch | VALU active | any instr active | waiting (waitcnt / barrier) | issue stall | LDS instr / launch |")
print("|---|---|---|---|---|---|---|---|---|")
for (k, g), cs in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = cs.get("SQ_WAVE_CYCLES", 0)
    if not wc or "esl" not in k:
        continue
    n = cnt[(k, g)]
    print(f"| `{k}` | {g} | {n} | {cs.get('SQ_INSTS_VALU', 0) / n:.3g} | {cs.get('SQ_ACTIVE_INST_VALU', 0) / wc:.1%} | "
          f"{cs.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1%} | {cs.get('SQ_WAIT_ANY', 0) / wc:.1%} | {cs.get('SQ_WAIT_INST_ANY', 0) / wc:.1%} | "
          f"{cs.get('SQ_INSTS_LDS', 0) / n:.3g} |")
