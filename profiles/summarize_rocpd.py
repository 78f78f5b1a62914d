#!/usr/bin/env python
"""Dump the per-kernel stats (rocprofv3 --kernel-trace --stats, rocpd sqlite output of ROCm 7.2) as markdown."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("| kernel | calls | total us | avg us | % |")
print("|---|---|---|---|---|")
for name, calls, tot, avg, pct in rows:
    print(f"| `{name.split('(')[0]}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.1f} |")
try:
    r = db.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, lds_size, grid_x, workgroup_x "
                   "from kernels group by name").fetchall()
    print("\n| kernel | vgpr | agpr | sgpr | scratch B | lds B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|")
    for n, v, a, s, sc, l, gx, wx in r:
        print(f"| `{n.split('(')[0]}` | {v} | {a} | {s} | {sc} | {l} | {gx} | {wx} |")
except Exception as e:  # schema differences between rocprofv3 versions
    print("\n(kernel resource table unavailable:", e, ")")
