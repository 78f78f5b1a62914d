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

# Per kernel AND grid size: a device-driven LM run leaves a few no-op launches (they exit at their first instruction,
# 4-6 us) and bench.py also runs small streaming graphs through the same kernels, so the plain average above mixes
# three populations.  "live" = launches longer than half the longest one of that (kernel, grid).
try:
    import statistics
    rows = db.execute("select name, grid_x, (end - start) / 1000.0 from kernels").fetchall()
    groups = {}
    for n, g, t in rows:
        groups.setdefault((n.split("(")[0], g), []).append(t)
    print("\n| kernel | grid (threads) | launches | live | median live us | mean live us |")
    print("|---|---|---|---|---|---|")
    for (n, g), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        if len(v) < 3 or "rocclr" in n:
            continue
        live = [x for x in v if x > 0.5 * max(v)]
        print(f"| `{n}` | {g} | {len(v)} | {len(live)} | {statistics.median(live):.2f} | {statistics.mean(live):.2f} |")
except Exception as e:
    print("\n(per-grid table unavailable:", e, ")")
