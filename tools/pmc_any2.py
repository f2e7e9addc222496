#!/usr/bin/env python3
"""Per-kernel (largest grid only) averages of every counter in rocprofv3 --pmc databases.  usage: pmc_any2.py db1 [db2 ...]"""
import sqlite3
import sys
res = {}
for p in sys.argv[1:]:
    cur = sqlite3.connect(p).cursor()
    q = ("select kernel_name, grid_size, counter_name, count(*), avg(value) from counters_collection "
         "group by kernel_name, grid_size, counter_name")
    best = {}
    for name, grid, cn, n, v in cur.execute(q):
        if name.startswith("void at::") or name.startswith("__amd"):
            continue
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if short not in best or grid > best[short]:
            best[short] = grid
    for name, grid, cn, n, v in cur.execute(q):
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if best.get(short) == grid:
            res.setdefault(short, {})[cn] = v
cols = sorted({c for d in res.values() for c in d})
print("kernel," + ",".join(cols))
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(k + "," + ",".join(f"{d.get(c, float('nan')):.0f}" for c in cols))
