#!/usr/bin/env python3
"""Kernel timeline of the last steps of a traced run: python tools/timeline.py <rocpd.db> [n_last_kernels] -- start / end (us, relative), duration, queue, name"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = list(cur.execute(f"select {name_col}, start, end{', ' + qcol if qcol else ''} from kernels order by start"))
rows = rows[-n:]
t0 = rows[0][1]
print("columns:", cols)
for r in rows:
    nm = r[0]
    for pre in ("void (anonymous namespace)::", "(anonymous namespace)::", "void "):
        if nm.startswith(pre):
            nm = nm[len(pre):]
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  q{r[3] if qcol else '?'}  {nm[:70]}")
