#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite (--kernel-trace [--stats]) into a per-kernel table
(calls, total/avg/min/max duration in us, VGPRs, LDS) -- the `profiles/*.md` files come from this."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    q = f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by sum(end-start) desc"
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for n, c, s, a, mn, mx in rows:
        n = n.replace("|", "\\|")
        if len(n) > 110:
            n = n[:107] + "..."
        lines.append(f"| `{n}` | {c} | {s/1e3:.1f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.1f} |")
    txt = "\n".join(lines)
    if out:
        open(out, "a").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
