#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE / WRITE_SIZE (KB per launch, largest grid of each kernel) from two rocprofv3
--pmc passes (separate runs, as MI355X_MICROARCH.md prescribes).  Prints a markdown table with the
gfx950 correction for wide coalesced reads (FETCH_SIZE x2) next to the raw numbers."""
import sqlite3
import sys


def per_kernel(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    out = {}
    q = ("select kernel_name, grid_size, count(*), avg(value) from counters_collection group by kernel_name, grid_size "
         "order by kernel_name, grid_size")
    for name, grid, n, v in cur.execute(q):
        if name not in out or grid > out[name][0]:
            out[name] = (grid, n, v)
    return out


def main(fetch_db, write_db):
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    print("| kernel (largest grid) | launches | FETCH_SIZE KB (raw) | x2 corrected KB | WRITE_SIZE KB | traffic KB (2F+W) |")
    print("|---|---:|---:|---:|---:|---:|")
    for name in sorted(f, key=lambda k: -(f[k][2] * 2 + w.get(k, (0, 0, 0))[2])):
        if name.startswith("void at::") or name.startswith("__amd"):
            continue
        fv = f[name][2]
        wv = w.get(name, (0, 0, 0.0))[2]
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        print(f"| `{short}` | {f[name][1]} | {fv:.1f} | {2*fv:.1f} | {wv:.1f} | {2*fv+wv:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
