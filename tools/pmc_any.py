#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 --pmc result databases (rocpd sqlite).
usage: pmc_any.py <kernel-substring> db1 [db2 ...]"""
import sqlite3
import sys


def main(sub, paths):
    for p in paths:
        cur = sqlite3.connect(p).cursor()
        q = ("select kernel_name, counter_name, count(*), avg(value), max(value) from counters_collection "
             "where kernel_name like ? group by kernel_name, counter_name order by kernel_name, counter_name")
        for name, cn, n, v, vmax in cur.execute(q, (f"%{sub}%",)):
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            print(f"{short:40s} {cn:28s} n={n:4d} avg={v:14.1f} max={vmax:14.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
