#!/bin/bash
# kernel + memcpy durations of the drop-in SlamGraph::optimize call (svs_ba_set_problem device route): bash tools/trace_dropin.sh  (GPU box, repo root)
export TMPDIR=/tmp
out=gpurun_out/dropin
mkdir -p $out
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/trace -o trace -- python tools/time_dropin.py > $out/trace.log 2>&1
echo "trace rc=$?"
python tools/rocpd_summary.py $(find $out/trace -name "*.db") > $out/trace_summary.txt 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/dropin/trace/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
mc = [t for t in tabs if "memory_copy" in t and "rocpd" in t]
print("tables", mc)
for t in mc:
    cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
    print(t, cols)
    try:
        rows = c.execute(f"select size, (end - start) from {t} order by start").fetchall()
        big = [(s, d) for s, d in rows if s > 500000]
        print("copies > 0.5 MB:", len(big), "GB/s", [round(s / d, 2) for s, d in big[-12:]], "us", [round(d / 1e3, 1) for s, d in big[-12:]])
    except Exception as e:
        print("query failed", e)
PY
rm -rf $out/trace
