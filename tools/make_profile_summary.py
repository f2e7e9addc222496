#!/usr/bin/env python3
"""Builds profiles/<round>_summary.md from the artefacts of one gpurun session:
  bench log (the JSON line of `python bench.py`), the rocpd database of
  `rocprofv3 --kernel-trace --stats -- python bench.py ...`, and the two PMC databases
  (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate runs).
usage: make_profile_summary.py <bench.log> <trace.db> <fetch.db> <write.db> <out.md> [title] [notes.md]
(notes.md = hand-written analysis appended as the last section)"""
import io
import json
import sqlite3
import sys
from contextlib import redirect_stdout

import pmc_summary
import rocpd_summary


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def per_geometry(path):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
    gy = "grid_size_y" if "grid_size_y" in cols else ("grid_y" if "grid_y" in cols else None)
    if not gx:
        return []
    q = (f"select {name_col}, {gx}, {gy}, count(*), avg(end-start), min(end-start) from kernels "
         f"group by {name_col}, {gx}, {gy} order by avg(end-start) desc")
    return [(short(n), x, y, c, a / 1e3, m / 1e3) for n, x, y, c, a, m in cur.execute(q)
            if not (n.startswith("void at::") or n.startswith("__amd"))]


def capture(fn, *a):
    buf = io.StringIO()
    with redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


def main(bench_log, trace_db, fetch_db, write_db, out, title="Round 1", notes=None):
    line = [l for l in open(bench_log).read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    fe, sc, rf, cpu = d["frontend"], d["schur"], d["roofline"], d.get("cpu_baseline") or {}
    st = fe["stage_ms_per_batch"]
    B = d["config"]["batch_streams_per_gpu"]
    geo = per_geometry(trace_db)
    kavg = {}
    for n, x, y, c, a, m in geo:
        kavg.setdefault(n, a)
    md = []
    md.append(f"# {title} — measured results on 1 x MI355X (final commit of the round)\n")
    md.append("All numbers from `gpurun` boxes (1 GPU, ROCm 7.2, PyTorch 2.10+rocm7.0).  Raw rocpd databases stay in\n"
              "`gpurun_out/` (scratch); this file is produced by `tools/make_profile_summary.py` from them.\n")
    md.append(f"## 1. bench.py (`python bench.py --steps {d['steps']} --warmup {d['warmup']}`, N=1, batch {B} streams)\n")
    md.append("```json\n" + line + "\n```\n")
    md.append("Reading it:\n"
              f"* **front-end**: {d['value']:.0f} frames/s with {B} independent 640x480 streams per launch ({d['ms_per_step']:.2f} ms per step of\n"
              f"  {B} frames); one stream per launch (drop-in call pattern): {fe['latency_mode_B1']['ms_per_frame']:.3f} ms/frame =\n"
              f"  {fe['latency_mode_B1']['frames_per_s']:.0f} frames/s.  CPU oracle (1 core of the box, same stages): {cpu.get('value', float('nan')):.1f} frames/s.\n"
              f"  Stage times per batch (ms): " + ", ".join(f"{k} {v:.3f}" for k, v in st.items()) + ".\n"
              f"* **stereo block matching** (`calcDisparityCpu`, not part of the headline step): {st.get('stereo_bm', float('nan')):.2f} ms per {B} frames;\n"
              f"  CPU oracle {cpu.get('stereo_bm_ms_per_frame', float('nan'))} ms per frame (naive restatement, not OpenCV's SIMD code).\n"
              f"* **Schur / BA** (50 KF, 20k landmarks, {sc['edges']} edges, `OptParams(2,true,3)`, {sc['lm_trials_per_optimize']:.0f} LM trials):\n"
              f"  {sc['ms_per_optimize']:.3f} ms per `optimize`; CPU oracle {cpu.get('schur_ms_per_optimize', float('nan'))} ms\n"
              f"  => **{sc['speedup_vs_cpu_port']}x** (target >= 30x).  Kernel times per trial: Schur kernel {sc['kernel_ms']['landmark_reduce']*1e3:.0f} us,\n"
              f"  envelope LDL^T solve {sc['kernel_ms']['solve_cholesky']*1e3:.0f} us, back-substitution + trial chi2 {sc['kernel_ms']['backsub_chi2']*1e3:.0f} us.\n"
              f"* **roofline (Schur kernel `ba_landmark_kernel<0>`)**: {rf['alg_bytes_per_launch']/1e6:.2f} MB algorithmic / {rf['avg_launch_ms']*1e3:.1f} us\n"
              f"  = {rf['achieved']:.1f} GB/s = {100*rf['frac']:.2f} % of 8 TB/s; PMC traffic {(rf['traffic'] or 0)/1e6:.1f} MB per launch (section 3).\n"
              "  The kernel is not byte-bound; see section 4.\n")
    md.append("## 2. rocprofv3 --kernel-trace --stats (same build)\n")
    md.append("Command: `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d gpurun_out/<dir> -o <name> -- python bench.py --steps 10 --warmup 3 --no-cpu`\n")
    if geo:
        md.append(f"Per kernel and launch geometry (grid y = {B} rows are the batched front-end, y = 1 the latency-mode launches):\n")
        md.append("| kernel | grid (x,y) | calls | avg us | min us |\n|---|---|---:|---:|---:|")
        for n, x, y, c, a, m in geo:
            md.append(f"| `{n}` | {x},{y} | {c} | {a:.1f} | {m:.1f} |")
        md.append("")
    tr = kavg.get("ba_landmark_kernel<0>")
    if tr:
        md.append(f"Agreement check required by the contract: `ba_landmark_kernel<0>` averages **{tr:.1f} us** in the trace vs\n"
                  f"**{rf['avg_launch_ms']*1e3:.1f} us** from bench.py's HIP events on the ctx stream.\n")
    md.append("Totals:\n")
    md.append(capture(rocpd_summary.main, trace_db))
    md.append("\n## 3. PMC counters (HBM traffic), separate passes\n")
    md.append("Commands: `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu` and the same with\n"
              "`--pmc WRITE_SIZE` (two runs; FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md \"rocprofv3 PMC slots\").\n"
              "Units are KB per launch.  On gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x (same guide), so the\n"
              "corrected column doubles it; for narrow/gather patterns the factor is uncalibrated and the raw value is the lower bound.\n")
    md.append(capture(pmc_summary.main, fetch_db, write_db))
    if notes:
        md.append(open(notes).read())
    open(out, "w").write("\n".join(md) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main(*sys.argv[1:])
