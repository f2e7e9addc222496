#!/usr/bin/env python3
"""profiles/<tag>_summary.md from the text artefacts tools/collect_profiles_r2.sh leaves in gpurun_out/<tag>/ (the rocpd databases are
deleted on the GPU box: tens of MB each).  usage: make_r2_summary.py gpurun_out/<tag> profiles/<name>_summary.md "<title>" [notes.md]"""
import json
import os
import sys


def main(src, out, title, notes=None):
    rd = lambda n: open(os.path.join(src, n)).read()
    line = [l for l in rd("bench.log").splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    md = [f"# {title}\n",
          "1 x MI355X `gpurun` box (ROCm 7.2, PyTorch 2.10+rocm7.0).  Produced by `tools/collect_profiles_r<N>.sh` + `tools/make_r2_summary.py`;\n"
          "the rocpd databases stay on the box, the tables below are their per-kernel aggregates.\n",
          "## 1. `python bench.py` (N = 1)\n", "```json\n" + line + "\n```\n"]
    rf, sc = d["roofline"], d["schur"]
    md.append(f"* headline: {d['value']:.0f} frames/s ({d['ms_per_step']:.3f} ms per step of {d['config']['batch_streams_per_gpu']} streams); "
              f"`optimize` {sc['ms_per_optimize']:.3f} ms, kernels (HIP events) {sc['kernel_ms']}\n"
              f"* roofline kernel `{rf['kernel']}`: {rf['avg_launch_ms'] * 1e3:.1f} us per launch, {rf['alg_bytes_per_launch']} algorithmic bytes, "
              f"{rf['achieved']:.1f} GB/s = {rf['frac']:.4f} of {rf['peak']:.0f} GB/s; traffic {rf.get('traffic')}\n")
    if "dense_full" in d:
        md.append(f"* dense_full (BASELINE config 5): `{json.dumps(d['dense_full'])}`\n")
    md.append("## 2. `rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu`: per-kernel durations\n")
    md.append(rd("trace_summary.txt"))
    md.append("\n## 3. PMC, separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`; KB per launch as read)\n")
    md.append("```\n" + rd("pmc_FETCH_SIZE.txt") + "```\n```\n" + rd("pmc_WRITE_SIZE.txt") + "```\n")
    md.append("Full-resolution tracker alone (`tools/time_dense_full.py 64`):\n```\n" + rd("pmcdf_FETCH_SIZE.txt") + rd("pmcdf_WRITE_SIZE.txt") + "```\n")
    md.append("## 4. `profiles/pmc_latest.json` (what `bench.py` reports as `roofline.traffic`)\n```json\n" + rd("pmc_latest.json") + "```\n")
    if notes and os.path.exists(notes):
        md.append("## 5. Notes\n" + open(notes).read())
    open(out, "w").write("\n".join(md))


if __name__ == "__main__":
    main(*sys.argv[1:])
