#!/bin/bash
# profiles/r6_* from gpurun_out/r6 (what tools/collect_profiles_r6.sh + the `pytest -m gpu -s` log left there): bash tools/make_r6_summary.sh
set -e
cd "$(dirname "$0")/.."
python tools/make_r2_summary.py gpurun_out/r6 profiles/r6_summary.md "Round 6: bench, kernel trace, PMC"
python - <<'PY'
import json
src = 'gpurun_out/r6/'
md = open('profiles/r6_summary.md').read()
md += "\n## 5. Single-window Schur launches only: `rocprofv3 --kernel-trace --stats -- python tools/time_ba_kernels.py` (12 optimizes of the 50 KF / 20 k window, no concurrent windows)\n\n"
md += "`roofline.avg_launch_ms` of the bench line (HIP events) is the launch of `ba_landmark_kernel<0, 7, 2>` in a single window; the trace of the whole bench (section 2) mixes in the launches of the batched-windows rows (8 / 32 windows at once).  This trace holds the single-window launches alone:\n\n"
md += open(src + 'trace_ba_summary.txt').read()
md += "\n```\n" + "\n".join(l for l in open(src + 'trace_ba.log').read().splitlines() if l.startswith('us per trial')) + "\n```\n"
md += "\n## 6. The drop-in `SlamGraph::optimize` call (set_problem device route + optimize + get_state): `rocprofv3 --kernel-trace --stats -- python tools/time_dropin.py`\n\n"
md += open(src + 'trace_dropin_summary.txt').read()
md += "\n```\n" + "\n".join(l for l in open(src + 'trace_dropin.log').read().splitlines() if l.startswith('host_marshal') or l.startswith('window')) + "\n```\n"
t = json.loads([l for l in open(src + 'torchrun_bench.json').read().splitlines() if l.startswith('{')][-1])
md += "\n## 7. `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1` (world size 1: the library-owned RCCL communicator and the one-shot P2P transport both run their code paths)\n\n"
md += "* RCCL: `" + json.dumps(t['config']['collective']) + "`\n* one-shot P2P transport (`svs_comm_create_p2p`): `" + json.dumps(t['schur']['one_shot_p2p_transport']) + "`\n* value " + str(t['value']) + " frames/s, optimize " + str(t['schur']['ms_per_optimize']) + " ms\n"
md += "\n## 8. `__amd_rocclr_copyBuffer` launches in the traced run (section 2)\n\nThe traced command now contains rows that are host-IO by definition and did not exist in round 3's trace: the two-threads row (600 + 64 latency-mode frames with host images in and records out: 6 copies per frame, about 4 000; 120 + optimizes with state restore, about 700), the latency rows (about 110 host-IO frames, about 700) and the drop-in BA rows (6 - 8 staged pieces per call).  The per-stream setup copies of a 512-stream batch that round 3's verdict counted (about 2 000 per batch object) are gone: `svs_frontend_keep_keyframes` / `svs_frontend_set_candidates_all` are one staged upload per call (2 + 4 copies per batch object).\n"
md += "\n## 9. Block matching alone, per kernel: `rocprofv3 --kernel-trace --stats -- python tools/time_stereo.py 512` (512 pairs of 640 x 480)\n\n" + open(src + 'stereo_kernels.txt').read()
md += "\n```\n" + "\n".join(l for l in open(src + 'stereo.log').read().splitlines() if 'ms per' in l or 'differs' in l) + "\n```\n"
md += "\n## 9b. Grid FAST alone (round 6: no score image -- candidate lists + LDS bitmap compaction): `rocprofv3 --kernel-trace --stats -- python tools/time_fast.py 512`, then one `--pmc` pass each for FETCH_SIZE / WRITE_SIZE (KB per launch of 512 frames)\n\n" + open(src + 'fast_kernels.txt').read()
md += "\n```\n" + "\n".join(l for l in open(src + 'fast.log').read().splitlines() if 'ms per' in l or 'corners' in l) + "\n" + open(src + 'fast_pmc.txt').read() + "```\n"
md += "\n## 10. The tracker's float sums (`tools/time_seqsum.py`) and the solve's phase clocks (`SVS_BA_DEBUG=1 python tools/time_ba.py 50 20000`)\n\n```\n" + open(src + 'seqsum.log').read() + open(src + 'solve_phases.log').read() + "```\n"
md += "\n## 11. Two threads, one GPU (`tools/time_two_threads_row.py`)\n\n```\n" + open(src + 'two_threads.json').read() + "```\n"
open('profiles/r6_summary.md', 'w').write(md)
PY
cp gpurun_out/r6/pmc_latest.json profiles/pmc_latest.json
cp gpurun_out/r6/torchrun_bench.json profiles/r6_torchrun_bench.json
grep -v "^$" gpurun_out/r6/torchrun_bench.err | head -20 > profiles/r6_torchrun_rccl.log
(echo "# pytest tests -m gpu -q -s on the MI355X box (round 6, HEAD; prints = measured deviations the bars are set from)"; grep -v "amdgpu.ids\|^$\|Gloo\|c10d\|socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r6/gpu_tests_full2.log | cut -c1-700) > profiles/r6_gpu_tests.txt
