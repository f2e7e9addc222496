#!/bin/bash
# Collects everything profiles/r6_* is built from (run on the GPU box from the repo root): bash tools/collect_profiles_r6.sh <tag>
tag=${1:-r6}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 400 python bench.py > $out/bench.log 2> $out/bench.err
echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu > $out/trace.log 2>&1
echo "trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o pmc -- python bench.py --steps 4 --warmup 2 --no-cpu > $out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $out/pmcdf_$c -o pmc -- python tools/time_dense_full.py 64 > $out/pmcdf_$c.log 2>&1
  echo "pmc dense_full $c rc=$?"
  python tools/pmc_any.py dense_track_full $(find $out/pmcdf_$c -name "*.db") > $out/pmcdf_$c.txt 2>&1
  rm -rf $out/pmcdf_$c
done
# the multi-process launch line of the contract at world size 1: RCCL bound at run time, the library issues the collectives (count in schur.weak_scaling / config.collective)
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > $out/torchrun_bench.json 2> $out/torchrun_bench.err
echo "torchrun rc=$?"
python tools/rocpd_summary.py $(find $out/trace -name "*.db") > $out/trace_summary.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_any.py "" $(find $out/pmc_$c -name "*.db") > $out/pmc_$c.txt 2>&1; done
python tools/make_pmc_latest.py $out/pmc_FETCH_SIZE.txt $out/pmc_WRITE_SIZE.txt $out/bench.log $out/pmcdf_FETCH_SIZE.txt $out/pmcdf_WRITE_SIZE.txt $out/pmcdf_FETCH_SIZE.log > $out/pmc_latest.json
rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
# the Schur kernels of SINGLE windows only (no batch of concurrent windows in the trace): what roofline.avg_launch_ms must agree with
timeout 200 rocprofv3 --kernel-trace --stats -d $out/trace_ba -o trace -- python tools/time_ba_kernels.py > $out/trace_ba.log 2>&1
echo "trace single-window BA rc=$?"
python tools/rocpd_summary.py $(find $out/trace_ba -name "*.db") > $out/trace_ba_summary.txt 2>&1
rm -rf $out/trace_ba
# the drop-in SlamGraph::optimize call (set_problem device route): kernels + copies
timeout 200 rocprofv3 --kernel-trace --stats -d $out/trace_dropin -o trace -- python tools/time_dropin.py > $out/trace_dropin.log 2>&1
python tools/rocpd_summary.py $(find $out/trace_dropin -name "*.db") > $out/trace_dropin_summary.txt 2>&1
rm -rf $out/trace_dropin
# block matching alone, per kernel
timeout 200 rocprofv3 --kernel-trace --stats -d $out/trace_stereo -o trace -- python tools/time_stereo.py 512 > $out/stereo.log 2>&1
python tools/rocpd_summary.py $(find $out/trace_stereo -name "*.db") | cut -c1-200 | head -12 > $out/stereo_kernels.txt 2>&1
rm -rf $out/trace_stereo
# grid FAST alone (in the bench step it runs beside the dense tracker), per kernel + its HBM counters
timeout 200 rocprofv3 --kernel-trace --stats -d $out/trace_fast -o trace -- python tools/time_fast.py 512 > $out/fast.log 2>&1
python tools/rocpd_summary.py $(find $out/trace_fast -name "*.db") | cut -c1-200 | grep -E "^\| kernel|^\|---|fast_" > $out/fast_kernels.txt 2>&1
rm -rf $out/trace_fast
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $out/pmcfast -o pmc -- python tools/time_fast.py 512 > /dev/null 2>&1
  python tools/pmc_any.py fast_ $(find $out/pmcfast -name "*.db") >> $out/fast_pmc.txt 2>&1
  rm -rf $out/pmcfast
done
# the accept test of the quarter-grid tracker: what one float sum costs (parallel form / past the caches / sequential chain), and the solve's phase clocks
python tools/time_seqsum.py > $out/seqsum.log 2>&1
SVS_BA_DEBUG=1 python tools/time_ba.py 50 20000 2>&1 | grep -i "solve phases" | tail -3 > $out/solve_phases.log
python tools/time_two_threads_row.py 2>/dev/null | tail -1 > $out/two_threads.json
ls -la $out
