#!/bin/bash
# Collects everything profiles/<tag>_summary.md is built from (run on the GPU box from the repo root):
#   bash tools/collect_profiles.sh <tag>
# then, back in the dev container:  python tools/make_profile_summary.py gpurun_out/<tag>/bench.log gpurun_out/<tag>/trace/*.db \
#        gpurun_out/<tag>/pmc_FETCH_SIZE/*.db gpurun_out/<tag>/pmc_WRITE_SIZE/*.db profiles/<tag>_summary.md
tag=${1:-r1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 400 python bench.py > $out/bench.log 2> $out/bench.err
echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python bench.py --steps 10 --warmup 3 --no-cpu > $out/trace.log 2>&1
echo "trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu > $out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
find $out -name "*.db" | xargs ls -la
