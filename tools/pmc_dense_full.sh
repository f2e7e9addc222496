#!/bin/bash
# counter passes for the full-resolution tracker kernel (one rocprofv3 --pmc run per group, kernel-trace only)
# usage (GPU box, repo root): bash tools/pmc_dense_full.sh <outdir> <batch>
out=${1:-gpurun_out/pmc_df}
B=${2:-256}
mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o pmc -- python tools/time_dense_full.py $B > $out/g$i.log 2>&1
  echo "group $i rc=$?"
done
for d in $out/g*/; do python tools/pmc_any.py dense_track_full $(find $d -name "*.db"); done > $out/summary.txt 2>&1
cat $out/summary.txt
