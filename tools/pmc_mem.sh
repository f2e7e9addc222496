#!/bin/bash
# TA / TCP / TD counter passes for the front-end kernels (one rocprofv3 --pmc run per group, kernel-trace only): bash tools/pmc_mem.sh <outdir> [kernel-substring]
out=${1:-gpurun_out/pmc_mem}
sub=${2:-match_kernel}
mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_READ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu > $out/g$i.log 2>&1
  echo "group $i rc=$?"
  python tools/pmc_any.py "$sub" $(find $out/g$i -name "*.db") > $out/g$i.txt 2>&1
  rm -rf $out/g$i
done
