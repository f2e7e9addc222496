#!/bin/bash
# SQ counter passes for the front-end kernels at batch 256 (one rocprofv3 --pmc run per group, kernel-trace only).
out=${1:-gpurun_out/pmc_fe}
mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu > $out/g$i.log 2>&1
  echo "group $i rc=$?"
done
