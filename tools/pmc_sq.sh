#!/bin/bash
# SQ counter passes for the BA kernels (one rocprofv3 --pmc run per group, kernel-trace only).
# usage (on the GPU box, from the repo root): bash tools/pmc_sq.sh <outdir>
out=${1:-gpurun_out/pmc_sq}
mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_VALU_MFMA_F64"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu --batch 4 > $out/g$i.log 2>&1
  echo "group $i rc=$?"
done
