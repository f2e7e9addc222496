#!/bin/bash
# SQ counter passes for the front-end kernels of the bench batch (one rocprofv3 --pmc run per group, kernel-trace only): bash tools/pmc_sq_fe.sh <outdir> <kernel-substring>
out=${1:-gpurun_out/pmc_sq_fe}
sub=${2:-match_kernel3}
mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o pmc -- python bench.py --quick > $out/g$i.log 2>&1
  echo "group $i rc=$?"
  python tools/pmc_any.py "$sub" $(find $out/g$i -name "*.db") > $out/g$i.txt 2>&1
  rm -rf $out/g$i
done
cat $out/g*.txt
