#!/bin/bash
# SQ / LDS counter passes for the BA kernels at 50 KF / 20k (one rocprofv3 --pmc run per group, kernel-trace only)
out=${1:-gpurun_out/pmc_ba}
mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o pmc -- python tools/time_ba.py 50 20000 > $out/g$i.log 2>&1
  echo "group $i rc=$?"
done
for d in $out/g*/; do python tools/pmc_any.py ba_landmark_kernel $(find $d -name "*.db"); done > $out/summary.txt 2>&1
grep "ba_landmark_kernel<0" $out/summary.txt
