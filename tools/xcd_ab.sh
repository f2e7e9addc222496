set -x
mkdir -p gpurun_out/xcd
export TMPDIR=/tmp
for s in 1 0 1 0; do python tools/time_fast.py 512 xcd_swizzle=$s; done > gpurun_out/xcd/fast_ab.txt 2>&1
for s in 1 0; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/xcd/kt_fast_$s -o kt -- python tools/time_fast.py 512 xcd_swizzle=$s > /dev/null 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/xcd/kt_fast_$s -name "*.db") 2>/dev/null | head -12 > gpurun_out/xcd/kt_fast_$s.txt
  rm -rf gpurun_out/xcd/kt_fast_$s
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d gpurun_out/xcd/p -o pmc -- python tools/time_fast.py 512 xcd_swizzle=$s > /dev/null 2>&1
    python tools/pmc_any.py fast_ $(find gpurun_out/xcd/p -name "*.db") >> gpurun_out/xcd/pmc_fast_$s.txt 2>&1
    rm -rf gpurun_out/xcd/p
  done
done
cat gpurun_out/xcd/*.txt
# stereo
for s in 1 0 1 0; do python tools/time_stereo.py 512 xcd_swizzle=$s 2>&1 | head -3; done > gpurun_out/xcd/stereo_ab.txt 2>&1
cat gpurun_out/xcd/stereo_ab.txt
