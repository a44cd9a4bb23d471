#!/bin/bash
# Shape-generic kernels: timing table (tools/time_generic.py, 256 x 10 s) + rocprofv3 kernel stats for n_fft 400 and 512.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
B=256 python tools/time_generic.py 2>&1 | grep -v amdgpu.ids > $OUT/generic_times.txt
cd /tmp && export TMPDIR=/tmp
for nf in 400 512; do
  export SGX_PROF_NFFT=$nf SGX_PROF_HOP=$((nf * 2 / 5))
  [ $nf = 512 ] && export SGX_PROF_HOP=128
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gen_$nf -- python3 $ROOT/tools/prof_driver.py linear_power 6 > $OUT/gen_$nf.log 2>&1
  f=$(find $OUT/gen_$nf -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/generic_${nf}_kernel_stats.csv
done
cat $OUT/generic_times.txt
