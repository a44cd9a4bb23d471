#!/bin/bash
# rocprofv3 kernel-trace stats for the "(f) next" rows (2-D path, frequency mappings, inverse STFT), on the GPU box from the
# repo root.  Writes gpurun_out/next_*; copy what should be judged into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 300 python3 tools/bench_fft2d.py > $OUT/next_fft2d.json 2> $OUT/next_fft2d.err
timeout 200 python3 tools/time_mappings.py > $OUT/next_mappings.txt 2> $OUT/next_mappings.err
timeout 200 python3 tools/bench_istft.py > $OUT/next_istft.json 2> $OUT/next_istft.err
cd /tmp && export TMPDIR=/tmp
for t in bench_fft2d time_mappings bench_istft; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/next_prof_$t -- python3 $ROOT/tools/$t.py > $OUT/next_prof_$t.log 2>&1
  f=$(find $OUT/next_prof_$t -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/next_${t}_kernel_stats.csv
done
cd $ROOT
cat $OUT/next_fft2d.json $OUT/next_mappings.txt $OUT/next_istft.json
