#!/bin/bash
# rocprofv3 kernel-trace stats of the SAME command the driver runs (bench.py), per workload; run on the GPU box from the repo
# root.  Writes gpurun_out/bench_prof_<workload>/ and gpurun_out/bench_<workload>.json.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
for wl in "$@"; do
  cd $ROOT
  timeout 600 python3 bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_prof_$wl -- python3 $ROOT/bench.py --workload $wl --no-cpu-baseline > $OUT/bench_prof_$wl.log 2>&1
  f=$(find $OUT/bench_prof_$wl -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/bench_${wl}_kernel_stats.csv
  tail -1 $OUT/bench_prof_$wl.log | head -c 600; echo
done
