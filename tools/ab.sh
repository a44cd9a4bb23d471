#!/bin/bash
# A/B of an environment switch on the bench workloads: tools/ab.sh VAR "v1 v2" "workloads"
# e.g. tools/ab.sh SGX_WIDE "0 1" "linear_power stft" — prints ms_per_step and kernel_ms per (value, workload), two passes each
VAR=$1; VALS=${2:-"0 1"}; WLS=${3:-"linear_power stft"}
for pass in 1 2; do
for wl in $WLS; do for v in $VALS; do
  line=$(env $VAR=$v python bench.py --workload $wl --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1)
  echo "$VAR=$v $wl $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('ms_per_step=%.4f kernel_ms=%.4f frac=%.3f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))" "$line")"
done; done; done
