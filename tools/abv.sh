#!/bin/bash
# Times prebuilt library variants (tools/mkvariant.sh): tools/abv.sh "base nostore ..." "workloads" [extra env]
for wl in ${2:-linear_power}; do for n in $1; do
  line=$(env SGX_LIB_PATH=build/libsgx_$n.so $3 python bench.py --workload $wl --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1)
  echo "$n $3 $wl $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('kernel_us=%.1f'%(1e3*d['roofline']['kernel_ms']))" "$line")"
done; done
