#!/bin/bash
# Times prebuilt library variants (python -m spectrograms_amd.build --variant NAME flags...): tools/abv.sh "base nostore ..." "workloads" [reps]
# "product" = the in-tree library.  Variants are interleaved and the whole list is repeated `reps` times (box clocks drift).
reps=${3:-2}
for wl in ${2:-linear_power}; do for r in $(seq $reps); do for n in $1; do
  if [ "$n" = product ]; then lib=spectrograms_amd/libspectro_hip.so; else lib=build/libsgx_$n.so; fi
  line=$(env SGX_LIB_PATH=$lib python bench.py --workload $wl --no-cpu-baseline --no-legs --steps 200 --warmup 20 2>/dev/null | tail -1)
  echo "$n $wl $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('kernel_us=%.1f'%(1e3*d['roofline']['kernel_ms']))" "$line")"
done; done; done
