#!/bin/bash
# Times prebuilt library variants (python -m spectrograms_amd.build --variant NAME flags...): tools/abv.sh "base nostore ..." "workloads"
# "product" = the in-tree library.
for wl in ${2:-linear_power}; do for n in $1; do
  if [ "$n" = product ]; then lib=spectrograms_amd/libspectro_hip.so; else lib=build/libsgx_$n.so; fi
  line=$(env SGX_LIB_PATH=$lib python bench.py --workload $wl --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1)
  echo "$n $wl $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('kernel_us=%.1f ms_per_step=%.4f'%(1e3*d['roofline']['kernel_ms'], d['ms_per_step']))" "$line")"
done; done
