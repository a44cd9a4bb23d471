#!/bin/bash
for r in 1 2; do for v in product $VARIANTS; do
  if [ $v = product ]; then lib=spectrograms_amd/libspectro_hip.so; else lib=build/libsgx_$v.so; fi
  echo "$v $(SGX_LIB_PATH=$lib timeout -k 10 200 python tools/bench_istft.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us' % (1e3*d['ms']))")"
done; done
