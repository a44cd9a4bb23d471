#!/bin/bash
for r in 1 2 3; do for v in nochunk product chunk32 chunk128; do
  if [ $v = product ]; then lib=spectrograms_amd/libspectro_hip.so; else lib=build/libsgx_$v.so; fi
  echo "$v $(SGX_LIB_PATH=$lib timeout -k 10 120 python tools/bench_fft2d.py 512 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fft2d %.3f convolve %.3f ms' % (d['fft2d']['ms_per_batch'], d['convolve_fft']['ms_per_batch']))")"
done; done > gpurun_out/conv_ab.txt 2>&1
cat gpurun_out/conv_ab.txt
