#!/bin/bash
for r in 1 2; do for v in product rrstag2 rrstag8; do
  if [ $v = product ]; then lib=spectrograms_amd/libspectro_hip.so; else lib=build/libsgx_$v.so; fi
  echo "-- $v"; B=256 ITERS=20 SGX_LIB_PATH=$lib timeout -k 10 200 python tools/time_generic.py 2>&1 | grep linear
done; done > gpurun_out/f64_stagger.txt 2>&1
cat gpurun_out/f64_stagger.txt
