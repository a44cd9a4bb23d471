#!/bin/bash
for r in 1 2; do for v in product stagef64; do
  if [ $v = product ]; then lib=spectrograms_amd/libspectro_hip.so; else lib=build/libsgx_$v.so; fi
  echo "-- $v"; B=256 ITERS=20 SGX_LIB_PATH=$lib timeout -k 10 200 python tools/time_generic.py 2>&1 | grep "float64.*linear"
done; done > gpurun_out/f64_staged.txt 2>&1
cat gpurun_out/f64_staged.txt
SGX_LIB_PATH=build/libsgx_stagef64.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "float64 and (pow2 or mixed or sizes)" 2>&1 | tail -3
