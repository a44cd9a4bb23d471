#!/bin/bash
for r in 1 2; do for lib in spectrograms_amd/libspectro_hip.so build/libsgx_bslds36.so build/libsgx_bslds18.so; do
  echo "-- $lib"
  SGX_LIB_PATH=$lib timeout -k 10 300 python tools/time_odd_lengths.py 127,251,509,1009,2003 float32,float64 2>&1 | grep -v amdgpu.ids | grep linear
done; done > gpurun_out/bs_lds.txt 2>&1
cat gpurun_out/bs_lds.txt
