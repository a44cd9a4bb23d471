#!/bin/bash
for v in nobsc2c product; do
  if [ $v = product ]; then lib=spectrograms_amd/libspectro_hip.so; else lib=build/libsgx_$v.so; fi
  echo "-- $v"
  SGX_LIB_PATH=$lib SHAPES=1023x1023,509x509,251x1024,1024x251,127x127,2003x2003,1000x1009 timeout -k 10 400 python tools/sweep2d.py 2>&1 | grep -v amdgpu
  for nf in 251 1009 1023; do
    SGX_LIB_PATH=$lib B=64 N_FFT=$nf timeout -k 10 200 python tools/bench_istft.py 2>&1 | tail -1
    SGX_LIB_PATH=$lib B=64 N_FFT=$nf DTYPE=float64 timeout -k 10 200 python tools/bench_istft.py 2>&1 | tail -1
  done
done > gpurun_out/bs_c2c_ab.txt 2>&1
cat gpurun_out/bs_c2c_ab.txt
