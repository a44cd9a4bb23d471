#!/bin/bash
# Generic inverse STFT and 2-D path timings (register-tiled k_c2r_reg / k_c2c_reg): the table of profiles/bench_rNN_generic_inverse_2d.txt
for cfg in "512 128 float32" "400 160 float32" "2048 512 float32" "1024 256 float64" "400 160 float64"; do
  set -- $cfg
  N_FFT=$1 HOP=$2 DTYPE=$3 python tools/bench_istft.py 2>/dev/null | tail -1
done
for cfg in "2048 512" "128 2048" "8192 256" "512 1024"; do
  set -- $cfg
  SIDE=$2 python tools/bench_fft2d.py $1 2>/dev/null | tail -1
done
