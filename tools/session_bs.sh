#!/bin/bash
# fused chirp-z kernel + chunked convolve: parity first, then timings
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bluestein or non_pow2 or fuzz or long_composite" > gpurun_out/bs_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/bs_pytest.log; echo "pytest bluestein rc=$rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_fft2d.py -x -q -m gpu > gpurun_out/bs_pytest2.log 2>&1; rc=$?
tail -5 gpurun_out/bs_pytest2.log; echo "pytest fft2d rc=$rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/time_odd_lengths.py 251,509,1006,1009,1024,2003,2048,4093 > gpurun_out/bs_times.txt 2>&1 && cat gpurun_out/bs_times.txt
echo "== SGX_BS_COST=1 (chirp-z for shorter primes too) vs product"
for lib in spectrograms_amd/libspectro_hip.so build/libsgx_bscost1.so; do
  echo "-- $lib"
  SGX_LIB_PATH=$lib timeout -k 10 300 python tools/time_odd_lengths.py 31,61,67,97,127,128,199,211,251,254 float32,float64 2>&1 | grep -v amdgpu.ids
done > gpurun_out/bs_cost.txt 2>&1
cat gpurun_out/bs_cost.txt
timeout -k 10 300 python tools/bench_fft2d.py > gpurun_out/bs_fft2d.txt 2>&1; tail -8 gpurun_out/bs_fft2d.txt
