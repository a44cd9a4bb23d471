#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python tools/exp_fft2d_chunks.py > gpurun_out/s5_fft2d_chunks.txt 2>&1; echo "chunks rc=$?"; cat gpurun_out/s5_fft2d_chunks.txt
timeout -k 10 300 python tools/time_f64.py > gpurun_out/s5_time_f64.txt 2>&1; echo "f64 rc=$?"; cat gpurun_out/s5_time_f64.txt
timeout -k 10 300 python tools/time_generic.py > gpurun_out/s5_time_generic.txt 2>&1; echo "generic rc=$?"; cat gpurun_out/s5_time_generic.txt
