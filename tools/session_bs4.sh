#!/bin/bash
for lib in spectrograms_amd/libspectro_hip.so build/libsgx_bscost025.so; do
  echo "-- $lib"
  SGX_LIB_PATH=$lib timeout -k 10 300 python tools/time_odd_lengths.py 11,17,18,22,23,26,29,34,38,46,50,58,100,118,441,1023 float32,float64 2>&1 | grep -v amdgpu.ids
done > gpurun_out/bs_cost2.txt 2>&1
cat gpurun_out/bs_cost2.txt
export SGX_PROF_NFFT=1009 SGX_PROF_HOP=252; bash tools/profile.sh bs1009 linear_power > gpurun_out/bs_prof.log 2>&1; python tools/summarize_prof.py gpurun_out/prof_bs1009_linear_power > gpurun_out/prof_bs1009_linear_power/summary.txt 2>&1; tail -60 gpurun_out/prof_bs1009_linear_power/summary.txt
