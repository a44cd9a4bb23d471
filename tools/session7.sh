#!/bin/bash
# GPU session 7: remaining GPU tests after the chirp-z test fix, the driver's bench command, rocprofv3 summaries of the three headline workloads
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bluestein or fuzz or packed or config" > gpurun_out/s7_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/s7_pytest.log)"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/s7_bench_default.json 2> gpurun_out/s7_bench_default.err; echo "bench rc=$?"
for wl in linear_power mel_power mel_db; do timeout -k 10 600 bash tools/profile.sh r03 $wl > gpurun_out/s7_profile_$wl.log 2>&1; echo "profile $wl rc=$?"; done
cat gpurun_out/traffic_latest.json
grep -h "steady state\|kernel stats" -A1 gpurun_out/prof_r03_*/summary.txt | head -30
