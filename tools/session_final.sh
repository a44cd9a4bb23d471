#!/bin/bash
# Round-end GPU session: all GPU tests, smoke, the driver's bench command, rocprofv3 summaries (+ traffic_latest.json) of the three headline workloads, inverse STFT / 2-D benches
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/sf_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/sf_pytest.log)"
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/sf_bench_default.json 2> gpurun_out/sf_bench_default.err; echo "bench rc=$?"
timeout -k 10 200 python tools/bench_istft.py > gpurun_out/sf_istft.json 2>&1; tail -1 gpurun_out/sf_istft.json
for wl in linear_power mel_power mel_db; do timeout -k 10 600 bash tools/profile.sh r03 $wl > gpurun_out/sf_profile_$wl.log 2>&1; echo "profile $wl rc=$?"; done
cat gpurun_out/traffic_latest.json
grep -h "steady state" gpurun_out/prof_r03_*/summary.txt
