#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_istft.py -x -q -m gpu > gpurun_out/istft_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/istft_pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit 1
for i in 1 2 3; do timeout -k 10 200 python tools/bench_istft.py 2>&1 | tail -1; done
HOP=128 timeout -k 10 200 python tools/bench_istft.py 2>&1 | tail -1
HOP=512 timeout -k 10 200 python tools/bench_istft.py 2>&1 | tail -1
B=2048 timeout -k 10 200 python tools/bench_istft.py 2>&1 | tail -1
