#!/bin/bash
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/bs_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/bs_pytest.log; echo "pytest rc=$rc"
timeout -k 10 300 python tools/time_odd_lengths.py > gpurun_out/bs_times5.txt 2>&1; cat gpurun_out/bs_times5.txt
