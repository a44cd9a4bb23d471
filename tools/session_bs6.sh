#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bluestein or non_pow2 or fuzz or long_composite" > gpurun_out/bs_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/bs_pytest.log; echo "pytest rc=$rc"
timeout -k 10 300 python tools/time_odd_lengths.py 2003,3000,4093,4096,5003,6000,8191,8192 > gpurun_out/bs_times4.txt 2>&1; cat gpurun_out/bs_times4.txt
