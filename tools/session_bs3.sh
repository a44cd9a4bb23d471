#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bluestein" > gpurun_out/bs_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/bs_pytest.log; echo "pytest bluestein rc=$rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/time_odd_lengths.py 251,509,1009,2003 2>&1 | grep linear > gpurun_out/bs_times2.txt; cat gpurun_out/bs_times2.txt
timeout -k 10 300 python tools/stamps_bs.py 1009:float32 251:float32 2003:float32 1009:float64 > gpurun_out/bs_stamps.txt 2>&1; cat gpurun_out/bs_stamps.txt
