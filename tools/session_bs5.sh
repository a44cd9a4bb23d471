#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_abi.py tests/test_fft2d.py -x -q -m gpu -k "bluestein or non_pow2 or fuzz or long_composite or kernel or fft2d or helpers or irfft" > gpurun_out/bs_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/bs_pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/time_odd_lengths.py > gpurun_out/bs_times3.txt 2>&1; cat gpurun_out/bs_times3.txt
