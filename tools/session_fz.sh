#!/bin/bash
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_fft2d.py -x -q -m gpu -k "fuzz" > gpurun_out/fz_pytest.log 2>&1; rc=$?
tail -12 gpurun_out/fz_pytest.log; echo "pytest rc=$rc"
