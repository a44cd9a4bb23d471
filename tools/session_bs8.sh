#!/bin/bash
timeout -k 10 1100 python -m pytest tests/test_fft2d.py tests/test_istft.py -x -q -m gpu > gpurun_out/bs_pytest.log 2>&1; rc=$?
tail -8 gpurun_out/bs_pytest.log; echo "pytest rc=$rc"
