#!/bin/bash
# GPU session 3: packed tiles (tests + the short-signal rows of tools/sweep_misc.py), static wave priority A/B
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed or config or ragged or short" > gpurun_out/s3_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/s3_pytest.log)"
timeout -k 10 300 python tools/sweep_short.py > gpurun_out/s3_sweep_short.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/s3_sweep_short.txt
timeout -k 10 500 bash tools/abv.sh "product prio1 prio2" "mel_power linear_power" 2 > gpurun_out/s3_abv.txt 2>&1; cat gpurun_out/s3_abv.txt
