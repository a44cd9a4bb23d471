#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s8_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/s8_pytest.log)"; grep -E "^FAILED|^E  " gpurun_out/s8_pytest.log | head -20
timeout -k 10 300 python tools/sweep_short.py > gpurun_out/s8_sweep_short.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/s8_sweep_short.txt
