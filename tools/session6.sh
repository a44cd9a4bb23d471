#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s6_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/s6_pytest.log)"; grep -E "FAILED|Error|assert" gpurun_out/s6_pytest.log | head -20
timeout -k 10 300 python tools/time_odd_lengths.py > gpurun_out/s6_odd_lengths.txt 2>&1; echo "odd rc=$?"; cat gpurun_out/s6_odd_lengths.txt
