#!/bin/bash
mkdir -p gpurun_out
python tools/dbg_pack512.py 2>&1 | tail -8
SGX_LIB_PATH=build/libsgx_oddhop.so timeout -k 10 300 python tools/check_oddhop.py > gpurun_out/s10_oddhop.txt 2>&1; echo "oddhop rc=$?"; grep -c " ok on" gpurun_out/s10_oddhop.txt; grep "FAIL" gpurun_out/s10_oddhop.txt | head; tail -16 gpurun_out/s10_oddhop.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s10_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/s10_pytest.log)"; grep -E "^FAILED|^E  " gpurun_out/s10_pytest.log | head -20
