#!/bin/bash
# GPU session 2 of round 3: full GPU tests with the 3-instruction butterflies in every kernel + the multi-rank C test; the LDS-DMA staging variant
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s2_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/s2_pytest.log)"
SGX_LIB_PATH=spectrograms_amd/libspectro_hip.so timeout -k 10 200 python tools/ab_check.py product > gpurun_out/s2_ab_product.log 2>&1; echo "ab product rc=$?"
SGX_LIB_PATH=build/libsgx_dma.so timeout -k 10 200 python tools/ab_check.py dma > gpurun_out/s2_ab_dma.log 2>&1; echo "ab dma rc=$?"
python tools/ab_check.py --diff product dma >> gpurun_out/s2_ab_dma.log 2>&1
tail -12 gpurun_out/s2_ab_dma.log
timeout -k 10 500 bash tools/abv.sh "product dma" "mel_power mel_db" 3 > gpurun_out/s2_abv.txt 2>&1; echo "abv rc=$?"
SGX_STAMPS_LIB=build/libsgx_stampsdma.so timeout -k 10 120 python tools/stamps.py mel_power > gpurun_out/s2_stamps_mel_dma.txt 2>&1
cat gpurun_out/s2_abv.txt
python - <<'PY'
import ctypes as C, sys
sys.path.insert(0, '.')
from spectrograms_amd import _ffi
L = _ffi.lib()
for mode, name in ((0, 'copy'), (1, 'read'), (2, 'write')):
    g = C.c_double(); st = L.sgx_membench(0, 0, mode, 5, C.byref(g)); print('membench', name, st, round(g.value, 1))
PY
