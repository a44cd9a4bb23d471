#!/bin/bash
# GPU session 1 of round 3: full GPU tests, the default bench line, variant gates + A/B timings, phase stamps.
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/s1_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/s1_pytest.log)"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/s1_bench_default.json 2> gpurun_out/s1_bench_default.err; echo "bench rc=$?"
SGX_LIB_PATH=spectrograms_amd/libspectro_hip.so timeout -k 10 200 python tools/ab_check.py product > gpurun_out/s1_ab_product.log 2>&1; echo "ab product rc=$?"
for v in bandpf bfly3 both; do
  SGX_LIB_PATH=build/libsgx_$v.so timeout -k 10 200 python tools/ab_check.py $v > gpurun_out/s1_ab_$v.log 2>&1; echo "ab $v rc=$?"
  python tools/ab_check.py --diff product $v >> gpurun_out/s1_ab_$v.log 2>&1
done
timeout -k 10 600 bash tools/abv.sh "product bandpf bfly3 both" "mel_power linear_power" 3 > gpurun_out/s1_abv.txt 2>&1; echo "abv rc=$?"
SGX_STAMPS_LIB=build/libsgx_stamps.so timeout -k 10 120 python tools/stamps.py mel_power > gpurun_out/s1_stamps_mel.txt 2>&1
SGX_STAMPS_LIB=build/libsgx_stampspf.so timeout -k 10 120 python tools/stamps.py mel_power > gpurun_out/s1_stamps_mel_pf.txt 2>&1
cat gpurun_out/s1_abv.txt
