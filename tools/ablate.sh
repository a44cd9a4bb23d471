#!/bin/bash
# A/B timing of kernel variants built with -DSGX_ABL_* (timing experiments; outputs of ablated builds are wrong).
# Usage on the GPU box: bash tools/ablate.sh "<flags1>" "<flags2>" ...   e.g. bash tools/ablate.sh "" "-DSGX_ABL_NOSTORE"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
i=0
for flags in "$@"; do
  lib=/tmp/libsgx_abl_$i.so
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $flags -Iinclude -Ispectrograms_amd/csrc -o $lib \
     spectrograms_amd/csrc/plan.hip spectrograms_amd/csrc/fft2d.hip spectrograms_amd/csrc/kernels_generic.hip spectrograms_amd/csrc/kernels_r32x16.hip spectrograms_amd/csrc/kernels_fft2d.hip spectrograms_amd/csrc/kernels_c2c1024.hip spectrograms_amd/csrc/kernels_reg2d.hip spectrograms_amd/csrc/kernels_q16x32.hip 2>/dev/null || { echo "build failed: $flags"; continue; }
  for wl in ${WORKLOADS:-linear_power}; do
    SGX_LIB_PATH=$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload $wl 2>&1 | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %-12s kernel_ms=%.4f  Mframes/s=%.1f' % ('$flags', '$wl', d['roofline']['kernel_ms'], d['value']/1e6))"
  done
  i=$((i+1))
done
