#!/bin/bash
# A/B timing of the shape-generic kernels under different build flags / environment (run on the GPU box).
# Usage: bash tools/ablate_generic.sh "<hipcc flags>|<env assignments>" ...   e.g. "-DSGX_RRW32=4|SGX_REG_LDS_KB=40"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
i=0
for spec in "$@"; do
  flags=${spec%%|*}; envs=${spec#*|}
  lib=/tmp/libsgx_ablg_$i.so
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $flags -Iinclude -Ispectrograms_amd/csrc -o $lib \
     spectrograms_amd/csrc/plan.hip spectrograms_amd/csrc/fft2d.hip spectrograms_amd/csrc/kernels_generic.hip spectrograms_amd/csrc/kernels_r32x16.hip spectrograms_amd/csrc/kernels_fft2d.hip spectrograms_amd/csrc/kernels_c2c1024.hip spectrograms_amd/csrc/kernels_reg2d.hip 2>/dev/null || { echo "build failed: $flags"; continue; }
  echo "=== flags='$flags' env='$envs'"
  env $envs SGX_LIB_PATH=$lib B=${B:-256} python tools/time_generic.py 2>&1 | grep -v amdgpu.ids
  i=$((i+1))
done
