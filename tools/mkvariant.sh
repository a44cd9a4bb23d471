#!/bin/bash
# Build a variant of the library in build/ (travels to the GPU box): tools/mkvariant.sh NAME [flags...]
# Only kernels_r32x16.hip and kernels_q16x32.hip are rebuilt with the flags; the other translation units are compiled once into build/obj.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); cd $ROOT
NAME=$1; shift
CS=spectrograms_amd/csrc
CF="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -I$CS"
mkdir -p build/obj
for f in plan fft2d kernels_generic kernels_fft2d kernels_c2c1024 kernels_reg2d; do
  if [ ! -f build/obj/$f.o ] || [ $CS/$f.hip -nt build/obj/$f.o ] || [ $CS/sgx_internal.h -nt build/obj/$f.o ]; then
    hipcc $CF -c $CS/$f.hip -o build/obj/$f.o &
  fi
done
hipcc $CF "$@" -c $CS/kernels_r32x16.hip -o build/obj/r32x16_$NAME.o &
hipcc $CF "$@" -c $CS/kernels_q16x32.hip -o build/obj/q16x32_$NAME.o
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsgx_$NAME.so build/obj/{plan,fft2d,kernels_generic,kernels_fft2d,kernels_c2c1024,kernels_reg2d}.o build/obj/r32x16_$NAME.o build/obj/q16x32_$NAME.o
echo build/libsgx_$NAME.so
