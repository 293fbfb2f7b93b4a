#!/bin/bash
# Build a named variant of libro_stft.so into build/ab/ (CPU box).  usage: ab_build.sh NAME [extra hipcc flags...]
set -e
R=/root/repo
NAME=$1; shift
mkdir -p $R/build/ab
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -shared "$@" \
   -o $R/build/ab/libro_stft_$NAME.so $R/radio-observer_amd/csrc/ro_kernels.hip $R/radio-observer_amd/csrc/ro_stft32k.hip $R/radio-observer_amd/csrc/ro_fourstep.hip $R/radio-observer_amd/csrc/ro_f64fused.hip $R/radio-observer_amd/csrc/ro_f64reg.hip $R/radio-observer_amd/csrc/ro_stft_capi.cpp $R/radio-observer_amd/csrc/ro_abi_helpers.cpp $R/radio-observer_amd/csrc/ro_exchange.cpp $R/radio-observer_amd/csrc/ro_stream.cpp $R/radio-observer_amd/csrc/ro_czt.cpp $AB_EXTRA_SOURCES
echo built build/ab/libro_stft_$NAME.so
