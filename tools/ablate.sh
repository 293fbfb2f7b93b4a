#!/bin/bash
# Build diagnostic variants of libro_stft.so (one phase removed each) here on the CPU box;
# run them on the GPU with tools/ablate_run.sh.  Outputs gpurun_out-independent files in build/ablate/.
set -e
R=/root/repo
mkdir -p $R/build/ablate
for A in ${@:-0 1 2 3 4 8 16 32}; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -shared -DRO_ABLATE=$A $EXTRA \
     -o $R/build/ablate/libro_stft_a$A.so $R/radio-observer_amd/csrc/ro_kernels.hip $R/radio-observer_amd/csrc/ro_stft_capi.cpp &
done
wait
ls -la $R/build/ablate/
