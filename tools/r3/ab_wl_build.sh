#!/bin/bash
# CPU box: a build of libro_stft.so with the round-3 experiment tools/r3/ro_stft_wl.hip (stft32k_kernel's structure at
# N = 16384 / 8192) routed in for those sizes' magnitude rows.  usage: ab_wl_build.sh NAME [-DRO_WL_FUSE16=0 ...]
R=/root/repo
NAME=$1; shift
AB_EXTRA_SOURCES="-I$R/radio-observer_amd/csrc $R/tools/r3/ro_stft_wl.hip" $R/tools/ab_build.sh $NAME -DRO_DIAG=1 -DRO_USE_WL=1 "$@"
