#!/bin/bash
# CPU box: AddressSanitizer + UBSan builds of the oracle and of the host-side C++ mirror, and the CPU tests that
# exercise them (ring, frontends, recorders, FITS/CSV writers, oracle).  GPU sanitizers are not available on the pool.
set -e
R=/root/repo
mkdir -p $R/build/asan
gcc -O1 -g -fPIC -std=c11 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-fast-math -ffp-contract=off \
    -shared -o $R/build/asan/libro_oracle.so $R/oracle/ro_oracle.c -lm
( cd $R/radio-observer_amd/host && g++ -O1 -g -fPIC -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o $R/build/asan/libro_host.so HipWaterfallBackend.cpp CsvLog.cpp BolidRecorder.cpp SnapshotRecorder.cpp \
    FITSWriter.cpp Frontends.cpp $R/tests/harness/host_capi.cpp -I. -L.. -lro_stft -Wl,-rpath,$R/radio-observer_amd )
export RO_ORACLE_LIB=$R/build/asan/libro_oracle.so RO_HOST_LIB=$R/build/asan/libro_host.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
cd $R
for T in tests/test_oracle.py tests/test_ring.py tests/test_host_cpu.py tests/test_golden_cpu.py; do
  python -m pytest $T -x -q -p no:cacheprovider
done
