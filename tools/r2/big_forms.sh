#!/bin/bash
# GPU box: the large-transform forms side by side on one device (diagnostic build with RO_DIAG_KNOBS in build/ab/).
# usage: big_forms.sh [forms...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export RO_STFT_LIB=$ROOT/build/ab/libro_stft_diag.so
for F in ${@:-dif fold twopass}; do
  for B in "65536 49152 8192" "131072 98304 4096" "262144 196608 2048" "524288 393216 1024" "1048576 786432 512"; do
    echo -n "$F "; RO_BIG_FORM=$F bash $ROOT/tools/sizes_run.sh "$B"
  done
done
