#!/bin/bash
# GPU box: row priority on / off (build/ab/libro_stft_rp1.so = -DRO_DIAG=1, rp0 = -DRO_DIAG=1 -DRO_ROW_PRIO=0), interleaved
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for S in "$@"; do
  set -- $S
  echo "== --bins $1 --overlap $2 --rows $3"
  AB_ARGS="--bins $1 --overlap $2 --rows $3 --no-strict --no-streaming" bash $ROOT/tools/ab_run.sh 2 rp0 rp1 | grep -v parity
done
