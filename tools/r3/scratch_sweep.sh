#!/bin/bash
# GPU box: the large transforms' scratch form with chunks of 2048 ... 64 MiB of folded sub-rows (variants built by
# `tools/ab_build.sh sMB -DRO_SPEC_SCRATCH_MB=MB`): does a chunk that fits the 256 MiB Infinity Cache keep the
# trips between fold / transform / interleave on the die?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for SHAPE in "524288 393216 1024" "262144 196608 2048" "1048576 786432 512" "524288 262144 1024"; do
  set -- $SHAPE
  echo "== --bins $1 --overlap $2 --rows $3"
  AB_ARGS="--bins $1 --overlap $2 --rows $3 --no-strict --no-streaming" bash $ROOT/tools/ab_run.sh 2 s2048 s512 s256 s128 s64
done
