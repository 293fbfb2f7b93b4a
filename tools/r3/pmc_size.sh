#!/bin/bash
# GPU box: instruction counts of one transform size.  usage: tools/r3/pmc_size.sh BINS OVERLAP ROWS
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_size_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="python3 $ROOT/bench.py --bins $1 --overlap $2 --rows $3 --steps 3 --warmup 1 --prewarm 2 --no-cpu-baseline --no-parity --no-strict --no-streaming"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/p1 -- $SHORT > /dev/null 2> $OUT/p1.log || echo "pmc failed"
python3 $ROOT/tools/pmc_summary.py $OUT | grep -A9 "stft" 
