#!/bin/bash
# GPU box: SQ / instruction-cache counters of the N = 32768 kernel (separate --pmc passes, no tracing alongside).
# usage: tools/r3/pmc_k32.sh <tag>  -> gpurun_out/pmc_<tag>/SUMMARY.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_${1:-k32}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="python3 $ROOT/bench.py --steps 3 --warmup 1 --prewarm 2 --no-cpu-baseline --no-parity --no-strict --no-streaming"
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- $SHORT > /dev/null 2> $OUT/p$i.log || echo "pmc pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/SUMMARY.txt 2>&1
cat $OUT/SUMMARY.txt
