#!/bin/bash
# GPU box: the SQ counters of tools/r3/pmc_k32.sh for any bench shape.  usage: pmc_shape.sh TAG BINS OVERLAP ROWS
# (RO_STFT_LIB / RO_SLOTS etc. are inherited from the environment: export them before, never through `env` behind rocprofv3)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="python3 $ROOT/bench.py --bins $2 --overlap $3 --rows $4 --steps 3 --warmup 1 --prewarm 2 --no-cpu-baseline --no-parity --no-strict --no-streaming"
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES" \
         "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS" \
         "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- $SHORT > /dev/null 2> $OUT/p$i.log || echo "pmc pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT 2>&1 | grep -A40 "stft" > $OUT/SUMMARY.txt
cat $OUT/SUMMARY.txt
