#!/bin/bash
# Stall picture of the 16-pair fill launch: three --pmc passes (8 SQ counters each) over a short default bench.
# usage (through gpurun): tools/pmc_fill.sh <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {
  tag=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" -d $O/$tag -o $tag -- python3 $R/bench.py --steps 32 --warmup 0 --no-extra --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; exit 1; }
  echo "$tag done"
}
run a SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM
run c SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM
cd $R
for d in a b c; do
  db=$(ls $O/$d/*/*.db $O/$d/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 tools/rocpd_summary.py $db $O/$d.csv > $O/$d.summary.txt
done
ls $O
