#!/bin/bash
# Collects the rocprofv3 evidence for profiles/<round>/ on the GPU box (run through gpurun):
#   kernel trace of the default bench, of the driver's flags and of one pair alone, then one --pmc pass per counter
#   group over a short run (one engine, 1 + 3 launches of 20 pairs, one after the other: the counters of a launch do not
#   depend on what is queued behind it).
# usage: tools/profile_round.sh <outdir under gpurun_out>
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/default -o default -- python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/driver -o driver -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
echo "driver flags done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/one -o one -- python3 $R/bench.py --slots 1 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_slots_1.json 2> $O/bench_slots_1.err
echo "slots 1 done"
for P in "w WRITE_SIZE" "f FETCH_SIZE" "sq SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES" "act SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
         "stalls_a SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
         "stalls_b SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE" \
         "stalls_c SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"; do
  set -- $P; tag=$1; shift
  timeout -s KILL 100 rocprofv3 --pmc "$@" -d $O/pmc_$tag -o pmc_$tag -- python3 $R/bench.py --steps 3 --warmup 0 --engines 1 --no-extra --no-cpu-baseline > $O/pmc_$tag.json 2> $O/pmc_$tag.err
  echo "pmc $tag done"
done
cd $R
for d in default driver one pmc_w pmc_f pmc_sq pmc_act pmc_stalls_a pmc_stalls_b pmc_stalls_c; do
  db=$(ls $O/$d/*/*.db $O/$d/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 tools/rocpd_summary.py $db $O/$d.csv > $O/$d.summary.txt
done
ls $O
