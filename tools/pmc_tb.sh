#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2/pmc_tb
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES -d $O/a -o a -- python3 $R/tools/oneoff/prof_tb.py 16 > $O/a.out 2>&1
echo "a rc=$?"
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SMEM -d $O/b -o b -- python3 $R/tools/oneoff/prof_tb.py 16 > $O/b.out 2>&1
echo "b rc=$?"
cd $R
for d in a b; do
  db=$(ls $O/$d/*/*.db $O/$d/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 tools/rocpd_summary.py $db $O/$d.csv > $O/$d.summary.txt
done
ls $O
