#!/bin/bash
# Counters (cycles, waits, VALU activity) of the fill launch in timing-only library variants (see tools/x_sweep.sh):
# SQ_BUSY_CYCLES / 32 shader engines / duration = the clock the launch ran at.  usage: tools/x_pmc.sh x0 x4
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/xpmc
for v in "$@"; do
  export PSA_LIB_PATH=$R/tools/exp/libpsa_$v.so
  timeout -s KILL 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $R/gpurun_out/xpmc/$v -o $v -- python3 $R/tools/pk_time.py 20 > $R/gpurun_out/xpmc/$v.txt 2>&1 || { echo "$v failed"; exit 1; }
  db=$(ls $R/gpurun_out/xpmc/$v/*/*.db $R/gpurun_out/xpmc/$v/*.db 2>/dev/null | head -1)
  python3 $R/tools/rocpd_summary.py $db $R/gpurun_out/xpmc/$v.csv > /dev/null
  echo "== $v"; grep "fill ms" $R/gpurun_out/xpmc/$v.txt | tail -1; grep "fill_strips_pk" $R/gpurun_out/xpmc/$v.csv | cut -d, -f3,5
done
