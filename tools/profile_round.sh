#!/bin/bash
# Collects the rocprofv3 evidence for profiles/<round>/ on the GPU box (run through gpurun):
#   kernel trace of the default bench and of one pair alone, then one --pmc pass per counter group
#   over a short run of the default job (8 pairs per fill launch).
# usage: tools/profile_round.sh <outdir under gpurun_out>
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/default -o default -- python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/one -o one -- python3 $R/bench.py --slots 1 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_slots_1.json 2> $O/bench_slots_1.err
echo "slots 1 done"
for P in "w WRITE_SIZE" "f FETCH_SIZE" "sq SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES" "act SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  set -- $P; tag=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" -d $O/pmc_$tag -o pmc_$tag -- python3 $R/bench.py --steps 32 --warmup 0 --no-extra --no-cpu-baseline > $O/pmc_$tag.json 2> $O/pmc_$tag.err
  echo "pmc $tag done"
done
cd $R
for d in default one pmc_w pmc_f pmc_sq pmc_act; do
  db=$(ls $O/$d/*/*.db $O/$d/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 tools/rocpd_summary.py $db $O/$d.csv > $O/$d.summary.txt
done
ls $O
