#!/bin/bash
# Collects the rocprofv3 evidence for profiles/<round>/ on the GPU box (run through gpurun):
#   kernel trace of the default bench, of the driver's flags and of one pair alone, then one --pmc pass per counter
#   group over short runs of both launch shapes (20 pairs per fill launch in both; a: two engines, b: one launch as the driver runs it).
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
for P in "w WRITE_SIZE" "f FETCH_SIZE" "sq SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES" "act SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  set -- $P; tag=$1; shift
  for cfg in "a 40" "b 20"; do
    set -- $cfg "$@"; sfx=$1; steps=$2; shift 2
    timeout -s KILL 200 rocprofv3 --pmc "$@" -d $O/pmc_${tag}_$sfx -o pmc_${tag}_$sfx -- python3 $R/bench.py --steps $steps --warmup 0 --no-extra --no-cpu-baseline > $O/pmc_${tag}_$sfx.json 2> $O/pmc_${tag}_$sfx.err
    echo "pmc $tag $sfx done"
  done
done
cd $R
for d in default driver one; do
  db=$(ls $O/$d/*/*.db $O/$d/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 tools/rocpd_summary.py $db $O/$d.csv > $O/$d.summary.txt
done
for tag in w f sq act; do
  for sfx in a b; do
    db=$(ls $O/pmc_${tag}_$sfx/*/*.db $O/pmc_${tag}_$sfx/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python3 tools/rocpd_summary.py $db $O/pmc_${tag}_$sfx.csv > $O/pmc_${tag}_$sfx.summary.txt
  done
  { head -1 $O/pmc_${tag}_a.csv; tail -n +2 $O/pmc_${tag}_a.csv; tail -n +2 $O/pmc_${tag}_b.csv; } > $O/pmc_${tag}.csv
done
ls $O
