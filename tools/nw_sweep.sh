#!/bin/bash
# bench.py (after tools/check_packed_fill.py) against library variants tools/exp/libpsa_nw<name>.so, e.g. built with
# make PKNW=6 (strips per workgroup of the packed kernel) or PKEXTRA=-DPSA_PK_MASKED=1.  usage: tools/nw_sweep.sh 6 3 2
for nw in "$@"; do
  export PSA_LIB_PATH=$PWD/tools/exp/libpsa_nw$nw.so
  timeout -k 10 120 python -u tools/check_packed_fill.py > gpurun_out/nw${nw}_check.log 2>&1 || { echo "nw $nw check failed"; tail -3 gpurun_out/nw${nw}_check.log; exit 1; }
  timeout -k 10 200 python -u bench.py > gpurun_out/nw${nw}_bench.json 2> gpurun_out/nw${nw}_bench.err || { echo "nw $nw bench failed"; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/nw${nw}_bench.json"))
print("nw ${nw}: %.1f GCUPS  %.2f ms/step  kernel %.2f ms  frac %.3f  alone %.2f ms" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["alone"]["kernel_ms"]), flush=True)
PY
done
