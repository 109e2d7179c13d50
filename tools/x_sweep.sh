#!/bin/bash
# Fill duration of one launch of 20 pairs (tools/pk_time.py) with library variants built with the packed kernel's
# timing-only switches, e.g.
#   make PKEXTRA="-DPSA_PK_NOSYNC=1 -DPSA_PK_X=4" && cp parallel-sequence-alignment_amd/lib/libpsa_hip.so tools/exp/libpsa_x4.so
#   (touch csrc/psa_fill_pk.hip.h between variants; `make` again afterwards restores the product build)
# usage (on the GPU box): tools/x_sweep.sh x0 x4 ...     tools/exp/ is git-ignored but travels with gpurun.
for v in "$@"; do
  echo "== $v"
  PSA_LIB_PATH=$PWD/tools/exp/libpsa_$v.so timeout -k 10 120 python -u tools/pk_time.py 20 2>&1 | grep "fill ms" || exit 1
done
