#!/bin/bash
# K1b at the headline shape, kernel time of variant builds (scripts/debug/r06_variant.sh; some give wrong results ON PURPOSE: --no-parity).
#   usage: r06_ablate.sh OUTFILE name1 name2 ...      ("product" = the library as built)
cd "$GRAFT_REPO_ROOT"
OUT=$1; shift
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --no-parity --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; }
for rep in 1 2; do
for v in "$@"; do
  if [ $v = product ]; then unset PBSO_LIB; else export PBSO_LIB=$PWD/openpbso_amd/variants/lib_$v.so; fi
  echo "$v (run $rep): realtime_x ms_per_step kernel_ms = $(run)" >> $OUT
done; done
