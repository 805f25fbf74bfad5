#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03k2; export TMPDIR=/tmp; R=$PWD
for v in base RNG SCAN BOTH; do
  if [ $v = base ]; then unset PBSO_LIB; else export PBSO_LIB=$R/openpbso_amd/libpbso_abl_$v.so; fi
  # tiny oscillator bank (64 modes): K2 runs practically alone
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03k2/$v -- python3 $R/bench.py --no-cpu-baseline --no-parity --form block --objects 8 --modes 64 --scenario scraping --steps 30 --warmup 2 > $R/gpurun_out/r03k2/$v.log 2>&1)
  f=$(find gpurun_out/r03k2/$v -name "*kernel_stats.csv" | head -1); echo "== $v"; python3 -c "
import csv,sys
for r in csv.reader(open('$f')):
    if 'force_profile' in r[0] or 'iir_block' in r[0]: print(r[0][:40], r[1:4])
"
  rm -rf gpurun_out/r03k2/$v
done
