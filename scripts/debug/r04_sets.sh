#!/bin/bash
# plan sets: how far the preparation stream may run ahead of the oscillator bank (PBSO_N_SETS, compile time)
cd "$GRAFT_REPO_ROOT/openpbso_amd/csrc"
for n in 4 5; do
  for f in engine capi; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DPBSO_N_SETS=$n -c $f.cpp -o /tmp/${f}_$n.o; done
  hipcc --offload-arch=gfx950 -shared -fPIC kernels_iir.o kernels_block.o kernels_scan.o kernels_pipe.o kernels_exact.o /tmp/engine_$n.o loaders.o /tmp/capi_$n.o -o /tmp/libsets$n.so
done
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-second-form --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x']), 'x', round(d['ms_per_step'],4), 'ms')"; }
for n in 3 4 5; do
  [ $n = 3 ] && unset PBSO_LIB || export PBSO_LIB=/tmp/libsets$n.so
  for rep in 1 2; do
  echo "sets $n: 128x512 $(run --objects 128 --modes 512) | 256x512 $(run --objects 256 --modes 512) | 1x512 $(run --objects 1 --modes 512) | listener $(run --objects 64 --modes 256 --scenario listener) | scraping $(run --objects 8 --modes 4096 --scenario scraping) | default $(run)"
  done
done
