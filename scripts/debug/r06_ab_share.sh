#!/bin/bash
# the 128- and 512-object shares (one-second steps) with the product's R = 4 builds and with a variant library: realtime_x ms_per_step kernel_ms
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --no-parity --steps 60 --warmup 3 --buffers 86 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; }
for rep in 1 2; do
for lib in "$@"; do
  export PBSO_LIB=$PWD/openpbso_amd/variants/lib_$lib.so
  echo "$lib share 128x512: $(run --objects 128)   share 512x512: $(run --objects 512)"
done; done
