#!/bin/bash
# the small configurations with the product and with a variant library (PBSO_LIB), alternating: realtime_x ms_per_step kernel_ms
cd "$GRAFT_REPO_ROOT"
V=$1
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --steps 40 --warmup 2 --buffers 86 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['timing']['device_pipeline_ms'],4))"; }
for rep in 1 2; do
for lib in product $V; do
  if [ $lib = product ]; then unset PBSO_LIB; else export PBSO_LIB=$PWD/openpbso_amd/variants/lib_$lib.so; fi
  echo "$lib c2 1x512:           $(run --objects 1 --modes 512)"
  echo "$lib c5 8x4096 scraping: $(run --objects 8 --modes 4096 --scenario scraping)"
  echo "$lib share 128x512:      $(run --objects 128)"
done; done
