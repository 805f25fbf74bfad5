#!/bin/bash
# the strong-scaling shares of 1024 x 512 with one-second and ten-second steps: serial scan against the scan cut along the time axis
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x']), 'x', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4), 'err', d.get('max_err'))"; }
for o in 128 256 512; do
  for sk in 1 2; do
    echo "$o x 512 x 86  scan_kernel=$sk  $(PBSO_ENGINE_OPTS=scan_kernel=$sk run --objects $o --buffers 86 --steps 40 --warmup 3)"
    echo "$o x 512 x 860 scan_kernel=$sk  $(PBSO_ENGINE_OPTS=scan_kernel=$sk run --objects $o)"
  done
done
