#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-second-form --no-strong-share --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x']), 'x', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4))"; }
for rep in 1 2; do
for rp in 2 1 0; do
echo "rotate $rp: 128x512 $(PBSO_ROTATE_PRIO=$rp run --objects 128) | 512x512 $(PBSO_ROTATE_PRIO=$rp run --objects 512) | default $(PBSO_ROTATE_PRIO=$rp run)"
done; done
