#!/bin/bash
# buffers per step (and per launch: chunk_buffers) against the rate: a launch's fixed costs -- ramp, write drain, the hand-over
# from the scan -- are per LAUNCH, so longer steps amortise them (SURVEY 8(d) quotes throughput at 860 buffers per step)
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 30 --warmup 3 --settle 20 --no-cpu-baseline --no-second-form --no-strong-share "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f x  %8.4f ms per step  max_err %.1e' % (d['realtime_x'], d['ms_per_step'], d['max_err']))"; }
for cfg in "86 128" "430 430" "860 860"; do set -- $cfg; export PBSO_CHUNK_BUFFERS=$2
for n in 128 256 512 1024; do echo "buffers per step $1, per launch <= $2, objects $n x 512: $(run --objects $n --buffers $1)"; done; done
