#!/bin/bash
# the start gate, device form (stream_sync = 3: a waiting kernel) against host form (policy since round 5): the 128-object share at 860 buffers, eight runs each
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; }
for o in 128 512; do
for mode in 3 0; do
  echo "$o x 512 x 860 stream_sync=$mode: $(for i in 1 2 3 4 5 6; do PBSO_ENGINE_OPTS=stream_sync=$mode run --objects $o | tr '\n' ' '; echo -n '| '; done)"
done
done
