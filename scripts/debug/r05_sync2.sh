#!/bin/bash
# the preparation -> bank hand-over as a value in signal memory (stream_sync = 2) against the event (policy): shares at 860 buffers, headline
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), '|', end=' ')"; }
for o in 128 256 512 1024; do
 for m in 0 2; do
  echo "$o x 512 x 860 stream_sync=$m: $(for i in 1 2 3 4; do PBSO_ENGINE_OPTS=stream_sync=$m run --objects $o; done)"
 done
done
for o in 128 256; do
 for m in 0 2; do
  echo "$o x 512 x 86 stream_sync=$m: $(for i in 1 2 3; do PBSO_ENGINE_OPTS=stream_sync=$m run --objects $o --buffers 86 --steps 40 --warmup 3; done)"
 done
done
