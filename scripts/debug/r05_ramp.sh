#!/bin/bash
# do the strong-scaling shares run inside the shader clock's ramp?  the 128 / 256-object shares with the default settle (4 steps of 860
# buffers: 5 - 10 ms of device time) against 80 steps (> 100 ms), and the headline with 4 against 12
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), '|', end=' ')"; }
for o in 128 256 512; do
 for s in 4 80; do
  echo "$o x 512 x 860 settle=$s: $(for i in 1 2 3; do run --objects $o --settle $s; done)"
 done
done
for s in 4 12; do echo "1024 x 512 x 860 settle=$s: $(for i in 1 2 3; do run --settle $s; done)"; done
for s in 40 400; do echo "128 x 512 x 86 settle=$s: $(for i in 1 2 3; do run --objects 128 --buffers 86 --steps 40 --settle $s; done)"; done
