#!/bin/bash
# where the second preparation stream lives: one-buffer latency (two-stream row), configs[4], configs[1,2]
cd "$GRAFT_REPO_ROOT"
timeout 300 python scripts/latency.py 2>&1 | cut -c1-120
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), '|', end=' ')"; }
echo "c5 qnorm: $(for i in 1 2 3; do run --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2; done)"
echo "c5 qnorm off: $(for i in 1 2; do run --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40 --warmup 2; done)"
echo "c3: $(for i in 1 2 3; do run --objects 64 --modes 256 --scenario listener --buffers 86 --steps 40 --warmup 2; done)"
echo "c2: $(for i in 1 2 3; do run --objects 1 --modes 512 --buffers 86 --steps 40 --warmup 2; done)"
echo "128 x 512 x 86: $(for i in 1 2; do run --objects 128 --buffers 86 --steps 40 --warmup 3; done)"
