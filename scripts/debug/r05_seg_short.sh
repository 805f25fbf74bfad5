#!/bin/bash
# the scan cut along the time axis for SHORT chunks (one-second steps), clocks ramped: configs[4] with qnorm rows (6 chunks of 15), the 128 / 256-object shares (8 of 11)
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), '|', end=' ')"; }
for sk in 1 2; do
echo "c5 qnorm sk=$sk: $(for i in 1 2 3; do PBSO_ENGINE_OPTS=scan_kernel=$sk run --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2; done)"
echo "128 x 512 x 86 sk=$sk: $(for i in 1 2 3; do PBSO_ENGINE_OPTS=scan_kernel=$sk run --objects 128 --buffers 86 --steps 40 --warmup 3; done)"
echo "256 x 512 x 86 sk=$sk: $(for i in 1 2 3; do PBSO_ENGINE_OPTS=scan_kernel=$sk run --objects 256 --buffers 86 --steps 40 --warmup 3; done)"
done
