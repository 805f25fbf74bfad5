#!/bin/bash
# the preparation's kernels at wave priority 3 (build option PBSO_PREP_PRIO) against 0: configs[4], the shares, the headline
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), '|', end=' ')"; }
for lib in "" "$GRAFT_REPO_ROOT/build/variants/libprio3.so"; do
[ -n "$lib" ] && export PBSO_LIB=$lib && echo "== priority 3" || echo "== priority 0 (product)"
echo "c5 qnorm: $(for i in 1 2 3; do run --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2; done)"
echo "c5 qnorm off: $(for i in 1 2 3; do run --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40 --warmup 2; done)"
echo "128 x 512 x 860: $(for i in 1 2 3; do run --objects 128 --settle 80; done)"
echo "128 x 512 x 86: $(for i in 1 2 3; do run --objects 128 --buffers 86 --steps 40 --warmup 3; done)"
echo "512 x 512 x 860: $(for i in 1 2; do run --objects 512 --settle 20; done)"
echo "1024 x 512 x 860: $(for i in 1 2; do run; done)"
echo "1024 x 512 x 86: $(for i in 1 2; do run --buffers 86 --steps 40; done)"
echo "c3: $(for i in 1 2 3; do run --objects 64 --modes 256 --scenario listener --buffers 86 --steps 40 --warmup 2; done)"
done
