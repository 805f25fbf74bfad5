#!/bin/bash
# the segmented scan rebuilt on the serial scan's pipelined body: parity, then the shares serial (1) against segmented (2)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_time_chunks.py tests/test_gpu_headline_shapes.py -x -q -m gpu 2>&1 | tail -3
PBSO_SCAN_KERNEL=2 timeout 900 python -m pytest tests/test_gpu_time_chunks.py tests/test_gpu_headline_shapes.py -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), '|', end=' ')"; }
for o in 128 256 512; do
 for sk in 1 2; do
  echo "$o x 512 x 86  sk=$sk: $(for i in 1 2 3; do PBSO_ENGINE_OPTS=scan_kernel=$sk run --objects $o --buffers 86 --steps 40 --warmup 3; done)"
  echo "$o x 512 x 860 sk=$sk: $(for i in 1 2 3; do PBSO_ENGINE_OPTS=scan_kernel=$sk run --objects $o; done)"
 done
done
bash scripts/debug/r05_timeline_share.sh 128 2 | head -24
