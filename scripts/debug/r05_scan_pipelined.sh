#!/bin/bash
# the serial scan as a software pipeline over its batches: parity first, then the scan alone (128 x 512 x 86, device idle) and the shares
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_time_chunks.py tests/test_gpu_headline_shapes.py -x -q -m gpu 2>&1 | tail -3
bash scripts/debug/r05_scan_abl.sh 128 512 2>&1 | grep "stop 9"
bash scripts/debug/r05_scan_abl.sh 1 512 2>&1 | grep "stop 9"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), '|', end=' ')"; }
for o in 128 256 512; do
  echo "$o x 512 x 86 : $(for i in 1 2 3; do run --objects $o --buffers 86 --steps 40 --warmup 3; done)"
  echo "$o x 512 x 860: $(for i in 1 2 3; do run --objects $o; done)"
done
bash scripts/debug/r05_timeline_share.sh 128 0
