#!/bin/bash
# the preparation forked in two streams: parity with the fork forced wherever there is anything to fork, then configs[4] forked (policy) against not
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
PBSO_PREP_SPLIT=2 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_time_chunks.py tests/test_gpu_fullsize.py tests/test_gpu_headline_shapes.py -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['parity']['pass'], '|', end=' ')"; }
for sp in 0 1; do
echo "c5 qnorm split=$sp: $(for i in 1 2 3; do PBSO_PREP_SPLIT=$sp run --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2; done)"
echo "c5 qnorm off split=$sp: $(for i in 1 2 3; do PBSO_PREP_SPLIT=$sp run --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40 --warmup 2; done)"
echo "c5 qnorm 10 s split=$sp: $(for i in 1 2; do PBSO_PREP_SPLIT=$sp run --objects 8 --modes 4096 --scenario scraping; done)"
done
bash scripts/debug/r05_timeline_c5.sh sample | head -24
