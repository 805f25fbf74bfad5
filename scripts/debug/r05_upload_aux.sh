#!/bin/bash
# forked launches upload through the second stream (an engine.cpp that was NOT kept: it read PBSO_PREP_SPLIT=3 as "fork, upload on the first stream"; the committed library clamps 3 to 2): configs[4] with qnorm rows, then parity with every fork forced
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['parity']['pass'], '|', end=' ')"; }
echo "c5 qnorm, upload on the second stream: $(for i in 1 2; do run --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2; done)"
echo "c5 qnorm, upload on the first stream : $(PBSO_PREP_SPLIT=3 run --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2)"
PBSO_PREP_SPLIT=2 timeout 150 python -m pytest tests/test_gpu_time_chunks.py tests/test_gpu_fullsize.py tests/test_gpu_headline_shapes.py -x -q -m gpu 2>&1 | tail -2
