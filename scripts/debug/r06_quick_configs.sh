#!/bin/bash
# the small configurations, quickly (one-second steps): realtime_x ms_per_step kernel_ms
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --steps 40 --warmup 2 --buffers 86 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['parity']['pass'])"; }
echo "c2 1x512:            $(run --objects 1 --modes 512)"
echo "c3 64x256 listener:  $(run --objects 64 --modes 256 --scenario listener)"
echo "c5 8x4096 scraping:  $(run --objects 8 --modes 4096 --scenario scraping)"
echo "c5 qnorm off:        $(run --objects 8 --modes 4096 --scenario scraping --qnorm off)"
echo "share 128x512:       $(run --objects 128)"
echo "share 512x512:       $(run --objects 512)"
