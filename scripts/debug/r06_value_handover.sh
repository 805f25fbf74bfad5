#!/bin/bash
# the hand-over of a launch to the bank's stream through a value in signal memory (stream_sync = 2) against events (policy), small configurations
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --steps 200 --warmup 5 --buffers 86 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing']; print(round(d['realtime_x'],1), round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'plan', round(t['host_plan_ms'],4), d['parity']['pass'])"; }
for rep in 1 2; do
for opts in "" "stream_sync=2"; do
echo "'$opts' c3 64x256 listener st=1: $(PBSO_ENGINE_OPTS=$opts run --objects 64 --modes 256 --scenario listener --submit-thread 1)"
echo "'$opts' c2 1x512:                $(PBSO_ENGINE_OPTS=$opts run --objects 1 --modes 512)"
echo "'$opts' c5 8x4096 scraping:      $(PBSO_ENGINE_OPTS=$opts run --objects 8 --modes 4096 --scenario scraping)"
echo "'$opts' share 16x512:            $(PBSO_ENGINE_OPTS=$opts run --objects 16)"
echo "'$opts' share 64x512:            $(PBSO_ENGINE_OPTS=$opts run --objects 64)"
echo "'$opts' share 128x512:           $(PBSO_ENGINE_OPTS=$opts run --objects 128)"
echo "'$opts' share 256x512:           $(PBSO_ENGINE_OPTS=$opts run --objects 256)"
echo "'$opts' share 512x512:           $(PBSO_ENGINE_OPTS=$opts run --objects 512)"
done; done
