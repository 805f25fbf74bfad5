#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for st in 20 40 50 60 80 100; do
python bench.py --objects 128 --steps $st --warmup 5 --no-cpu-baseline --no-second-form --no-strong-share --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps', d['steps'], round(d['realtime_x']), 'x', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4), 'enqueue', round(d['timing']['host_enqueue_ms'],4), 'plan', round(d['timing']['host_plan_ms'],4))"
done
