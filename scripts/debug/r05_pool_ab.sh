#!/bin/bash
# planner pool: the headline (host_plan_ms, value) three times, and the 128 / 512-object shares at 860 buffers six times each
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value'],1), round(d['ms_per_step'],4), d['timing'].get('host_plan_ms'), round(d['roofline']['kernel_ms'],4))"
done
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; }
for o in 128 512; do
  echo "$o x 512 x 860: $(for i in 1 2 3 4 5 6; do run --objects $o | tr '\n' ' '; echo -n '| '; done)"
done
