#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { PBSO_HOST_PROFILE=1 python bench.py --objects 128 --steps 60 --warmup 5 --no-cpu-baseline --no-second-form --no-strong-share "$@" 2>/tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x']), 'x', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4), 'host', d['timing']['host_ms'])"; grep "host profile" /tmp/err.txt | cut -c1-260; }
for rep in 1 2 3; do
echo "with parity: $(run)"
echo "no parity:   $(run --no-parity)"
done
