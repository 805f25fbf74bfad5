#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --buffers 86 --steps 30 --warmup 3 --no-cpu-baseline --no-second-form --no-strong-share --objects 8 --modes 4096 --scenario scraping "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x']), 'x', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4), 'host', round(d['timing']['host_ms'],3), 'err', d.get('parity',{}).get('max_err') if isinstance(d.get('parity'),dict) else d.get('max_err'))"; }
for rep in 1 2; do
echo "policy qnorm   $(run)"
echo "policy noqnorm $(run --no-qnorm)"
done
for nc in 2 3 4; do
echo "nc=$nc qnorm   $(PBSO_PIPE_CONSUMERS=$nc run)"
echo "nc=$nc noqnorm $(PBSO_PIPE_CONSUMERS=$nc run --no-qnorm)"
done
