#!/bin/bash
# round 5: 8 x 4096 sustained scraping, time-chunked (dense increments + scan + CHUNKED FORCED block kernel): team shape x chunk length
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --buffers 86 --steps 30 --warmup 3 --no-cpu-baseline --no-second-form --no-strong-share --objects 8 --modes 4096 --scenario scraping "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x']), 'x', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4), 'host', round(d['timing']['host_ms'],3), 'err', d.get('parity',{}).get('max_err') if isinstance(d.get('parity'),dict) else d.get('max_err'))"; }
echo "policy qnorm   $(run)"
echo "policy noqnorm $(run --no-qnorm)"
for shape in 1 2 4; do
for tc in 2 4 6 11 22; do
  echo "shape=$shape tc=$tc qnorm   $(PBSO_TC_SHAPE=$shape PBSO_TIME_CHUNKS=$tc run)"
  echo "shape=$shape tc=$tc noqnorm $(PBSO_TC_SHAPE=$shape PBSO_TIME_CHUNKS=$tc run --no-qnorm)"
done
done
