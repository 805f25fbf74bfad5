#!/bin/bash
# the stream hand-over: values in signal memory (default) against events (stream_sync=1), every BASELINE configuration
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-second-form --no-strong-share "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x']), 'x', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4), 'host', round(d['timing']['host_ms'],3), 'err', d.get('max_err'), 'pass', d.get('parity_pass'))"; }
for sync in ${SYNCS:-0 1 0 1 0 1}; do
export PBSO_ENGINE_OPTS=stream_sync=$sync
echo "== stream_sync=$sync"
echo "1x512     $(run --objects 1 --modes 512)"
echo "listener  $(run --objects 64 --modes 256 --scenario listener)"
echo "scraping  $(run --objects 8 --modes 4096 --scenario scraping)"
echo "128x512   $(run --objects 128 --modes 512)"
echo "256x512   $(run --objects 256 --modes 512)"
echo "512x512   $(run --objects 512 --modes 512)"
echo "default   $(run)"
done
