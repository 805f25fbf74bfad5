#!/bin/bash
# 64 x 256 listener, 86 buffers per step: the window between two oscillator banks (listener lookup, scan, hand-over to the bank's stream)
# under the engine's options
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --steps 200 --warmup 5 --buffers 86 --objects 64 --modes 256 --scenario listener "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing']; print(round(d['realtime_x'],1), round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'plan', round(t['host_plan_ms'],4), d['parity']['pass'])"; }
for rep in 1 2; do
for opts in "" "stream_sync=2" "scan_kernel=2" "stream_sync=2,scan_kernel=2" "stream_sync=3" ; do
echo "submit_thread=1 PBSO_ENGINE_OPTS='$opts': $(PBSO_ENGINE_OPTS=$opts run --submit-thread 1)"
done; done
