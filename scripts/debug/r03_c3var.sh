#!/bin/bash
# configs[2] (64 x 256, a listener move per buffer) under a few switches: which one costs the step its overlap
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo -n "$1: "; shift; env "$@" python bench.py --no-cpu-baseline --no-parity --objects 64 --modes 256 --scenario listener --steps 40 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rt %.0f ms/step %.3f kernel %.3f pipeline %.2f plan %.3f enq %.3f' % (d['realtime_x'], d['ms_per_step'], d['roofline']['kernel_ms'], d['timing']['device_pipeline_ms'], d['timing']['host_plan_ms'], d['timing']['host_enqueue_ms']))"; }
run default A=1
run timing_every_0 PBSO_TIMING_EVERY=0
run timing_every_1 PBSO_TIMING_EVERY=1
run k2prio0 PBSO_K2_PRIO=0
run split0 PBSO_SPLIT=0
run host_profile PBSO_HOST_PROFILE=1
run default A=1
