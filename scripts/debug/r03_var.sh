#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { tag=$1; shift; for i in 1 2 3; do env "$@" timeout 600 python bench.py --no-cpu-baseline --no-second-form --no-parity --steps 100 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print('$tag', 'rt=%.1f ms/step=%.4f kernel=%.4f plan=%.3f enq=%.3f'%(d['realtime_x'],d['ms_per_step'],d['roofline']['kernel_ms'],d['timing']['host_plan_ms'],d['timing']['host_enqueue_ms']))"; done; }
run default A=1
run noevents PBSO_TIMING_EVERY=0
run plan1 PBSO_PLAN_THREADS=1
run sets2 PBSO_LIB=$PWD/openpbso_amd/libpbso_A.so
