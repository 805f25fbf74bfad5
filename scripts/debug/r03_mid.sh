#!/bin/bash
# mid-size scenes (more chunks of 64 modes than the pipeline kernel is given by default): K1p with the bound lifted against K1b
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { python bench.py --no-cpu-baseline --no-parity --no-second-form --modes-per-lane 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rt %.0f ms/step %.3f kernel %.3f R=%s split=%s' % (d['realtime_x'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('modes_per_lane'), d['roofline']['kernel']))"; }
for shape in "16 4096" "128 512" "192 512" "256 512"; do
  set -- $shape
  echo -n "$1 x $2 K1b (engine's choice of R): "; python bench.py --no-cpu-baseline --no-parity --no-second-form --objects $1 --modes $2 --steps 40 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rt %.0f ms/step %.3f kernel %.3f R=%s %s' % (d['realtime_x'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('modes_per_lane'), d['roofline']['kernel']))"
  echo -n "$1 x $2 K1p:                        "; PBSO_SPLIT_MAX_CHUNKS=100000 run --objects $1 --modes $2 --steps 40 --warmup 2
done
