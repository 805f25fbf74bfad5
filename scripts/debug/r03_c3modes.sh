#!/bin/bash
# configs[2] is bimodal run to run (0.21 / 0.36 ms per step with the same command): does the number of hardware queues decide?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { python bench.py --no-cpu-baseline --objects 64 --modes 256 --scenario listener --steps 40 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rt %.0f ms/step %.3f kernel %.3f pipeline %.2f' % (d['realtime_x'], d['ms_per_step'], d['roofline']['kernel_ms'], d['timing']['device_pipeline_ms']))"; }
for v in "" 2 8 16; do
  if [ -z "$v" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$v; fi
  for i in 1 2 3 4; do echo -n "GPU_MAX_HW_QUEUES=${v:-default}: "; run; done
done
