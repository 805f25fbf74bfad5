#!/bin/bash
# configs[2] with the builds of earlier commits (worktrees under .wt/, each with its own bench.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
for c in c7a521c c94bb8b 9ff8822 HEAD; do
  if [ $c = HEAD ]; then cd $R; else cd $R/.wt/$c; fi
  for i in 1 2; do echo -n "$c: "; python bench.py --no-cpu-baseline --no-parity --objects 64 --modes 256 --scenario listener --steps 40 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rt %.0f ms/step %.3f kernel %.3f pipeline %.2f plan %.3f enq %.3f' % (d['realtime_x'], d['ms_per_step'], d['roofline']['kernel_ms'], d['timing']['device_pipeline_ms'], d['timing']['host_plan_ms'], d['timing']['host_enqueue_ms']))"; done
done
