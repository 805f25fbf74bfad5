#!/bin/bash
# plan sets 2 / 3 / 4 (openpbso_amd/libpbso_S2.so, _S3.so, the product) on configs[2], the headline and configs[4]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { python bench.py --no-cpu-baseline --no-parity --no-second-form "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rt %.0f ms/step %.3f kernel %.3f pipeline %.2f plan %.3f enq %.3f' % (d['realtime_x'], d['ms_per_step'], d['roofline']['kernel_ms'], d['timing']['device_pipeline_ms'], d['timing']['host_plan_ms'], d['timing']['host_enqueue_ms']))"; }
for s in 2 3 4; do
  if [ $s = 4 ]; then unset PBSO_LIB; else export PBSO_LIB=$PWD/openpbso_amd/libpbso_S$s.so; fi
  echo -n "sets $s c3:       "; run --objects 64 --modes 256 --scenario listener --steps 40 --warmup 2
  echo -n "sets $s c2:       "; run --objects 1 --modes 512 --steps 40 --warmup 2
  echo -n "sets $s headline: "; run --steps 20 --warmup 5
  echo -n "sets $s c5 off:   "; run --objects 8 --modes 4096 --scenario scraping --qnorm off --steps 40 --warmup 2
done
