#!/bin/bash
# configs[4] (8 x 4096 scraping) with openpbso_amd/libpbso_A.so (A) against the current library (B), qnorm off and on
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03c5
for q in off sample; do for lib in A B; do
  if [ $lib = A ]; then export PBSO_LIB=$PWD/openpbso_amd/libpbso_A.so; else unset PBSO_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-parity --objects 8 --modes 4096 --scenario scraping --qnorm $q --steps 40 --warmup 2 > gpurun_out/r03c5/$lib$q.json 2> gpurun_out/r03c5/$lib$q.err
  python - gpurun_out/r03c5/$lib$q.json $lib-$q <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "rt=%.1f ms/step=%.4f kernel=%.4f pipeline=%.3f plan=%.3f enq=%.3f" % (d["realtime_x"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["timing"]["device_pipeline_ms"], d["timing"]["host_plan_ms"], d["timing"]["host_enqueue_ms"]))
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done; done
