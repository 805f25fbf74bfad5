#!/bin/bash
# A/B of two builds on one box: openpbso_amd/libpbso_A.so (A) against the current library (B)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03ab
ARGS="${ARGS:---no-cpu-baseline --no-second-form --form block --steps 100 --warmup 3}"
for rep in 1 2; do for lib in A B; do
  if [ $lib = A ]; then export PBSO_LIB=$PWD/openpbso_amd/libpbso_A.so; else unset PBSO_LIB; fi
  timeout 600 python bench.py $ARGS > gpurun_out/r03ab/$lib$rep.json 2> gpurun_out/r03ab/$lib$rep.err
  python - gpurun_out/r03ab/$lib$rep.json $lib$rep <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "rt=%.1f ms/step=%.4f kernel=%.4f plan=%.3f enq=%.3f maxerr=%s" % (d["realtime_x"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["timing"]["host_plan_ms"], d["timing"]["host_enqueue_ms"], d.get("max_err")))
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done; done
