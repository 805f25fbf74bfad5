#!/bin/bash
# secondary configurations of BASELINE.json (not the headline line)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/final
run() { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/final/bench_$name.json 2> gpurun_out/final/bench_$name.err; echo "$name rc=$?"; python - gpurun_out/final/bench_$name.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("  value=%.4g samples/s  rt=%.1fx  ms/step=%.3f  kernel_ms=%.3f  plan_ms=%.2f  R=%d W=%d" % (d["value"], d["realtime_x"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["timing"]["host_plan_ms"], d["config"]["modes_per_lane"], d["config"]["waves_per_object"]))
except Exception as e:
    print("  bad json", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
run c2_1x512 --objects 1 --modes 512 --steps 40 --warmup 2
run c3_64x256_listener --objects 64 --modes 256 --scenario listener --steps 40 --warmup 2
run c5_8x4096_scraping --objects 8 --modes 4096 --scenario scraping --steps 40 --warmup 2
# the N > 1 path on one GPU: two ranks started by bench.py, gloo through the host instead of RCCL (transport only)
PBSO_BENCH_BACKEND=gloo run 2ranks_one_gpu_gloo --gpus 2 --steps 20 --warmup 2
PBSO_DEVICE_PROFILES=0 run c5_8x4096_scraping_hostprof --objects 8 --modes 4096 --scenario scraping --steps 6 --warmup 2
