#!/bin/bash
# the under-filled BASELINE configurations, time-split kernel on / off
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03s
run() { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline --no-second-form "$@" > gpurun_out/r03s/$name.json 2> gpurun_out/r03s/$name.err; python - gpurun_out/r03s/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-28s rt=%8.1f ms/step=%.3f kernel=%.3f plan=%.3f maxerr=%s split=%s" % (sys.argv[2], d["realtime_x"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["timing"]["host_plan_ms"], d.get("max_err"), d["config"].get("split_launches")))
except Exception as e:
    print(sys.argv[2], "bad", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
for sp in 1 0; do
export PBSO_SPLIT=$sp
run c2_1x512_split$sp --objects 1 --modes 512 --steps 40 --warmup 2
run c3_64x256_listener_split$sp --objects 64 --modes 256 --scenario listener --steps 40 --warmup 2
run c5_scraping_split$sp --objects 8 --modes 4096 --scenario scraping --steps 40 --warmup 2
run c5_scraping_qoff_split$sp --objects 8 --modes 4096 --scenario scraping --qnorm off --steps 40 --warmup 2
done
