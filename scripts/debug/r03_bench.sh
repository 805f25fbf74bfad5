#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03i; export TMPDIR=/tmp; R=$PWD
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03i/bench_default_driverflags.json 2> gpurun_out/r03i/b0.err; echo rc=$?; tail -3 gpurun_out/r03i/b0.err
timeout 600 python bench.py --no-cpu-baseline --objects 64 --modes 256 --scenario listener --steps 40 --warmup 2 > gpurun_out/r03i/bench_c3.json 2> gpurun_out/r03i/b1.err; echo rc=$?; tail -3 gpurun_out/r03i/b1.err
timeout 600 python bench.py --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --steps 40 --warmup 2 > gpurun_out/r03i/bench_c5.json 2> gpurun_out/r03i/b2.err; echo rc=$?; tail -3 gpurun_out/r03i/b2.err
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -x -q -m gpu 2>&1 | tail -5
for f in gpurun_out/r03i/bench_*.json; do python - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], "dtype=%s rt=%.1f ms/step=%.3f kernel=%.3f plan=%.3f enq=%.3f frac=%.3f hbm=%.3f maxerr=%s" % (d["dtype"][:12], d["realtime_x"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["timing"]["host_plan_ms"], d["timing"]["host_enqueue_ms"], d["roofline"]["frac"], d["hbm_frac"], d.get("max_err")))
for k in ("mixed_precision_projection", "host_delivered"):
    if k in d: print("   ", k, {kk: vv for kk, vv in d[k].items() if kk not in ("note", "roofline")})
PY
done
