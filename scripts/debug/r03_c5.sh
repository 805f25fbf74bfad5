#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03c; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_listener_mix_edges.py -x -q -m gpu 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k config5 2>&1 | tail -8
PBSO_FUZZ_SEEDS=60 PBSO_FUZZ_SHAPE_SEEDS=30 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -8
for q in sample off; do
timeout 600 python bench.py --no-cpu-baseline --no-parity --form block --objects 8 --modes 4096 --scenario scraping --qnorm $q --steps 40 --warmup 2 > gpurun_out/r03c/bench_c5_$q.json 2> gpurun_out/r03c/c5_$q.err; echo rc=$?
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03c/c5stats -- python3 $R/bench.py --no-cpu-baseline --no-parity --form block --objects 8 --modes 4096 --scenario scraping --steps 30 --warmup 2 > $R/gpurun_out/r03c/c5log.txt 2>&1)
f=$(find gpurun_out/r03c/c5stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03c/c5_kernel_stats.csv; rm -rf gpurun_out/r03c/c5stats
cut -c1-60,250-400 gpurun_out/r03c/c5_kernel_stats.csv | head -6
for f in gpurun_out/r03c/bench_*.json; do python - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], "rt=%.1f ms/step=%.3f kernel=%.3f plan=%.3f enq=%.3f" % (d["realtime_x"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["timing"]["host_plan_ms"], d["timing"]["host_enqueue_ms"]), d["roofline"]["kernel"])
PY
done
