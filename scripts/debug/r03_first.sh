#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03a; export TMPDIR=/tmp; R=$PWD
./scripts/debug/mfma_valu_mix > gpurun_out/r03a/mfma_valu_mix.txt 2>&1; echo "mix rc=$?"
timeout 600 python bench.py --no-cpu-baseline --form block --steps 20 --warmup 5 > gpurun_out/r03a/bench_block_f32_driverflags.json 2> gpurun_out/r03a/b1.err; echo rc=$?
timeout 600 python bench.py --no-cpu-baseline --no-f32-leg --steps 20 --warmup 5 > gpurun_out/r03a/bench_bf16_driverflags.json 2> gpurun_out/r03a/b2.err; echo rc=$?
timeout 600 python bench.py --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --steps 40 --warmup 2 > gpurun_out/r03a/bench_c5.json 2> gpurun_out/r03a/b3.err; echo rc=$?
timeout 600 python bench.py --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off --steps 40 --warmup 2 > gpurun_out/r03a/bench_c5_qoff.json 2> gpurun_out/r03a/b4.err; echo rc=$?
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03a/c5stats -- python3 $R/bench.py --no-cpu-baseline --no-parity --objects 8 --modes 4096 --scenario scraping --steps 30 --warmup 2 > $R/gpurun_out/r03a/c5log.txt 2>&1)
f=$(find gpurun_out/r03a/c5stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03a/c5_kernel_stats.csv; rm -rf gpurun_out/r03a/c5stats
head -5 gpurun_out/r03a/c5_kernel_stats.csv | cut -c1-200
for f in gpurun_out/r03a/bench_*.json; do python - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], "rt=%.1f ms/step=%.3f kernel=%.3f plan=%.3f enq=%.3f" % (d["realtime_x"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["timing"]["host_plan_ms"], d["timing"]["host_enqueue_ms"]))
PY
done
cat gpurun_out/r03a/mfma_valu_mix.txt
