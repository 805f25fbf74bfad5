#!/bin/bash
# round 4, first look at the time-chunked launches: new tests, then the shapes the verdict names
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_time_chunks.py -x -q 2>&1 | tail -15 > gpurun_out/r04/tc_tests.log
for o in 128 256 512; do
  timeout 300 python bench.py --objects $o --modes 512 --steps 20 --warmup 5 --no-cpu-baseline --no-second-form > gpurun_out/r04/bench_${o}x512.json 2> gpurun_out/r04/bench_${o}x512.err
done
timeout 300 python bench.py --objects 1 --modes 512 --steps 40 --warmup 5 --no-cpu-baseline --no-second-form > gpurun_out/r04/bench_1x512.json 2> gpurun_out/r04/bench_1x512.err
timeout 300 python bench.py --objects 64 --modes 256 --scenario listener --steps 40 --warmup 5 --no-cpu-baseline --no-second-form > gpurun_out/r04/bench_64x256.json 2> gpurun_out/r04/bench_64x256.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-form > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
tail -3 gpurun_out/r04/tc_tests.log
for f in gpurun_out/r04/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'x', round(d.get('realtime_x', 0)), 'ms', round(d['ms_per_step'],4), 'kernel', d.get('timing',{}).get('kernel_ms'), 'parity', d.get('parity',{}).get('max_err_over_peak'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
