#!/bin/bash
# with the scratch-engine clock ramp in front of every leg: the shares alone (default settle), c5 / c3 / c2, then the default line
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), '|', end=' ')"; }
for o in 128 256 512; do
  echo "$o x 512 x 860 ramp 150: $(for i in 1 2 3; do run --objects $o; done)"
  echo "$o x 512 x 860 ramp 0  : $(for i in 1 2; do run --objects $o --clock-ramp-ms 0; done)"
done
echo "128 x 512 x 86: $(for i in 1 2 3; do run --objects 128 --buffers 86 --steps 40 --warmup 3; done)"
echo "c5 qnorm: $(for i in 1 2; do run --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2; done)"
echo "c5 qnorm off: $(for i in 1 2; do run --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40 --warmup 2; done)"


python bench.py > gpurun_out/r05_bench_default_ramp.json 2> gpurun_out/r05_bench_default_ramp.err; echo "default rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_default_ramp.json'))
print(d['realtime_x'], d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['pass'])
for s in d['strong_share']['shares']: print(s['n_gpus'], s['ms_per_step_min_median_max'], s['implied_efficiency_min_median_max'])
print(d['steps_of_one_second']['realtime_x'])
"
