#!/bin/bash
# Round 5 evidence, the short-run lines again after bench.py's settle steps follow the device time (auto_settle): one-second-step lines of
# configs[1,2,4] and of the shares, and the default line (its one-second leg and shares settle by the same rule)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p5; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p5
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$? $(python -c "import json; d=json.load(open('$O/bench_$name.json')); print(round(d['realtime_x'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['settle_steps'], d['parity']['pass'])")"; }
b c2_1x512 --no-cpu-baseline --objects 1 --modes 512 --buffers 86 --steps 40 --warmup 2
b c3_64x256_listener --no-cpu-baseline --objects 64 --modes 256 --scenario listener --buffers 86 --steps 40 --warmup 2
b c5_8x4096_scraping --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2
b c5_8x4096_scraping_qnorm_off --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40 --warmup 2
for o in 512 256 128; do
b share_${o}x512 --no-cpu-baseline --no-second-form --no-strong-share --objects $o --buffers 86 --steps 40 --warmup 3
b share_${o}x512_10s_steps --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --objects $o
done
b default
b driver_flags_steps20_warmup5 --steps 20 --warmup 5
