#!/bin/bash
# Round 5 evidence, last pass: the second preparation stream is created at the first fork only (created with the engine it doubled the
# cross-stream hand-over of one-buffer steps: 58 -> 112 us in r05_realtime_latency.txt's latency_path = -1 row) -- the lines of small scenes again
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p5; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p5
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$? $(python -c "import json; d=json.load(open('$O/bench_$name.json')); print(round(d['realtime_x'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['settle_steps'], d['parity']['pass'])")"; }
b c2_1x512 --no-cpu-baseline --objects 1 --modes 512 --buffers 86 --steps 40 --warmup 2
b c3_64x256_listener --no-cpu-baseline --objects 64 --modes 256 --scenario listener --buffers 86 --steps 40 --warmup 2
b c2_1x512_10s_steps --no-cpu-baseline --objects 1 --modes 512
b c3_64x256_listener_10s_steps --no-cpu-baseline --objects 64 --modes 256 --scenario listener
b c5_8x4096_scraping --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2
timeout 300 python scripts/latency.py > $O/realtime_latency.txt 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
