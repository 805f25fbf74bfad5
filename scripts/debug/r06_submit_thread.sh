#!/bin/bash
# the second submitting thread (pbso_engine_desc::submit_thread): bit-identity tests, the suites again with the thread switched on
# through PBSO_ENGINE_OPTS, and the small configurations with / without it
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06st
timeout 900 python -m pytest tests/test_gpu_submit_thread.py -x -q > gpurun_out/r06st/tests_new.log 2>&1; echo "new tests rc=$?"; tail -3 gpurun_out/r06st/tests_new.log
PBSO_ENGINE_OPTS=submit_thread=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_time_chunks.py tests/test_gpu_direct_hits.py tests/test_gpu_listener_mix_edges.py -x -q -m gpu > gpurun_out/r06st/tests_opts.log 2>&1; echo "suites with the thread rc=$?"; tail -3 gpurun_out/r06st/tests_opts.log
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --steps 200 --warmup 5 --buffers 86 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing']; print(round(d['realtime_x'],1), round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'plan', round(t['host_plan_ms'],4), 'enq', round(t['host_enqueue_ms'],4), d['parity']['pass'])"; }
for rep in 1 2; do
for st in 0 1; do
echo "submit_thread=$st c3 64x256 listener:  $(run --objects 64 --modes 256 --scenario listener --submit-thread $st)"
echo "submit_thread=$st c2 1x512:            $(run --objects 1 --modes 512 --submit-thread $st)"
echo "submit_thread=$st c5 8x4096 scraping:  $(run --objects 8 --modes 4096 --scenario scraping --submit-thread $st)"
echo "submit_thread=$st share 128x512:       $(run --objects 128 --submit-thread $st)"
done
done
echo "submit_thread=1 headline (860 buffers): $(python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --submit-thread 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x'],1), round(d['ms_per_step'],4), d['parity']['pass'])")"
echo "submit_thread=0 headline (860 buffers): $(python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --submit-thread 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x'],1), round(d['ms_per_step'],4), d['parity']['pass'])")"
