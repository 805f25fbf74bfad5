cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06ff
timeout 900 python -m pytest tests/test_gpu_ffat_shared.py tests/test_gpu_parity.py tests/test_gpu_listener_mix_edges.py tests/test_gpu_fullsize.py -q -k "ffat or listener or transfer or config3" > gpurun_out/r06ff/tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r06ff/tests.log
bash scripts/debug/r06_timeline_c3.sh 1
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --steps 200 --warmup 5 --buffers 86 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing']; print(round(d['realtime_x'],1), round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'plan', round(t['host_plan_ms'],4), 'enq', round(t['host_enqueue_ms'],4), d['parity']['pass'])"; }
for rep in 1 2; do for st in 0 1; do
echo "submit_thread=$st c3 64x256 listener:  $(run --objects 64 --modes 256 --scenario listener --submit-thread $st)"
done; done
echo "PBSO_FFAT_SHARED=0 st=0:  $(PBSO_FFAT_SHARED=0 run --objects 64 --modes 256 --scenario listener --submit-thread 0)"
