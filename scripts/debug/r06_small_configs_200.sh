set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p6; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p6
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"; }
for rep in 1 2; do
b c2_1x512 --no-cpu-baseline --objects 1 --modes 512 --buffers 86 --steps 200 --warmup 5
python -c "import json; d=json.loads(open('gpurun_out/p6/bench_c2_1x512.json').read().strip().splitlines()[-1]); print('c2', d['realtime_x'], d['ms_per_step'], d['timing']['host_submit_ms'])"
done
b c3_64x256_listener --no-cpu-baseline --objects 64 --modes 256 --scenario listener --buffers 86 --steps 200 --warmup 5
b c5_8x4096_scraping --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 200 --warmup 5
b c5_8x4096_scraping_qnorm_off --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 200 --warmup 5
for o in 512 256 128; do
b share_${o}x512 --no-cpu-baseline --no-second-form --objects $o --buffers 86 --steps 200 --warmup 5
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/p6/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):62s} rt {d['realtime_x']:9.1f}  ms/step {d['ms_per_step']:8.4f}  bank {d['roofline']['kernel_ms']:7.4f}  steps {d['steps']}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", repr(ex)[:80])
PY
