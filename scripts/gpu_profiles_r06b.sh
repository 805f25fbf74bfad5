#!/bin/bash
# Round 6 evidence, part 2 (gpurun_out/p6/, copied to profiles/r06_* by scripts/collect_profiles_r06.py): the bench lines of the other
# BASELINE configurations and forms with the final build, the per-rank shares alone, the N > 1 path on one GPU, kernel-trace stats of
# configs[2] and configs[4].
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p6; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p6
(rocminfo | grep -E "Marketing Name|gfx9" | sort | uniq -c | head -4; lscpu | grep -E "Model name|^CPU\(s\)"; echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; free -g | head -2) > $O/env.txt 2>&1
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"; }
b form_block_bf16 --no-cpu-baseline --form block_bf16 --no-strong-share
b qnorm_off --no-cpu-baseline --qnorm off --no-second-form --no-strong-share
b host_delivery --host-delivery --no-cpu-baseline --no-second-form --no-strong-share
b c2_1x512 --no-cpu-baseline --objects 1 --modes 512 --buffers 86 --steps 200 --warmup 5
b c3_64x256_listener --no-cpu-baseline --objects 64 --modes 256 --scenario listener --buffers 86 --steps 200 --warmup 5
b c5_8x4096_scraping --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 200 --warmup 5
b c5_8x4096_scraping_qnorm_off --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 200 --warmup 5
b c2_1x512_10s_steps --no-cpu-baseline --objects 1 --modes 512
b c3_64x256_listener_10s_steps --no-cpu-baseline --objects 64 --modes 256 --scenario listener
b c5_8x4096_scraping_10s_steps --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping
b c5_8x4096_scraping_qnorm_off_10s_steps --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off
for o in 512 256 128; do
b share_${o}x512 --no-cpu-baseline --no-second-form --objects $o --buffers 86 --steps 200 --warmup 5
done
# (1024 x 512 sustained scraping, one-second steps -- mostly dense, chip-filling: the policy's path against the walk)
b c4scr_1024x512_scraping --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --scenario scraping --buffers 86 --steps 10 --warmup 2
PBSO_TIME_CHUNKS=-1 b c4scr_1024x512_scraping_walk --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --scenario scraping --buffers 86 --steps 10 --warmup 2
PBSO_BENCH_BACKEND=gloo b 2ranks_one_gpu_gloo --no-cpu-baseline --gpus 2 --steps 20 --warmup 2
(PBSO_BENCH_GATHER_SELF=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 1 --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_1rank_torchrun_device_group_selfgather.json 2> $O/bench_1rank.err; echo "selfgather rc=$?")
st() { name=$1; shift; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -- python3 $R/bench.py --no-cpu-baseline --no-second-form "$@" > $O/st_$name.log 2>&1); f=$(find $O/st_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$name.csv; rm -rf $O/st_$name; echo "stats $name: $(sed -n 2p $O/kernel_stats_$name.csv | cut -c1-70 | tr -d '\n')"; }
st c3_64x256_listener --objects 64 --modes 256 --scenario listener --buffers 86 --steps 40
st c5_8x4096_scraping --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40
st c5_8x4096_scraping_qnorm_off --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40
PBSO_CENSUS=1 python scripts/census.py 1024 512 > $O/census_1024x512_block_f32.txt 2>&1
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/p6/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):62s} rt {d['realtime_x']:9.1f}  ms/step {d['ms_per_step']:8.4f}  bank {d['roofline']['kernel_ms']:7.4f}  frac {d['roofline']['frac']:.3f}  err {d.get('max_err')}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", repr(ex)[:80])
PY
