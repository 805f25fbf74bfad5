#!/bin/bash
# Round 5 evidence, refresh after the scan's rebuild (pipelined batches; the scan cut along the time axis on the same body, by policy
# for long chunks) and the planner pool's claimed shares: the files named here replace those of the earlier r05 scripts in gpurun_out/p5/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p5; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p5
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"; }
st() { name=$1; shift; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -- python3 $R/bench.py --no-cpu-baseline --no-second-form "$@" > $O/st_$name.log 2>&1); f=$(find $O/st_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$name.csv; rm -rf $O/st_$name; echo "stats $name done"; }
b default
b driver_flags_steps20_warmup5 --steps 20 --warmup 5
st default --no-one-second-leg
b c2_1x512 --no-cpu-baseline --objects 1 --modes 512 --buffers 86 --steps 40 --warmup 2
b c3_64x256_listener --no-cpu-baseline --objects 64 --modes 256 --scenario listener --buffers 86 --steps 40 --warmup 2
b c5_8x4096_scraping --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2
b c5_8x4096_scraping_qnorm_off --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40 --warmup 2
for o in 512 256 128; do
b share_${o}x512 --no-cpu-baseline --no-second-form --objects $o --buffers 86 --steps 40 --warmup 3
b share_${o}x512_10s_steps --no-cpu-baseline --no-second-form --no-one-second-leg --objects $o
done
st share_128x512_10s_steps --objects 128 --no-one-second-leg --no-strong-share
(echo "every build timed with the device otherwise idle (step, wait, step: scripts/debug/r05_scan_abl.sh <objects> <modes>; stop 1 / 2 / 3 = the serial kernel cut short after its stages, 9 = the product)"; bash scripts/debug/r05_scan_abl.sh 1 512; bash scripts/debug/r05_scan_abl.sh 128 512) 2>&1 | grep -E "every build|stop" > $O/scan_kernel_stages.txt
(for o in 128 256; do echo "== $o x 512 x 860, policy (the scan cut along the time axis)"; bash scripts/debug/r05_timeline_share.sh $o 0 | tail -22; done; echo "== 128 x 512 x 860, serial scan forced (scan_kernel = 1)"; bash scripts/debug/r05_timeline_share.sh 128 1 | tail -22; echo "== 512 x 512 x 860, policy (serial scan)"; bash scripts/debug/r05_timeline_share.sh 512 0 | tail -22) > $O/timeline_share_860.txt 2>&1
bash scripts/debug/r05_shares.sh > $O/shares_serial_vs_segmented_scan.txt 2>&1
st c5_8x4096_scraping --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40
st c5_8x4096_scraping_qnorm_off --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40
(echo "== with qnorm rows (cut in time, the preparation forked)"; bash scripts/debug/r05_timeline_c5.sh sample | head -30; echo "== the same, one preparation stream (PBSO_PREP_SPLIT=0)"; PBSO_PREP_SPLIT=0 bash scripts/debug/r05_timeline_c5.sh sample | head -24; echo "== without qnorm rows (five-role teams)"; bash scripts/debug/r05_timeline_c5.sh off | head -24) > $O/timeline_c5_8x4096_scraping.txt 2>&1
b c5_8x4096_scraping_10s_steps --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping
b c5_8x4096_scraping_qnorm_off_10s_steps --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off
b c2_1x512_10s_steps --no-cpu-baseline --objects 1 --modes 512
b c3_64x256_listener_10s_steps --no-cpu-baseline --objects 64 --modes 256 --scenario listener
b qnorm_off --no-cpu-baseline --qnorm off --no-second-form --no-strong-share
b host_delivery --host-delivery --no-cpu-baseline --no-second-form --no-strong-share
timeout 300 python scripts/latency.py > $O/realtime_latency.txt 2>&1
PBSO_BENCH_BACKEND=gloo b 2ranks_one_gpu_gloo --no-cpu-baseline --gpus 2 --steps 20 --warmup 2
(PBSO_BENCH_GATHER_SELF=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 1 --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_1rank_torchrun_device_group_selfgather.json 2> $O/bench_1rank.err; echo "selfgather rc=$?")
ls $O | wc -l
