#!/bin/bash
# Round 5 evidence, refresh after the last kernel changes (five-role order, segmented scan by square-and-multiply): the files named here
# replace those of scripts/gpu_profiles_r05.sh / _r05b.sh in gpurun_out/p5/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p5; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p5
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"; }
st() { name=$1; shift; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -- python3 $R/bench.py --no-cpu-baseline --no-second-form "$@" > $O/st_$name.log 2>&1); f=$(find $O/st_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$name.csv; rm -rf $O/st_$name; echo "stats $name done"; }
b default
b driver_flags_steps20_warmup5 --steps 20 --warmup 5
b c5_8x4096_scraping --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2
b c5_8x4096_scraping_qnorm_off --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40 --warmup 2
b c5_8x4096_scraping_10s_steps --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping
b c5_8x4096_scraping_qnorm_off_10s_steps --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off
PBSO_ENGINE_OPTS=time_chunks=-1,pipe_consumers=4 b c5_8x4096_scraping_five_role_teams --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2
st c5_8x4096_scraping_qnorm_off --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40
(PBSO_PIPE_CONSUMERS=4 python scripts/debug/census_split.py off; PBSO_PIPE_CONSUMERS=4 python scripts/debug/census_split.py; python scripts/debug/r05_pipe5_placement.py off) 2>&1 | grep -v amdgpu.ids > $O/census_8x4096_scraping_five_role_teams.txt
(echo "every build timed with the device otherwise idle (step, wait, step: scripts/debug/r05_scan_abl.sh <objects> <modes>; stop 1 / 2 / 3 = the serial kernel cut short after its stages, 9 = the product)"; bash scripts/debug/r05_scan_abl.sh 1 512; bash scripts/debug/r05_scan_abl.sh 128 512) 2>&1 | grep -E "every build|stop" > $O/scan_kernel_stages.txt
(for o in 128 512; do echo "== $o x 512 x 860, serial scan (scan_kernel = 1)"; bash scripts/debug/r05_timeline_share.sh $o 1 | tail -16; done; echo "== 128 x 512 x 860, the scan cut along the time axis forced (scan_kernel = 2)"; bash scripts/debug/r05_timeline_share.sh 128 2 | tail -16) > $O/timeline_share_860.txt 2>&1
(python scripts/debug/r05_census_share.py 1024; python scripts/debug/r05_census_share.py 128; python scripts/debug/r05_census_share.py 512) 2>&1 | grep -v amdgpu.ids > $O/census_walk_and_shares_860.txt
bash scripts/debug/r05_shares.sh > $O/shares_serial_vs_segmented_scan.txt 2>&1
ls $O | wc -l
