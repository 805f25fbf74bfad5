#!/bin/bash
# Round 5 evidence: rocprofv3 kernel-trace stats of the default command with the final code, and how long the default command takes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p5; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p5
t0=$(date +%s); python bench.py > $O/bench_default_timed.json 2> /dev/null; t1=$(date +%s); echo "python bench.py: $((t1 - t0)) s wall" | tee $O/default_command_wall_seconds.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_default -- python3 $R/bench.py --no-cpu-baseline --no-second-form --no-one-second-leg > $O/st_default.log 2>&1); f=$(find $O/st_default -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_default.csv; rm -rf $O/st_default; head -6 $O/kernel_stats_default.csv | cut -c1-120
