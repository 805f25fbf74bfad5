#!/bin/bash
# timeline of the measured loop of a share: bash scripts/debug/r04_tl.sh <objects> [rows]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/tl; mkdir -p $O
n=${1:-128}; rows=${2:-30}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$n -- python3 $R/bench.py --no-cpu-baseline --no-second-form --no-parity --no-strong-share --objects $n --steps 20 --warmup 3 > $O/tl_$n.log 2>&1)
python scripts/debug/r04_timeline.py $O/tl_$n $rows | tee $O/timeline_$n.txt; rm -rf $O/tl_$n
