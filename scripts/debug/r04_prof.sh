#!/bin/bash
# rocprofv3 kernel stats of one shape: r04_prof.sh <objects> <modes> [extra bench args]
cd "$GRAFT_REPO_ROOT"
o=$1; m=$2; shift 2
out=gpurun_out/r04/prof_${o}x${m}
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --objects $o --modes $m --steps 20 --warmup 5 --no-cpu-baseline --no-second-form --no-parity "$@" > $out/bench.json 2> $out/bench.err
python3 - $out <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), 'avg_us', round(float(r['AverageNs'])/1e3,2), 'min', round(float(r['MinNs'])/1e3,2), 'max', round(float(r['MaxNs'])/1e3,2), 'pct', r['Percentage'])
PY
