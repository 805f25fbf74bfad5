#!/bin/bash
# kernel trace of the 8 x 4096 scraping scene: which kernels, how long, in what order
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
Q="$1"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5prof$Q -o c5 -- python3 bench.py --buffers 86 --steps 30 --warmup 3 --no-cpu-baseline --no-second-form --no-strong-share --no-parity --objects 8 --modes 4096 --scenario scraping $Q > /dev/null 2>&1
f=$(find /tmp/c5prof$Q -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %5s avg %9.1f us min %8.1f max %8.1f  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
