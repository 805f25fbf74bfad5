#!/bin/bash
# kernel timeline of configs[4] (8 x 4096 x 86 sustained scraping), with qnorm rows (cut in time) or without (five-role teams): what sits between two banks
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
q=${1:-sample}
rm -rf /tmp/tlc5; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tlc5 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-form --no-parity --no-strong-share --no-one-second-leg --clock-ramp-ms 0 --objects 8 --modes 4096 --scenario scraping --qnorm $q --buffers 86 --steps 30 --warmup 3 > /dev/null 2>&1)
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/tlc5/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:34], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f)))
banks = [i for i, r in enumerate(rows) if "iir_block" in r[2] or "iir_pipe" in r[2]]
mid = banks[min(len(banks) - 1, 50)]
lo = max(0, mid - 10)
t0 = rows[lo][0]
for s, e, name, q in rows[lo:lo + 36]:
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  (+{(e - s) / 1e3:7.1f})  q{q}  {name}")
PY
