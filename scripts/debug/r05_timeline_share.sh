#!/bin/bash
# kernel timeline of a strong-scaling share at ten-second steps: what sits between two banks
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
o=${1:-128}; sk=${2:-0}
rm -rf /tmp/tlshare; (cd /tmp && PBSO_ENGINE_OPTS=scan_kernel=$sk rocprofv3 --kernel-trace --output-format csv -d /tmp/tlshare -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-form --no-parity --no-strong-share --no-one-second-leg --objects $o --steps 12 --warmup 3 > /dev/null 2>&1)
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/tlshare/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:30], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f)))
banks = [i for i, r in enumerate(rows) if "iir_block" in r[2]]
mid = banks[min(len(banks) - 1, 9)]          # (the main timed region: 3 warm-up + 12 steps come first, the side legs after)
lo = max(0, mid - 8)
t0 = rows[lo][0]
for s, e, name, q in rows[lo:lo + 30]:
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  (+{(e - s) / 1e3:7.1f})  q{q}  {name}")
PY
