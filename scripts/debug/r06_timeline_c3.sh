#!/bin/bash
# kernel + copy timeline of the 64 x 256 listener scene at 86 buffers per step (BASELINE configs[2]): what runs when on the device
# over three consecutive steps in the middle of the timed region
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
ST=${1:-0}
rm -rf /tmp/tlc3
(cd /tmp && PBSO_TIMING_EVERY=0 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tlc3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-form --no-strong-share \
   --no-one-second-leg --steps 60 --warmup 5 --buffers 86 --objects 64 --modes 256 --scenario listener --submit-thread $ST > /tmp/tlc3.json 2>/dev/null)
python3 -c "import json; d=json.loads(open('/tmp/tlc3.json').read().strip().splitlines()[-1]); print('under the tracer: ms_per_step', round(d['ms_per_step'],4))"
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/tlc3/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:40] + " q" + r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
for g in glob.glob("/tmp/tlc3/**/*memory_copy_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")) for r in csv.DictReader(open(g))]
rows.sort()
banks = [i for i, r in enumerate(rows) if "iir_block" in r[2]]
n = len(banks)
lo = banks[n // 2] + 1           # (the run's last launches are bench.py's side legs: host delivery, the object mix)
hi = banks[n // 2 + 3] + 3
t0 = rows[lo][0]
for s, e, name in rows[lo:hi]:
    print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  (+{(e - s) / 1e3:6.1f})  {name}")
PY
