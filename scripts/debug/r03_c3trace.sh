#!/bin/bash
# kernel timeline of configs[2] (64 x 256, a listener move per buffer): where a step's 0.44 ms goes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c3; export TMPDIR=/tmp; R=$PWD
ARGS="${ARGS:---objects 64 --modes 256 --scenario listener}"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c3/tr -- python3 $R/bench.py --no-cpu-baseline --no-parity --no-second-form $ARGS --steps 12 --warmup 2 --settle 2 > $R/gpurun_out/c3/log.txt 2>&1)
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/c3/tr/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f)))
k1 = [i for i, r in enumerate(rows) if "iir_" in r[2]]
lo = rows[k1[-6]][0]; hi = rows[k1[-3]][1]
t0 = lo
for s, e, n, q in rows:
    if lo <= s <= hi: print(f"{(s - t0) / 1e3:9.1f} us +{(e - s) / 1e3:7.1f} us  q{q:>4s}  {n}")
PY
tail -2 gpurun_out/c3/log.txt | cut -c1-300; rm -rf gpurun_out/c3/tr
