#!/bin/bash
# 128 x 512 x 860: one run in five steps 2.0 - 2.6 ms instead of 1.3 -- trace runs until one shows, print what sits in its long gaps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
rm -rf /tmp/tlo; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tlo -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-form --no-parity --no-strong-share --no-one-second-leg --objects 128 --steps 16 --warmup 3 > /tmp/tlo.json 2>/dev/null)
python3 - $i <<'PY'
import csv, glob, json, sys
d = json.loads(open("/tmp/tlo.json").read().strip().splitlines()[-1])
f = sorted(glob.glob("/tmp/tlo/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:30], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f)))
banks = [r for r in rows if "iir_block" in r[2] and (r[1] - r[0]) > 800e3]
gaps = [(banks[k + 1][0] - banks[k][1]) / 1e3 for k in range(len(banks) - 1)]
print(f"run {sys.argv[1]}: ms_per_step {d['ms_per_step']:.3f}; {len(banks)} long banks; gaps between them us: median {sorted(gaps)[len(gaps) // 2]:.0f} max {max(gaps):.0f}; banks us: min {min((b[1] - b[0]) / 1e3 for b in banks):.0f} max {max((b[1] - b[0]) / 1e3 for b in banks):.0f}")
inner = [g for g in gaps if g < 20e3]
print("   gaps inside the legs, largest:", [round(g) for g in sorted(inner)[-6:]])
if d["ms_per_step"] > 1.5:
    k = max((j for j in range(len(gaps)) if gaps[j] < 20e3), key=lambda j: gaps[j])
    t0 = banks[k][0]
    for s, e, name, q in rows:
        if banks[k][0] - 50e3 <= s <= banks[k + 1][1] + 50e3:
            print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  (+{(e - s) / 1e3:7.1f})  q{q}  {name}")
    sys.exit(7)
PY
[ $? = 7 ] && break
done
