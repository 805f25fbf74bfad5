#!/usr/bin/env python3
"""Timeline of one profiled bench run: per-kernel start/end and the gaps between
consecutive oscillator-bank launches (rocprofv3 --kernel-trace CSV)."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
t0 = rows[0][0]
k1 = [r for r in rows if "iir_bank" in r[2] or "iir_block" in r[2]]
print("all kernels in the last two steps:")
if len(k1) >= 3:
    lo = k1[-3][1]
    for s, e, n in rows:
        if s >= lo:
            print(f"  {(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  {n}")
print("K1 launches: duration, gap to next")
for a, b in zip(k1, k1[1:]):
    print(f"  dur {(a[1] - a[0]) / 1e3:8.1f} us   gap {(b[0] - a[1]) / 1e3:8.1f} us")
