#!/usr/bin/env python3
"""start / end of the pbso kernels of the last steps of a profiled run (rocprofv3 --kernel-trace CSV): who overlaps whom"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:28], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows = sorted(r for r in rows if "pbso" in f or True)
pb = [r for r in rows if any(k in r[2] for k in ("iir_", "combine", "ffat", "project", "sum_parts", "copy_rows", "force_", "ar_"))]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
# the window around the LONGEST oscillator-bank launch: the measured loop (bench.py's later legs run smaller engines)
banks = [i for i, r in enumerate(pb) if "iir_block" in r[2] or "iir_pipe" in r[2] or "iir_bank" in r[2]]
mid = max(banks, key=lambda i: pb[i][1] - pb[i][0]) if banks else len(pb) - n // 2
lo = max(0, min(mid - n // 2, len(pb) - n))
t0 = pb[lo][0]
for s, e, name, q in pb[lo:lo + n]:
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  (+{(e - s) / 1e3:7.1f})  q{q}  {name}")
