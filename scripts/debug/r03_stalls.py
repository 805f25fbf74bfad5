#!/usr/bin/env python3
"""Reads a PBSO_TIMELINE=1 log: host-side stalls (a stage of the submission that took > 1 ms) and what they cost."""
import re
import sys
rows = [l for l in open(sys.argv[1]) if "pbso timeline" in l]
pat = re.compile(r"step (\d+) device: prep starts ([\d.]+) \| bank starts ([\d.]+) ends ([\d.]+) \| launch done ([\d.]+) \|\| host: step entered ([\d.-]+), prep submitted from ([\d.-]+) \(upload call returned ([\d.-]+)\), bank submitted at ([\d.-]+), all submitted ([\d.-]+)")
prev_done = None
n = 0
for l in rows:
    m = pat.search(l)
    if not m:
        continue
    step = int(m.group(1))
    p0, k0, k1, p1, h_enter, h_prep, h_copy, h_bank, h_done = (float(x) for x in m.groups()[1:])
    stages = {"plan": h_prep - h_enter, "upload": h_copy - h_prep, "prep launches": h_bank - h_copy, "bank launches": h_done - h_bank}
    if prev_done is not None:
        stages["between steps (caller)"] = h_enter - prev_done
    prev_done = h_done
    for k, v in stages.items():
        if v > 1.0:
            n += 1
            print(f"step {step}: {k} took {v:.3f} ms (host)")
print(f"{len(rows)} launches, {n} stalls > 1 ms")
