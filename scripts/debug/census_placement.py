#!/usr/bin/env python3
"""Where the three waves of the pipeline kernel's teams sit (PBSO_CENSUS=1: HW_ID of producer / consumer 0 / consumer 1 in census words
3 / 9 / 10): SIMD of each role, and for the CUs that hold two teams, which roles share a SIMD."""
import collections
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PBSO_CENSUS"] = "1"
os.environ["PBSO_SPLIT"] = "2"
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402
n_obj, M, nb = 8, 4096, 86
eng = Engine(qnorm=capi.QNORM_ALL, form=capi.FORM_BLOCK)
rng = np.random.default_rng(0)
for i in range(n_obj):
    eng.add_object(synth.eigenvalues(M, 100 + i), synth.RHO, synth.ALPHA, synth.BETA)
eng.finalize()
for i in range(n_obj):
    eng.set_use_transfer(i, False)
    eng.enqueue_force(i, ForceMessage(data=rng.standard_normal(M) * 1e-3), 0)
eng.step(nb)
eng.sync()
c = eng.census(n_obj * M // 64)
hw = {r: c[:, w].astype(np.uint64) for r, w in (("P", 3), ("C0", 9), ("C1", 10))}
simd = {r: ((v >> np.uint64(4)) & np.uint64(3)).astype(int) for r, v in hw.items()}
cu = {r: (((v >> np.uint64(32)) << np.uint64(8)) | ((v >> np.uint64(8)) & np.uint64(0xFF))).astype(int) for r, v in hw.items()}
print("teams", c.shape[0])
print("SIMD triples (P, C0, C1):", collections.Counter(zip(simd["P"], simd["C0"], simd["C1"])).most_common(8))
by_cu = collections.defaultdict(list)
for t in range(c.shape[0]):
    by_cu[cu["P"][t]].append(t)
share = collections.Counter()
for k, ts in by_cu.items():
    occ = collections.defaultdict(list)
    for t in ts:
        for r in ("P", "C0", "C1"):
            occ[simd[r][t]].append(r[0])
    for sid, roles in occ.items():
        share["".join(sorted(roles))] += 1
print("teams per CU:", collections.Counter(len(v) for v in by_cu.values()))
print("roles per SIMD (over the CUs seen):", share.most_common())
