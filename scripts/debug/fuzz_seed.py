#!/usr/bin/env python3
"""Re-runs one seed of tests/test_gpu_fuzz.py and prints where engine and oracle differ."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpbso_amd import synth, capi   # noqa: E402
from tests.scenarios import ObjSpec, run_engine, run_oracle   # noqa: E402
from tests.test_gpu_fuzz import random_script   # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
n_obj = int(rng.integers(1, 5))
n_modes = [int(rng.choice([3, 40, 64, 100, 129, 300])) for _ in range(n_obj)]
nb = int(rng.integers(6, 16))
with_maps = [bool(rng.random() < 0.5) for _ in range(n_obj)]
objs = []
for oi in range(n_obj):
    lam = synth.eigenvalues(n_modes[oi], 5000 + 10 * seed + oi)
    objs.append(ObjSpec(lam, maps=synth.ffat_maps(lam, 7000 + seed + oi, dim=4) if with_maps[oi] else None))
evs = random_script(rng, n_obj, n_modes, nb, with_maps)
split = None
if nb > 8:
    k = int(rng.integers(1, nb - 1))
    split = [k, nb - k]
print("n_modes", n_modes, "nb", nb, "maps", with_maps, "split", split)
B = 513
want = run_oracle(objs, evs, nb)
cfgs = [dict()] + [eval("dict(%s)" % a) for a in sys.argv[2:]]
for kw in cfgs:
    got = run_engine(objs, evs, nb, split=split if "split" not in kw else kw.pop("split"), **kw)
    print("engine", kw)
    for oi in range(n_obj):
        pk = np.abs(want["audio"][oi]).max()
        errs = [np.abs(got["audio"][oi][b * B:(b + 1) * B] - want["audio"][oi][b * B:(b + 1) * B]).max() / max(pk, 1e-300) for b in range(nb)]
        print(f"  obj {oi} ({n_modes[oi]} modes, maps {with_maps[oi]}): audio err/peak per buffer:", " ".join(f"{e:.1e}" for e in errs))
bad = int(sys.argv[1]) if False else None
for e in evs:
    print({k: (v if not isinstance(v, np.ndarray) else "arr") for k, v in e.items() if v is not None and v is not False})
