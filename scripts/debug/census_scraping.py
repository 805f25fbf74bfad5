#!/usr/bin/env python3
"""PBSO_CENSUS=1 on the 8 x 4096 sustained-scraping scene (BASELINE configs[4]) with K1b pinned (PBSO_SPLIT=0: one wave per 64 modes,
the forced block path): where wave 0 of every team spends its cycles.  (The engine's own choice for this scene is the time-split
kernel: scripts/debug/census_split.py.)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PBSO_CENSUS"] = "1"
os.environ["PBSO_SPLIT"] = "0"
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402
n_obj, M, nb = 8, 4096, 86
qn = capi.QNORM_OFF if len(sys.argv) > 1 and sys.argv[1] == "off" else capi.QNORM_ALL
eng = Engine(qnorm=qn, form=capi.FORM_BLOCK)
rng = np.random.default_rng(0)
for i in range(n_obj):
    eng.add_object(synth.eigenvalues(M, 100 + i), synth.RHO, synth.ALPHA, synth.BETA)
eng.finalize()
for i in range(n_obj):
    eng.set_use_transfer(i, False)
    eng.enqueue_force(i, ForceMessage(forceType=capi.AUTOREGRESSIVE_FORCE, sustainedForceStart=True), 0)
    for b in range(1, 3 * nb):
        eng.enqueue_force(i, ForceMessage(data=rng.standard_normal(M) * 1e-3, forceType=capi.AUTOREGRESSIVE_FORCE), b)
for _ in range(3):
    eng.step(nb)
eng.sync()
info = eng.info()
c = eng.census()
t0, t1 = c[:, 0].astype(np.int64), c[:, 1].astype(np.int64)
clk = (c[:, 5].astype(np.int64) - c[:, 4].astype(np.int64)) / np.maximum(t1 - t0, 1) * 100.0
print(f"R={info['modes_per_lane']} W={info['waves_per_object']} teams={info['n_teams']} kernel_ms={info['last_step_kernel_ms']:.3f} clock median {np.median(clk):.0f} MHz; "
      f"WG duration us median {np.median((t1 - t0) / 100.0):.0f} max {((t1 - t0) / 100.0).max():.0f}")
names = {6: "head", 7: "matrix (MFMA + operand reads)", 8: "barrier", 9: "combine", 10: "sample 0 + taps", 11: "per-sample stepping"}
tot = c[:, 6:12].astype(np.float64).sum(axis=1)
for k, name in names.items():
    v = c[:, k].astype(np.float64)
    print(f"  {name:32s} median {np.median(v) / nb:8.0f} cycles per buffer ({np.median(v / tot) * 100:5.1f} %)")
print(f"  total {np.median(tot) / nb:.0f} cycles per buffer")
