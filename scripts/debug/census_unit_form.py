#!/usr/bin/env python3
"""How many of consumer 0's qnorm chain groups took the unit-force form (census word 11) in the bench's scraping scene."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PBSO_CENSUS"] = "1"
import numpy as np                                                  # noqa: E402
from openpbso_amd import Engine, ForceMessage, synth, capi        # noqa: E402
n_obj, M, nb = 8, 4096, 86
eng = Engine(qnorm=capi.QNORM_ALL, form=capi.FORM_BLOCK)
for i in range(n_obj):
    seed = synth.seed_for(5, i)
    eng.add_object(synth.eigenvalues(M, seed), synth.RHO, synth.ALPHA, synth.BETA, M, synth.mode_shapes(M, seed))
eng.finalize()
rng = np.random.default_rng(1)
for i in range(n_obj):
    eng.set_use_transfer(i, False)
    eng.enqueue_force(i, ForceMessage(forceType=capi.AUTOREGRESSIVE_FORCE, sustainedForceStart=True), 0)
    vns = synth.unit_normals(2 * nb, i)
    for b in range(1, 2 * nb):
        bary = rng.random(3)
        eng.enqueue_force(i, ForceMessage(forceType=capi.AUTOREGRESSIVE_FORCE, vids=rng.integers(0, synth.N_VERTS, 3), coords=bary / bary.sum(), vn=vns[b]), b)
for _ in range(2):
    eng.step(nb)
eng.sync()
c = eng.census(n_obj * M // 64)
print("unit-form groups per team over the last launch (of", nb, "dense buffers):", np.unique(c[:, 11], return_counts=True))
