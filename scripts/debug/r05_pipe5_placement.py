#!/usr/bin/env python3
"""where do the five-role teams' waves sit?  (PBSO_CENSUS=1, 8 x 4096 scraping on the pipeline kernel: words 3, 9, 10 = HW_ID | XCC_ID << 32 of P, A, B)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PBSO_CENSUS"] = "1"
os.environ["PBSO_SPLIT"] = "2"
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402
n_obj, M, nb = 8, 4096, 86
eng = Engine(qnorm=capi.QNORM_OFF if "off" in sys.argv else capi.QNORM_ALL, form=capi.FORM_BLOCK)
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
c = eng.census(512)
def dec(w):
    hw = int(w) & 0xFFFFFFFF
    return dict(simd=(hw >> 4) & 3, cu=(hw >> 8) & 0xF, sh=(hw >> 12) & 1, se=(hw >> 13) & 7, xcc=(int(w) >> 32) & 15, wave=hw & 0xF)
for tix in (0, 1, 2, 3, 100, 101):
    print(tix, "P", dec(c[tix, 3]), "A", dec(c[tix, 9]), "B", dec(c[tix, 10]))
import collections
print("P simd histogram", collections.Counter(dec(w)["simd"] for w in c[:, 3]))
print("A simd histogram", collections.Counter(dec(w)["simd"] for w in c[:, 9]))
print("B simd histogram", collections.Counter(dec(w)["simd"] for w in c[:, 10]))
same = sum(1 for i in range(0, 512, 2) if dec(c[i, 3])["simd"] == dec(c[i + 1, 3])["simd"])
print("pairs of teams whose P share a SIMD:", same, "of 256")
