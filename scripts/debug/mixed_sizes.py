#!/usr/bin/env python3
"""K1 time of engines mixing object sizes (one launch per team-size class, back to back)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402


def run(sizes, label):
    eng = Engine(qnorm=capi.QNORM_ALL)
    rng = np.random.default_rng(0)
    for i, m in enumerate(sizes):
        eng.add_object(synth.eigenvalues(m, 100 + i), synth.RHO, synth.ALPHA, synth.BETA)
    eng.finalize()
    for i, m in enumerate(sizes):
        eng.set_use_transfer(i, False)
        eng.enqueue_force(i, ForceMessage(data=rng.standard_normal(m) * 1e-3))
    for _ in range(12):
        eng.step(86)
    eng.sync()
    i0 = eng.info()
    for _ in range(10):
        eng.step(86)
    eng.sync()
    i1 = eng.info()
    k = (i1["total_kernel_ms"] - i0["total_kernel_ms"]) / 10
    print(f"{label:44s} modes {sum(sizes):7d} teams {i1['n_teams']:5d} R={i1['modes_per_lane']} K1 {k:.3f} ms  = {k * 1e6 / sum(sizes):.2f} ns/mode")
    eng.close()


run([512] * 1024, "1024 x 512")
run([512] * 512 + [64] * 4096, "512 x 512 + 4096 x 64")
run([2048] * 128 + [128] * 2048, "128 x 2048 + 2048 x 128")
run([1024] * 256 + [300] * 600 + [50] * 1000, "256 x 1024 + 600 x 300 + 1000 x 50")
run([512] * 1900 + [4096] * 10, "1900 x 512 + 10 x 4096")
run([512] * 512 + [64] * 4096 + [2000] * 6, "512 x 512 + 4096 x 64 + 6 x 2000")
