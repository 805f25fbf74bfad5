#!/usr/bin/env python3
"""PBSO_CENSUS=1 with the kernel of under-filled engines (K1p; PBSO_SPLIT_KERNEL=time: K1s) on a force-free / impulse scene (1 x 512 and 64 x 256): the two waves' cycles per buffer."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PBSO_CENSUS"] = "1"
os.environ["PBSO_SPLIT"] = "2"
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402
nb = 86
for n_obj, M in ((1, 512), (64, 256)):
    eng = Engine(qnorm=capi.QNORM_ALL, form=capi.FORM_BLOCK)
    rng = np.random.default_rng(0)
    for i in range(n_obj):
        eng.add_object(synth.eigenvalues(M, 100 + i), synth.RHO, synth.ALPHA, synth.BETA)
    eng.finalize()
    for i in range(n_obj):
        eng.set_use_transfer(i, False)
        for b in range(0, 3 * nb, 7):
            eng.enqueue_force(i, ForceMessage(data=rng.standard_normal(M) * 1e-3), b)
    for _ in range(3):
        eng.step(nb)
    eng.sync()
    info = eng.info()
    c = eng.census().astype(np.float64)
    print(f"{n_obj} x {M}: kernel_ms={info['last_step_kernel_ms']:.3f} split launches {info['total_split_launches']}; cycles per buffer (median over {c.shape[0]} teams)")
    pipe = os.environ.get("PBSO_SPLIT_KERNEL") != "time"
    if pipe:      # K1p: words 0..2 the producer, 6..8 consumer 0
        for w, role, names in ((0, "producer  ", ["head + sample 0", "stepping + parks", "wait at the barrier"]),
                               (1, "consumer 0", ["head + taps", "projection", "wait at the barrier"])):
            row = c[:, 6 * w:6 * w + 3]
            print(f"  {role}: " + "; ".join(f"{n} {np.median(row[:, k]) / nb:.0f}" for k, n in enumerate(names)) + f"; total {np.median(row.sum(axis=1)) / nb:.0f}")
    else:         # K1s: words 0..5 wave 0, 6..11 wave 1
        names = ["head + taps", "first stepping phase", "wait at A", "second phase", "wait at B", "projection"]
        for w in (0, 1):
            row = c[:, 6 * w:6 * w + 6]
            print(f"  wave {w}: " + "; ".join(f"{n} {np.median(row[:, k]) / nb:.0f}" for k, n in enumerate(names)) + f"; total {np.median(row.sum(axis=1)) / nb:.0f}")
    eng.close()
