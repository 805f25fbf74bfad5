#!/usr/bin/env python3
"""computeTransfer(pos, T*) for a whole HUD sphere in one call
(tools/real_time_modal_sound.cpp:916-927 evaluates 10 242 sphere vertices x M modes, one
position at a time).  Wall time of pbso_compute_transfer_batch including the device-to-host
copy of the result; the kernel alone is in the rocprofv3 stats taken beside it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpbso_amd import Engine, synth   # noqa: E402

n_pos = 10242
rng = np.random.default_rng(4)
v = rng.standard_normal((n_pos, 3))
pos = 0.5 * v / np.linalg.norm(v, axis=1, keepdims=True)
for n_modes in (64, 256, 1024):
    lam = synth.eigenvalues(n_modes, 9)
    maps = synth.ffat_maps(lam, 9)
    eng = Engine()
    eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA)
    eng.set_ffat_maps(0, maps)
    eng.finalize()
    ts, tf = [], []
    keep = np.zeros((n_pos, n_modes))
    for _ in range(12):
        t0 = time.perf_counter()
        ok, out = eng.compute_transfer_batch(0, pos, n_modes, out=keep)
        ts.append(time.perf_counter() - t0)
    for _ in range(4):
        t0 = time.perf_counter()
        ok2, fresh = eng.compute_transfer_batch(0, pos, n_modes)           # a new array per call: its first touch is in the time
        tf.append(time.perf_counter() - t0)
    assert ok and ok2 and np.isfinite(out).all() and (out > 0).all() and np.array_equal(out, fresh)
    t = np.median(ts[2:])
    look = n_pos * n_modes
    print(f"modes={n_modes:5d}: {n_pos} positions -> {look / 1e6:.2f} M lookups, result {out.nbytes / 1e6:.1f} MB: "
          f"median {t * 1e3:.2f} ms per call = {look / t / 1e9:.2f} G lookups/s incl. D2H ({out.nbytes / t / 1e9:.1f} GB/s of results); "
          f"into a fresh array {np.median(tf) * 1e3:.2f} ms")
    eng.close()
