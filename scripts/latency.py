#!/usr/bin/env python3
"""Real-time mode: wall time of ONE ModalSolver::step() = one 513-sample
buffer for one object, including the device-to-host copy the audio callback needs.
Deadline in the reference: 11.63 ms per buffer (config.h:13-14)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402

for n_modes in (128, 512, 2048):
    eng = Engine(qnorm=capi.QNORM_ALL)
    eng.add_object(synth.eigenvalues(n_modes, 5), synth.RHO, synth.ALPHA, synth.BETA)
    eng.finalize()
    eng.set_use_transfer(0, False)
    rng = np.random.default_rng(0)
    ts, ts_read = [], []
    for i in range(300):
        if i % 7 == 0:
            eng.enqueue_force(0, ForceMessage(data=rng.standard_normal(n_modes) * 1e-3))
        t0 = time.perf_counter()
        eng.step(1)
        eng.sync()
        t1 = time.perf_counter()
        a = eng.audio()
        t2 = time.perf_counter()
        ts.append(t1 - t0)
        ts_read.append(t2 - t0)
    ts, ts_read = np.array(ts[50:]) * 1e6, np.array(ts_read[50:]) * 1e6
    info = eng.info()
    print(f"modes={n_modes:5d} R={info['modes_per_lane']} W={info['waves_per_object']}: step+sync median {np.median(ts):7.1f} us "
          f"p99 {np.percentile(ts, 99):7.1f} us; with audio D2H median {np.median(ts_read):7.1f} us p99 {np.percentile(ts_read, 99):7.1f} us; "
          f"kernel {info['last_step_kernel_ms'] * 1e3:6.1f} us  (deadline 11 630 us)")
    eng.close()
