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

# latency_path = -1: every launch prepared on the preparation stream (round 3's behaviour), for comparison
for n_modes, lp in ((128, 0), (512, 0), (2048, 0), (512, -1)):
    eng = Engine(qnorm=capi.QNORM_ALL, latency_path=lp)
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
    # ... and the facade's delivery: pbso_step_to_host into a pinned buffer (the bank's own stores), pbso_host_wait
    hb = eng.host_buffer(1)
    ts_host = []
    for i in range(300):
        if i % 7 == 0:
            eng.enqueue_force(0, ForceMessage(data=rng.standard_normal(n_modes) * 1e-3))
        t0 = time.perf_counter()
        eng.step_to_host(1, hb)
        eng.host_wait()
        ts_host.append(time.perf_counter() - t0)
    ts_host = np.array(ts_host[50:]) * 1e6
    info = eng.info()
    print(f"modes={n_modes:5d} latency_path={lp:2d} R={info['modes_per_lane']} W={info['waves_per_object']}: step+sync median {np.median(ts):7.1f} us "
          f"p99 {np.percentile(ts, 99):7.1f} us; with audio D2H median {np.median(ts_read):7.1f} us p99 {np.percentile(ts_read, 99):7.1f} us; "
          f"step_to_host + host_wait (samples in pinned host memory) median {np.median(ts_host):7.1f} us p99 {np.percentile(ts_host, 99):7.1f} us; "
          f"kernel {info['last_step_kernel_ms'] * 1e3:6.1f} us; one-stream launches {info['total_one_stream_launches']} of {info['total_steps']}  (deadline 11 630 us)")
    eng.close()


# Sustained contact in real-time mode (the facade's use: ONE buffer per step, an AutoregressiveForce alive, a new face hit every
# buffer: tools/real_time_modal_sound.cpp:754-776, 1127-1160): the force-profile kernels (K2) and the oscillator bank of every
# buffer are on the critical path here -- nothing runs a step ahead.
# (fuse: pbso_engine_desc::fuse_short_launches -- 0 the policy of round 6: the three profile kernels as one, projection and scatter inside
#  the combine kernel; -1 the seven launches of rounds 3 - 5.  data: a face hit projected on the device / explicit modal data, the facade's)
for n_modes, lp, fuse, explicit in ((512, 0, 0, False), (2048, 0, 0, False), (512, 0, -1, False), (512, 0, 0, True), (512, 0, -1, True), (512, -1, 0, False)):
    eng = Engine(qnorm=capi.QNORM_ALL, latency_path=lp, fuse_short_launches=fuse)
    shapes = synth.mode_shapes(n_modes, 6)
    eng.add_object(synth.eigenvalues(n_modes, 6), synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes)
    eng.finalize()
    eng.set_use_transfer(0, False)
    assert eng.enqueue_force(0, ForceMessage(forceType=capi.AUTOREGRESSIVE_FORCE, sustainedForceStart=True), 0)
    vns = synth.unit_normals(400, 6)
    drng = np.random.default_rng(1)
    ts, ks, ds = [], [], []
    for i in range(400):
        if i > 0 and explicit:
            eng.enqueue_force(0, ForceMessage(data=drng.standard_normal(n_modes) * 1e-3, forceType=capi.AUTOREGRESSIVE_FORCE), i)
        elif i > 0:
            eng.enqueue_force(0, ForceMessage(vids=[0, 1, 2], coords=[0.2, 0.3, 0.5], vn=vns[i], forceType=capi.AUTOREGRESSIVE_FORCE), i)
        t0 = time.perf_counter()
        eng.step(1)
        eng.sync()
        ts.append(time.perf_counter() - t0)
        info = eng.info()
        ks.append(info["last_step_kernel_ms"] * 1e3)
        ds.append(info["last_step_device_ms"] * 1e3)
    ts = np.array(ts[50:]) * 1e6
    print(f"sustained AR scraping, modes={n_modes:5d} latency_path={lp:2d} fuse_short_launches={fuse:2d} {'explicit data' if explicit else 'face hit     '}: step+sync median {np.median(ts):7.1f} us p99 {np.percentile(ts, 99):7.1f} us; "
          f"device pipeline (K2 + projection + combine + bank) median {np.median(ds[50:]):6.1f} us, bank alone {np.median(ks[50:]):6.1f} us  (deadline 11 630 us)")
    eng.close()
