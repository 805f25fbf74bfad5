"""device-to-host delivery: pbso_step_to_host serial vs pipelined, against a plain torch pinned copy"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from openpbso_amd import Engine, capi, synth
n_obj, n_modes, nb = 1024, 512, 86
eng = Engine(qnorm=capi.QNORM_ALL, timing_every=4)
for i in range(n_obj):
    eng.add_object(synth.eigenvalues(n_modes, 10 + (i % 7)), synth.RHO, synth.ALPHA, synth.BETA)
eng.finalize()
for i in range(n_obj):
    eng.set_use_transfer(i, False)
hb = [eng.host_buffer(nb), eng.host_buffer(nb)]
for k in range(3):
    eng.step_to_host(nb, hb[k % 2]); eng.host_wait()
t0 = time.perf_counter()
for k in range(6):
    eng.step_to_host(nb, hb[k % 2]); eng.host_wait()
print("serial (step + copy, waited each):", (time.perf_counter() - t0) / 6 * 1e3, "ms")
t0 = time.perf_counter()
calls = []
for k in range(10):
    tc = time.perf_counter()
    eng.step_to_host(nb, hb[k % 2])
    calls.append((time.perf_counter() - tc) * 1e3)
eng.host_wait()
print("pipelined:", (time.perf_counter() - t0) / 10 * 1e3, "ms per step; host time of each call:", [round(c, 2) for c in calls])
# plain copies of the same size
dev = torch.empty((n_obj, nb * 513), dtype=torch.float32, device="cuda")
pin = torch.empty((n_obj, nb * 513), dtype=torch.float32, pin_memory=True)
torch.cuda.synchronize()
for _ in range(2):
    pin.copy_(dev, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    pin.copy_(dev, non_blocking=True); torch.cuda.synchronize()
print("torch pinned copy:", (time.perf_counter() - t0) / 5 * 1e3, "ms")
mine = torch.from_numpy(hb[0])
t0 = time.perf_counter()
for _ in range(5):
    mine.copy_(dev, non_blocking=False); torch.cuda.synchronize()
print("torch copy into pbso_host_alloc memory:", (time.perf_counter() - t0) / 5 * 1e3, "ms")
# zero-copy: the bank writes its audio straight into the pinned host buffer (device-accessible)
for k in range(2):
    eng.step(nb, into=hb[k % 2].ctypes.data)
eng.sync()
t0 = time.perf_counter()
for k in range(10):
    eng.step(nb, into=hb[k % 2].ctypes.data)
eng.sync()
print("bank writing to pinned host memory directly:", (time.perf_counter() - t0) / 10 * 1e3, "ms per step", "finite", bool(np.isfinite(hb[0]).all()))
