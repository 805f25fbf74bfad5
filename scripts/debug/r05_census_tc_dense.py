#!/usr/bin/env python3
"""PBSO_CENSUS=1 on the 8 x 4096 sustained-scraping scene (BASELINE configs[4]) run TIME-CHUNKED (round 5: dense increments + scan +
the CHUNKED FORCED block kernel): where wave 0 of every (team, chunk) workgroup spends its cycles, when the workgroups start and end.
usage: r05_census_tc_dense.py [off] [shape] [chunk_buffers]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PBSO_CENSUS"] = "1"
args = sys.argv[1:]
qn_off = bool(args) and args[0] == "off"
if qn_off:
    args = args[1:]
if len(args) > 0:
    os.environ["PBSO_TC_SHAPE"] = args[0]
if len(args) > 1:
    os.environ["PBSO_TIME_CHUNKS"] = args[1]
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402
n_obj, M, nb = 8, 4096, 86
eng = Engine(qnorm=capi.QNORM_OFF if qn_off else capi.QNORM_ALL, form=capi.FORM_BLOCK)
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
R, cb, teams = info["last_time_chunk_shape"], info["last_time_chunk_buffers"], info["last_time_chunk_teams"]
n_chunks = (nb + cb - 1) // cb if cb else 0
print(f"time-chunked launches {info['total_time_chunk_launches']} (dense increments: {info['total_dense_increment_launches']}), "
      f"shape R={R}, {cb} buffers per chunk, {teams} teams x {n_chunks} chunks, kernel_ms={info['last_step_kernel_ms']:.3f}")
c = eng.census(teams * n_chunks)
t0, t1 = c[:, 0].astype(np.int64), c[:, 1].astype(np.int64)
base = t0.min()
clk = (c[:, 5].astype(np.int64) - c[:, 4].astype(np.int64)) / np.maximum(t1 - t0, 1) * 100.0
dur = (t1 - t0) / 100.0
print(f"WG start us after the first: median {np.median(t0 - base) / 100:.1f} max {(t0 - base).max() / 100:.1f}; duration us min {dur.min():.0f} "
      f"median {np.median(dur):.0f} max {dur.max():.0f}; span {(t1.max() - base) / 100:.1f} us; clock median {np.median(clk):.0f} MHz")
nbw = np.minimum(cb, nb - (np.arange(len(c)) // teams) * cb).astype(np.float64)      # buffers of the row's chunk
names = {6: "head", 7: "matrix (MFMA + operand reads)", 8: "barrier", 9: "combine", 10: "sample 0 + taps", 11: "state stepping"}
tot = c[:, 6:12].astype(np.float64).sum(axis=1)
for k, name in names.items():
    v = c[:, k].astype(np.float64)
    print(f"  {name:32s} median {np.median(v / nbw):8.0f} cycles per buffer ({np.median(v / tot) * 100:5.1f} %)")
print(f"  total {np.median(tot / nbw):.0f} cycles per buffer; WG cycles median {np.median(c[:, 5].astype(np.int64) - c[:, 4].astype(np.int64)):.0f}")
# residency: how many workgroups of the launch shared a CU at once
hw, xcc = c[:, 2], c[:, 3]
key = np.array([((int(a) & 0xF) << 16) | (int(b) & 0xFF00) | ((int(b) >> 13) & 7) for a, b in zip(xcc, hw)])
conc = []
for kk in set(key.tolist()):
    m = key == kk
    ev = sorted([(int(t), 1) for t in t0[m]] + [(int(t), -1) for t in t1[m]])
    cur = best = 0
    for _, d in ev:
        cur += d
        best = max(best, cur)
    conc.append(best)
W = None
print(f"CUs used {len(conc)}; max concurrent workgroups per CU: min {min(conc)} median {int(np.median(conc))} max {max(conc)}")
