#!/usr/bin/env python3
"""Where and when did the oscillator-bank workgroups run?  (PBSO_CENSUS=1)"""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PBSO_CENSUS"] = "1"
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402

n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nb = 86
rng = np.random.default_rng(0)
R = int(sys.argv[3]) if len(sys.argv) > 3 else 0
eng = Engine(qnorm=capi.QNORM_ALL, modes_per_lane=R)
for i in range(n_obj):
    eng.add_object(synth.eigenvalues(M, 100 + i), synth.RHO, synth.ALPHA, synth.BETA)
eng.finalize()
for i in range(n_obj):
    eng.set_use_transfer(i, False)
    eng.enqueue_force(i, ForceMessage(data=rng.standard_normal(M) * 1e-3))
for _ in range(3):
    eng.step(nb)
eng.sync()
info = eng.info()
c = eng.census()
t0, t1, hw, xcc = c[:, 0].astype(np.int64), c[:, 1].astype(np.int64), c[:, 2], c[:, 3]
base = t0.min()
dur = (t1 - t0) / 100.0   # us
print(f"objects={n_obj} R={info['modes_per_lane']} W={info['waves_per_object']} lds={info['lds_bytes_per_workgroup']} kernel_ms={info['last_step_kernel_ms']:.3f}")
print(f"WG start (us after first): min {(t0 - base).min() / 100:.1f} median {np.median(t0 - base) / 100:.1f} max {(t0 - base).max() / 100:.1f}")
print(f"WG duration (us): min {dur.min():.1f} median {np.median(dur):.1f} max {dur.max():.1f}; span {(t1.max() - base) / 100:.1f} us")
late = (t0 - base) > 0.1 * (t1.max() - base)
print(f"WGs starting later than 10% of the span: {late.sum()}")
cu = collections.Counter(zip(xcc.tolist(), ((hw >> 8) & 0xFF).tolist(), ((hw >> 13) & 0x7).tolist()))
hist = collections.Counter(cu.values())
print(f"distinct (xcc, cu/sh, se) = {len(cu)}; WGs per CU histogram: {sorted(hist.items())}")
print("hw_id samples:", [hex(int(x)) for x in hw[:6]], "xcc:", sorted(set(int(x) & 0xF for x in xcc)))
simd = collections.Counter(((hw >> 4) & 0x3).tolist())
print("wave0 simd histogram:", sorted(simd.items()))
clk = (c[:, 5].astype(np.int64) - c[:, 4].astype(np.int64)) / np.maximum(t1 - t0, 1) * 100.0   # MHz
print(f"per-WG effective shader clock (MHz): min {clk.min():.0f} median {np.median(clk):.0f} max {clk.max():.0f}")
x = (xcc & 0xF).astype(int)
for xi in sorted(set(x.tolist())):
    m = x == xi
    print(f"  xcc {xi}: n={m.sum()} dur median {np.median(dur[m]):.0f} max {dur[m].max():.0f} us; clock median {np.median(clk[m]):.0f} MHz")
# spread inside one CU
key = np.array([(int(a) << 16) | (int(b) & 0xFF00) | ((int(b) >> 13) & 7) for a, b in zip(x, hw)])
spreads = []
for kk in set(key.tolist()):
    dd = dur[key == kk]
    spreads.append((dd.max() - dd.min(), dd.min(), dd.max()))
spreads.sort()
print("within-CU (max-min) duration spread us: median %.0f, max %.0f; slowest CU durations: %s" % (
    np.median([s_[0] for s_ in spreads]), spreads[-1][0], sorted(dur[key == max(set(key.tolist()), key=lambda kk: dur[key == kk].max())].round().tolist())))
cyc = (c[:, 5].astype(np.int64) - c[:, 4].astype(np.int64))
print(f"per-WG shader cycles: min {cyc.min():.3e} median {np.median(cyc):.3e} max {cyc.max():.3e}")
# does the finishing order inside a CU follow the wave slot of the workgroup?
wid = (hw & 0xF).astype(int)
for w in sorted(set(wid.tolist())):
    m = wid == w
    print(f"  wave slot {w}: n={m.sum()} duration median {np.median(dur[m]):.0f} us (min {dur[m].min():.0f}, max {dur[m].max():.0f})")
order = np.argsort(t0, kind="stable")
rank_in_cu = np.zeros(len(t0), dtype=int)
for kk in set(key.tolist()):
    idx = np.nonzero(key == kk)[0]
    idx = idx[np.argsort(c[idx, 0].astype(np.int64) * 0 + idx)]          # dispatch order = team index order
    rank_in_cu[idx] = np.arange(len(idx))
for r in sorted(set(rank_in_cu.tolist())):
    m = rank_in_cu == r
    print(f"  {r}-th team dispatched to its CU: duration median {np.median(dur[m]):.0f} us")

if c.shape[1] >= 10 and c[:, 6:10].any():
    tot = c[:, 6:10].astype(np.float64).sum(axis=1)
    print(f"outside the buffer loop (launch preamble: operand table, coefficients, state; write-back): median {np.median(cyc - tot):.3e} cycles of {np.median(cyc):.3e} per workgroup; time-chunked launches {info['total_time_chunk_launches']}")
    for name, k in (("head", 6), ("pipeline", 7), ("barrier", 8), ("combine", 9)):
        v = c[:, k].astype(np.float64)
        print(f"block form, wave 0 of each team: {name:9s} median {np.median(v):.3e} cycles ({np.median(v / tot) * 100:.1f} % of the loop), per buffer {np.median(v) / nb:.0f}")
