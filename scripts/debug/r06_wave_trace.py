#!/usr/bin/env python3
"""K1b at the headline shape: do the two waves of a SIMD (they belong to different teams) leave the matrix pipeline TOGETHER
for their heads / barriers / combines, or one under the other's pipeline?   (diagnostics build -DPBSO_WAVE_TRACE:
scripts/debug/r06_variant.sh trace "-DPBSO_WAVE_TRACE" all;   PBSO_LIB=openpbso_amd/variants/lib_trace.so python scripts/debug/r06_wave_trace.py)"""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PBSO_CENSUS"] = "1"
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402
import ctypes as C                                            # noqa: E402

TRACE_NB = 12
TRACE_K = 8
WORDS = 12 + 2 * (2 + TRACE_NB * TRACE_K)
n_obj, M, nb = 1024, 512, 86
rng = np.random.default_rng(0)
eng = Engine(qnorm=capi.QNORM_ALL)
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
n = info["n_teams"] * WORDS
raw = np.empty(n, dtype=np.uint64)
eng._chk(eng._l.pbso_read_census(eng._h, raw.ctypes.data_as(C.POINTER(C.c_uint64)), n))
c = raw.reshape(-1, WORDS).astype(np.int64)
print(f"objects={n_obj} R={info['modes_per_lane']} W={info['waves_per_object']} kernel_ms={info['last_step_kernel_ms']:.3f} (census stamps on)")
# per wave: hw_id, xcc, stamps [TRACE_NB][4] = head end, pipeline end, barrier end, combine end
waves = []
for t in range(c.shape[0]):
    for w in range(2):
        base = 12 + w * (2 + TRACE_NB * TRACE_K)
        hw, xcc = int(c[t, base]), int(c[t, base + 1]) & 7
        st = c[t, base + 2: base + 2 + TRACE_NB * TRACE_K].reshape(TRACE_NB, TRACE_K)
        waves.append(dict(team=t, wave=w, simd=(hw >> 4) & 3, cu=(xcc, (hw >> 8) & 0xFF, (hw >> 13) & 7), st=st))
by_simd = collections.defaultdict(list)
for w in waves:
    by_simd[(w["cu"], w["simd"])].append(w)
print("waves per SIMD histogram:", sorted(collections.Counter(len(v) for v in by_simd.values()).items()))
same_team = sum(1 for v in by_simd.values() if len(v) == 2 and v[0]["team"] == v[1]["team"])
print("SIMDs whose two waves belong to ONE team:", same_team, "of", len(by_simd))


def out_intervals(st):
    """intervals in which the wave is OUTSIDE its pipeline: [pipeline end of buffer i, head end of buffer i + 1]"""
    return [(st[i, 1], st[i + 1, 0]) for i in range(TRACE_NB - 1)]


def overlap(a, b):
    tot = 0
    for x0, x1 in a:
        for y0, y1 in b:
            tot += max(0, min(x1, y1) - max(x0, y0))
    return tot


fr, per_buf, out_len, pipe_len = [], [], [], []
for key, v in by_simd.items():
    if len(v) != 2:
        continue
    a, b = out_intervals(v[0]["st"]), out_intervals(v[1]["st"])
    la = sum(x1 - x0 for x0, x1 in a)
    fr.append(overlap(a, b) / max(la, 1))
    for w in v:
        st = w["st"]
        per_buf.append((st[-1, 3] - st[0, 3]) / (TRACE_NB - 1))
        out_len.append(np.mean([x1 - x0 for x0, x1 in out_intervals(st)]))
        pipe_len.append(np.mean(st[1:, 1] - st[1:, 0]))
fr = np.array(fr)
print(f"cycles per buffer (a wave): median {np.median(per_buf):.0f};  outside the pipeline (pipeline end -> next head end): median {np.median(out_len):.0f};  inside: median {np.median(pipe_len):.0f}")
print(f"share of a wave's time outside the pipeline: {np.median(out_len) / np.median(per_buf):.3f}")
print("fraction of a wave's time outside the pipeline that its SIMD partner ALSO spends outside (1 = they leave together, "
      f"share above = independent): median {np.median(fr):.3f}, quartiles {np.percentile(fr, 25):.3f} / {np.percentile(fr, 75):.3f}")
# the pipeline's pace alone and beside the partner: regress a wave's pipeline length on the part of it that the partner spent outside
xs, ys = [], []
for key, v in by_simd.items():
    if len(v) != 2:
        continue
    for me, other in ((v[0], v[1]), (v[1], v[0])):
        o = out_intervals(other["st"])
        for i in range(1, TRACE_NB):
            p0, p1 = me["st"][i, 0], me["st"][i, 1]
            xs.append(overlap([(p0, p1)], o))
            ys.append(p1 - p0)
xs, ys = np.array(xs, float), np.array(ys, float)
A = np.stack([np.ones_like(xs), xs], 1)
coef, *_ = np.linalg.lstsq(A, ys, rcond=None)
print(f"pipeline length of a buffer = {coef[0]:.0f} {coef[1]:+.3f} x (cycles of it the SIMD partner spent outside its own pipeline)   [n = {len(xs)}, partner-outside cycles: median {np.median(xs):.0f}]")
# one SIMD's timeline
key = sorted(k for k, v in by_simd.items() if len(v) == 2)[len(by_simd) // 3]
v = by_simd[key]
t0 = min(w["st"][0, 0] for w in v)
# where the time outside the pipeline goes (stamps: 0 head end, 1 pipeline end, 2 barrier end, 3 combine end, 4 gains + direct-hit prefetch done,
# 5 sample 0 done, 6 qnorm rows done, 7 first slice parked and its 32 operands requested)
seg = collections.defaultdict(list)
for w in waves:
    st = w["st"]
    for i in range(1, TRACE_NB):
        seg["pipeline end -> barrier end"].append(st[i - 1, 2] - st[i - 1, 1])
        seg["barrier end -> combine end"].append(st[i - 1, 3] - st[i - 1, 2])
        seg["combine end -> gains / descriptor / direct prefetch"].append(st[i, 4] - st[i - 1, 3])
        seg["-> sample 0"].append(st[i, 5] - st[i, 4])
        seg["-> qnorm closed form + store"].append(st[i, 6] - st[i, 5])
        seg["-> wave sum of sample 0 (head end)"].append(st[i, 0] - st[i, 6])
        seg["head end -> first slice parked, operands requested"].append(st[i, 7] - st[i, 0])
        seg["-> pipeline end (8 matrix bursts, 7 vector bursts)"].append(st[i, 1] - st[i, 7])
for k_, v_ in seg.items():
    print(f"  {k_:58s} median {np.median(v_):7.0f}   quartiles {np.percentile(v_, 25):7.0f} / {np.percentile(v_, 75):7.0f}")
print(f"timeline of SIMD {key[1]} of CU {key[0]} (cycles after the first stamp): per buffer  head-end  pipe-end  barrier-end  combine-end  gains  sample0  qnorm  first-burst")
for w in v:
    print(f"  team {w['team']} wave {w['wave']}:")
    for i in range(TRACE_NB):
        print("     " + "  ".join(f"{int(x - t0):8d}" for x in w["st"][i]))
