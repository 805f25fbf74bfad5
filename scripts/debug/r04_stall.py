"""per-call host times of feed + step for a small scene: where do multi-millisecond stalls come from?"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from openpbso_amd import Engine, capi, synth
n_obj, n_modes, nb, steps = int(sys.argv[1]), 512, 86, int(sys.argv[2])
eng = Engine(qnorm=capi.QNORM_ALL, timing_every=4)
rng = np.random.default_rng(1)
for i in range(n_obj):
    s = synth.seed_for(3, i)
    eng.add_object(synth.eigenvalues(n_modes, s), synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=synth.mode_shapes(n_modes, s))
eng.finalize()
for i in range(n_obj):
    eng.set_use_transfer(i, False)
nv = synth.N_VERTS
feeds = []
for k in range(steps + 1):
    o, v, t = [], [], []
    for i in range(n_obj):
        hb = np.nonzero(rng.random(nb) < 0.233)[0]
        o.append(np.full(hb.size, i, np.int32)); v.append(rng.integers(0, nv, hb.size).astype(np.int32)); t.append((k * nb + hb).astype(np.int64))
    o, v, t = np.concatenate(o), np.concatenate(v), np.concatenate(t)
    feeds.append((o, v, synth.unit_normals(o.size, k), t))
eng.enqueue_vertex_hits(*feeds[0])
te, ts = [], []
t00 = time.perf_counter()
for k in range(steps):
    t0 = time.perf_counter(); eng.step(nb); t1 = time.perf_counter(); eng.enqueue_vertex_hits(*feeds[k + 1]); t2 = time.perf_counter()
    ts.append((t1 - t0) * 1e3); te.append((t2 - t1) * 1e3)
eng.sync()
tot = (time.perf_counter() - t00) * 1e3
te, ts = np.array(te), np.array(ts)
print(f"{n_obj}x512, {steps} steps: total {tot:.1f} ms = {tot/steps:.3f} per step; step() median {np.median(ts):.3f} max {ts.max():.2f} at {ts.argmax()}; enqueue median {np.median(te):.3f} max {te.max():.2f} at {te.argmax()}")
print("  calls over 1 ms: step", [(int(i), round(float(x), 1)) for i, x in enumerate(ts) if x > 1.0], "enqueue", [(int(i), round(float(x), 1)) for i, x in enumerate(te) if x > 1.0])
