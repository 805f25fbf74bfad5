#!/usr/bin/env python3
"""Does a time-chunked launch of several ROUNDS of workgroups run its buffers slower, and what does it depend on?
   python scripts/debug/r04_tcrounds.py <objects> <time_chunks> <p_hit> [direct_hits]  -> kernel ms per launch (HIP events)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpbso_amd import Engine, synth   # noqa: E402

n_obj, tc, p_hit = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
direct = int(sys.argv[4]) if len(sys.argv) > 4 else 0
M, nb, steps = 512, int(os.environ.get("NB", "86")), 14
eng = Engine(time_chunks=tc, direct_hits=direct, timing_every=1, chunk_buffers=max(128, nb))
lam = synth.eigenvalues(M, 7)
shp = synth.mode_shapes(M, 7)
nv = shp.shape[1] // 3
for i in range(n_obj):
    eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shp)
eng.finalize()
for i in range(n_obj):
    eng.set_use_transfer(i, False)
rng = np.random.default_rng(3)
ks = []
for k in range(steps):
    hit = rng.random((n_obj, nb)) < p_hit
    o, b = np.nonzero(hit)
    if o.size:
        v = rng.integers(0, nv, o.size).astype(np.int32)
        vn = synth.unit_normals(o.size, 11 + k)
        assert eng.enqueue_vertex_hits(o.astype(np.int32), v, vn, (k * nb + b).astype(np.int64)) == o.size
    eng.step(nb)
    if os.environ.get("SYNC", "1") == "1":
        eng.sync()
    ks.append(eng.info()["last_step_kernel_ms"])
info = eng.info()
eng.sync()
if os.environ.get("ALL"):
    print("kernel ms per step:", [round(x, 3) for x in ks])
print(f"objects={n_obj} nb={nb} time_chunks={tc} p_hit={p_hit} direct_hits={direct}: kernel ms per launch median {np.median(ks[4:]):.4f} (min {min(ks[4:]):.4f}); "
      f"time-chunked launches {info['total_time_chunk_launches']} of {info['total_block_launches']}")
eng.close()
