#!/usr/bin/env python3
"""PBSO_CENSUS=1: wave 0's cycles per buffer by phase (head / pipeline / barrier / combine) for the 1024 x 512 walk and its time-chunked
shares, N buffers per launch.  usage: r05_census_share.py <objects> [buffers]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PBSO_CENSUS"] = "1"
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402
n_obj = int(sys.argv[1]); nb = int(sys.argv[2]) if len(sys.argv) > 2 else 860
eng = Engine(qnorm=capi.QNORM_ALL, form=capi.FORM_BLOCK, chunk_buffers=nb)
for i in range(n_obj):
    s = synth.seed_for(4, i)
    eng.add_object(synth.eigenvalues(512, s), synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=synth.mode_shapes(512, s))
eng.finalize()
fo, fv, fn, ft = [], [], [], []
for i in range(n_obj):
    eng.set_use_transfer(i, False)
    s = synth.seed_for(4, i)
    hits, vns = synth.poisson_hits(3 * nb, s), synth.unit_normals(3 * nb, s)
    hb = np.nonzero(hits >= 0)[0]
    fo.append(np.full(hb.size, i, dtype=np.int32)); fv.append(hits[hb].astype(np.int32)); fn.append(vns[hb]); ft.append(hb)
fo, fv, fn, ft = (np.concatenate(x) for x in (fo, fv, fn, ft))
for k in range(3):
    m = (ft // nb) == k
    o = np.lexsort((ft[m], fo[m]))
    eng.enqueue_vertex_hits(fo[m][o], fv[m][o], np.ascontiguousarray(fn[m][o]), ft[m][o].astype(np.int64))
    eng.step(nb)
eng.sync()
info = eng.info()
tc = info["total_time_chunk_launches"] > 0
cb = info["last_time_chunk_buffers"] if tc else nb
teams = info["last_time_chunk_teams"] if tc else info["n_teams"]
n_chunks = (nb + cb - 1) // cb if tc else 1
c = eng.census(teams * n_chunks)
nbw = np.minimum(cb, nb - (np.arange(len(c)) // teams) * cb).astype(np.float64) if tc else np.full(len(c), float(nb))
t0, t1 = c[:, 0].astype(np.int64), c[:, 1].astype(np.int64)
clk = (c[:, 5].astype(np.int64) - c[:, 4].astype(np.int64)) / np.maximum(t1 - t0, 1) * 100.0
tot = c[:, 6:10].astype(np.float64).sum(axis=1)
print(f"{n_obj} x 512 x {nb}: {'time-chunked, ' + str(n_chunks) + ' chunks of ' + str(cb) if tc else 'walk'}; kernel_ms {info['last_step_kernel_ms']:.3f}; "
      f"WG duration us median {np.median((t1 - t0) / 100.0):.0f} max {((t1 - t0) / 100.0).max():.0f}, span {(t1.max() - t0.min()) / 100.0:.0f}; clock median {np.median(clk):.0f} MHz")
for k, name in ((6, "head"), (7, "pipeline"), (8, "barrier"), (9, "combine")):
    v = c[:, k].astype(np.float64)
    print(f"  {name:9s} {np.median(v / nbw):8.0f} cycles per buffer ({np.median(v / tot) * 100:5.1f} %)")
print(f"  total     {np.median(tot / nbw):8.0f} cycles per buffer; outside the loop {np.median((c[:, 5].astype(np.int64) - c[:, 4].astype(np.int64)) - tot):.0f} cycles per workgroup")
