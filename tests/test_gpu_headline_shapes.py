"""Oracle checks at the launch shapes the headline numbers are quoted on (NB = 860, ten seconds of audio per
launch): 1024 x 512 x 860 in ONE launch of the f32 block form, its 512 / 256 / 128-object shares (time-chunked, gated by
policy) and 8 x 4096 x 860 of sustained scraping.  A sample of objects spread over the id range goes through the fp64 oracle
(modal_solver.h:181-276 restated); `emitted` is compared for all.  bench.py checks eight rows of its own first timed step; this is
the same shape inside the -m gpu suite."""
import os

import numpy as np
import pytest

from openpbso_amd import Engine, capi, synth
from tests.scenarios import ObjSpec, force_ev, rel_errors, run_oracle

pytestmark = pytest.mark.gpu
NB = 860
THREADS = max(1, min(8, len(os.sched_getaffinity(0))))
_cache = {}


def _object(i, M):
    key = (i, M)
    if key not in _cache:
        seed = synth.seed_for(4, i)
        _cache[key] = (synth.eigenvalues(M, seed), synth.mode_shapes(M, seed), synth.poisson_hits(NB, seed), synth.unit_normals(NB, seed))
    return _cache[key]


@pytest.mark.parametrize("n_obj", [1024, 512, 256, 128])
def test_1024x512_and_its_shares_at_860_buffers_per_launch(n_obj):
    M = 512
    with Engine(qnorm=capi.QNORM_ALL, form=capi.FORM_BLOCK, chunk_buffers=NB, plan_threads=4) as eng:
        for i in range(n_obj):
            lam, shapes, _, _ = _object(i, M)
            eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes)
        eng.finalize()
        fo, fv, fn, ft = [], [], [], []
        for i in range(n_obj):
            eng.set_use_transfer(i, False)
            _, _, hits, vns = _object(i, M)
            hb = np.nonzero(hits >= 0)[0]
            fo.append(np.full(hb.size, i, dtype=np.int32)); fv.append(hits[hb].astype(np.int32)); fn.append(vns[hb]); ft.append(hb)
        fo, fv, fn, ft = (np.concatenate(x) for x in (fo, fv, fn, ft))
        order = np.lexsort((ft, fo))                     # object-major, each object's hits in time order
        eng.enqueue_vertex_hits(fo[order], fv[order], np.ascontiguousarray(fn[order]), ft[order].astype(np.int64))
        eng.step(NB)
        rng = np.random.default_rng(n_obj)
        sample = sorted(set([0, 1, n_obj // 2 - 1, n_obj // 2, n_obj - 2, n_obj - 1] + rng.integers(0, n_obj, 5).tolist()))
        audio = eng.audio_rows(sample)
        emitted = eng.emitted().copy()
        qn_got = {(k, b): eng.qnorm(i, b).copy() for k, i in enumerate(sample[:4]) for b in (0, 430, NB - 1)}
        info = eng.info()
    # ONE launch of 860 buffers: the full scene walks them in buffer order (one round of workgroups fills the chip), its shares
    # are cut in time behind the scan (K5)
    assert info["total_block_launches"] == 1 and info["total_sample_launches"] == 0 and info["total_dropped_hits"] == 0
    assert info["total_time_chunk_launches"] == (0 if n_obj == 1024 else 1), info["total_time_chunk_launches"]
    assert emitted.shape == (n_obj, NB) and emitted.all()
    objs, evs = [], []
    for k, i in enumerate(sample):
        lam, shapes, hits, vns = _object(i, M)
        objs.append(ObjSpec(lam, shapes=shapes))
        evs += [force_ev(int(b), k, vid=int(v), vn=vns[b]) for b, v in enumerate(hits) if v >= 0]
        evs.append(dict(t=0, obj=k, kind="use_transfer", use=False))
    want = run_oracle(objs, evs, NB, threads=THREADS)
    mx, l2 = rel_errors(audio, want["audio"])
    assert np.isfinite(audio).all() and (mx <= 5e-4).all() and (l2 <= 1e-3).all(), (mx.max(), l2.max())
    assert mx.max() <= 5e-5, mx.max()                       # measured 1e-5: ten seconds of audio from rest
    for key, g in qn_got.items():
        w = want["qnorm"][key]
        assert np.abs(g[:M] - w).max() <= 5e-4 * np.abs(w).max(), key
    print(f"{n_obj} x 512 x 860 in one launch: {len(sample)} objects, max/peak {mx.max():.2e} relL2 {l2.max():.2e}, "
          f"time-chunked launches {info['total_time_chunk_launches']}")


@pytest.mark.parametrize("qnorm", [capi.QNORM_ALL, capi.QNORM_OFF])
def test_8x4096_sustained_scraping_at_860_buffers_per_launch(qnorm):
    """BASELINE configs[4] at the throughput duration: ten seconds of sustained AutoregressiveForce contact (forces.h:107-128,
    modal_solver.h:222-240), a face hit per buffer, an AR parameter update on the way, in ONE launch of 860 dense-profile buffers --
    with qnorm rows cut in time (dense increments + scan + the forced block path), without them on the five-role pipeline teams"""
    n_obj, M = 8, 4096
    objs, evs = [], []
    for i in range(n_obj):
        seed = synth.seed_for(5, i)
        lam = synth.eigenvalues(M, seed)
        objs.append(ObjSpec(lam, shapes=synth.mode_shapes(M, seed)))
        rng = np.random.default_rng(seed)
        vns = synth.unit_normals(NB, seed)
        evs.append(dict(t=0, obj=i, kind="use_transfer", use=False))
        evs.append(force_ev(0, i, force_type=2, start=True))           # dummy start message: data = 0
        for b in range(1, NB - 40):
            bary = rng.random(3)
            evs.append(force_ev(b, i, vids=rng.integers(0, synth.N_VERTS, 3), coords=bary / bary.sum(), vn=vns[b], force_type=2))
        evs.append(dict(t=300, obj=i, kind="arprm", a=[0.6, 0.2], sigma=0.002, mu=0.1))
        evs.append(force_ev(NB - 40, i, force_type=2, end=True))
    from tests.scenarios import run_engine
    got = run_engine(objs, evs, NB, qnorm=qnorm, form=capi.FORM_BLOCK, chunk_buffers=NB)
    info = got["info"]
    assert info["total_block_launches"] == 1 and info["total_sample_launches"] == 0
    if qnorm == capi.QNORM_ALL:
        assert info["total_time_chunk_launches"] == 1 and info["total_dense_increment_launches"] == 1
    else:
        assert info["total_split_launches"] == 1
    want = run_oracle(objs, evs, NB, threads=THREADS)
    assert np.array_equal(got["emitted"], want["emitted"])
    mx, l2 = rel_errors(got["audio"], want["audio"])
    assert np.isfinite(got["audio"]).all() and (mx <= 5e-4).all() and (l2 <= 1e-3).all(), (mx.max(), l2.max())
    if qnorm == capi.QNORM_ALL:
        for (i, b), w in want["qnorm"].items():
            if b in (0, 1, 299, 300, 301, NB - 41, NB - 40, NB - 1):
                assert np.abs(got["qnorm"][(i, b)] - w).max() <= 5e-4 * np.abs(w).max(), (i, b)
    print(f"8 x 4096 x 860 scraping, qnorm {'on' if qnorm else 'off'}: max/peak {mx.max():.2e} relL2 {l2.max():.2e}")
