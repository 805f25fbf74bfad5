"""Launches with more ROWS than a grid's y dimension may count (65535): the row-parallel preparation kernels -- projection,
combine, scatter, transfer-row copies, FFAT lookups, dense increments -- put their rows on grid.x (round 6; ADVICE r05: a
failed launch marks the engine failed for good).  Two scenes whose single launch has more rows than that, against the fp64
oracle (modal_solver.h:181-276, forces.h:81-137 restated) on a sample of objects, `emitted` for all."""
import os

import numpy as np
import pytest

from openpbso_amd import Engine, ForceMessage, capi, synth
from tests.scenarios import ObjSpec, force_ev, rel_errors, run_oracle

pytestmark = pytest.mark.gpu
THREADS = max(1, min(8, len(os.sched_getaffinity(0))))
M = 64


def _lam(i):
    return synth.eigenvalues(M, synth.seed_for(7, i))


def test_more_face_hits_in_one_launch_than_grid_y_counts():
    """70 400 face hits (GetModalForceFace, tools/real_time_modal_sound.cpp:281-295) in one launch: every one is a projection
    event and a combine row of its own"""
    n_obj, nb = 1100, 64
    shapes = {i: synth.mode_shapes(M, synth.seed_for(7, i)) for i in range(n_obj)}
    rng = np.random.default_rng(11)
    objs = np.repeat(np.arange(n_obj, dtype=np.int32), nb)
    stamps = np.tile(np.arange(nb, dtype=np.int64), n_obj)
    vids = rng.integers(0, synth.N_VERTS, (objs.size, 3))
    bary = rng.random((objs.size, 3))
    bary /= bary.sum(axis=1, keepdims=True)
    vns = rng.standard_normal((objs.size, 3))
    vns /= np.linalg.norm(vns, axis=1, keepdims=True)
    with Engine(qnorm=capi.QNORM_OFF, form=capi.FORM_BLOCK, chunk_buffers=nb) as eng:
        for i in range(n_obj):
            eng.add_object(_lam(i), synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes[i])
        eng.finalize()
        for i in range(n_obj):
            eng.set_use_transfer(i, False)
        o, msgs, t = Engine.hit_messages(objs, vids, vns, stamps, coords=bary)
        assert eng.enqueue_force_batch(o, msgs, t) == objs.size
        eng.step(nb)
        sample = [0, 1, 547, 1023, 1024, n_obj - 1]
        audio = eng.audio_rows(sample)
        emitted = eng.emitted().copy()
        info = eng.info()
    assert info["total_block_launches"] + info["total_sample_launches"] == 1 and emitted.all()
    specs, evs = [], []
    for k, i in enumerate(sample):
        specs.append(ObjSpec(_lam(i), shapes=shapes[i]))
        evs.append(dict(t=0, obj=k, kind="use_transfer", use=False))
        for b in range(nb):
            j = i * nb + b
            evs.append(force_ev(b, k, vids=vids[j], coords=bary[j], vn=vns[j]))
    want = run_oracle(specs, evs, nb, threads=THREADS)
    mx, l2 = rel_errors(audio, want["audio"])
    assert np.isfinite(audio).all() and (mx <= 5e-4).all() and (l2 <= 1e-3).all(), (mx.max(), l2.max())
    print(f"{objs.size} face hits in one launch: max/peak {mx.max():.2e}")


@pytest.mark.parametrize("time_chunks", [0, 8])
def test_all_dense_scene_with_more_row_groups_than_grid_y_counts(time_chunks):
    """4200 objects in sustained AutoregressiveForce contact (forces.h:107-128, modal_solver.h:222-240) for 64 buffers: 268 800
    dense-profile rows in one launch -- 67 200 groups of four for dense_increment_kernel when the launch is cut in time (forced
    here, and whatever the policy picks)"""
    n_obj, nb = 4200, 64
    rng = np.random.default_rng(5)
    data = {i: rng.standard_normal(M) * 1e-3 for i in range(n_obj)}
    with Engine(qnorm=capi.QNORM_OFF, form=capi.FORM_BLOCK, chunk_buffers=nb, time_chunks=time_chunks) as eng:
        for i in range(n_obj):
            eng.add_object(_lam(i), synth.RHO, synth.ALPHA, synth.BETA)
        eng.finalize()
        for i in range(n_obj):
            eng.set_use_transfer(i, False)
            assert eng.enqueue_force(i, ForceMessage(data=data[i], forceType=2, sustainedForceStart=True), 0)
        eng.step(nb)
        sample = [0, 1, 2099, 4095, 4096, n_obj - 1]
        audio = eng.audio_rows(sample)
        emitted = eng.emitted().copy()
        info = eng.info()
    assert emitted.all()
    if time_chunks:
        assert info["total_time_chunk_launches"] == 1 and info["total_dense_increment_launches"] == 1, info
    specs, evs = [], []
    for k, i in enumerate(sample):
        specs.append(ObjSpec(_lam(i)))
        evs.append(dict(t=0, obj=k, kind="use_transfer", use=False))
        evs.append(force_ev(0, k, data=data[i], force_type=2, start=True))
    want = run_oracle(specs, evs, nb, threads=THREADS)
    mx, l2 = rel_errors(audio, want["audio"])
    assert np.isfinite(audio).all() and (mx <= 5e-4).all() and (l2 <= 1e-3).all(), (mx.max(), l2.max())
    print(f"{n_obj * nb} dense rows in one launch (time_chunks={time_chunks}): max/peak {mx.max():.2e}, "
          f"dense-increment launches {info['total_dense_increment_launches']}")
