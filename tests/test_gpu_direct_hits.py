"""DESC_DIRECT: the hit of a plain PointForce at a vertex of an object with no live force is projected by the oscillator
bank itself (three rows of the per-object (float)(c3 * shape) table, dotted with the hit's normal) -- no force row, no
work for the preparation kernels.  Every bank kernel against the oracle, the rows the planner emits, and the messages
that must NOT take the shortcut (tools/real_time_modal_sound.cpp:236-295 for the projections, modal_solver.h:195-221)."""
import numpy as np
import pytest

from openpbso_amd import capi, synth
from tests.scenarios import ObjSpec, force_ev, rel_errors, run_engine, run_oracle

pytestmark = pytest.mark.gpu
NB = 24


@pytest.fixture(autouse=True, params=["0", "1"], ids=["k1b", "auto"])
def _bank_kernel(request, monkeypatch):
    """these scenes are small: with "1" (the default) the f32 block form may hand a launch to the pipeline kernel K1p
    (kernels_pipe.hip) or cut it into time chunks (K5); "0" pins the one-wave-per-64-modes kernel.  DESC_DIRECT descriptors
    are read by all of them."""
    monkeypatch.setenv("PBSO_SPLIT", request.param)


def _scene(n_obj=6, n_modes=300, seed=5):
    objs, evs = [], []
    rng = np.random.default_rng(seed)
    for i in range(n_obj):
        s = synth.seed_for(7, i)
        objs.append(ObjSpec(synth.eigenvalues(n_modes, s), shapes=synth.mode_shapes(n_modes, s)))
        n_verts = objs[-1].shapes.shape[1] // 3
        vns = synth.unit_normals(NB, s)
        for b in range(NB):
            if rng.random() < 0.45:
                evs.append(force_ev(b, i, vid=int(rng.integers(0, n_verts)), vn=vns[b]))
        evs.append(dict(t=0, obj=i, kind="use_transfer", use=False))
    return objs, evs


@pytest.mark.parametrize("form,mpl", [(capi.FORM_BLOCK_BF16, 0), (capi.FORM_BLOCK_BF16, 4), (capi.FORM_BLOCK, 2), (capi.FORM_BLOCK, 4),
                                      (capi.FORM_VELOCITY, 0), (capi.FORM_VELOCITY, 3), (capi.FORM_DIRECT, 1)])
def test_vertex_hits_through_the_bank_match_the_oracle(form, mpl):
    objs, evs = _scene()
    want = run_oracle(objs, evs, NB)
    got = run_engine(objs, evs, NB, form=form, modes_per_lane=mpl, split=[10, 14])
    assert np.array_equal(got["emitted"], want["emitted"])
    mx, l2 = rel_errors(got["audio"], want["audio"])
    tol = (2e-2, 3e-2) if form == capi.FORM_DIRECT else (5e-4, 1e-3)
    assert (mx <= tol[0]).all() and (l2 <= tol[1]).all(), (mx.max(), l2.max())
    assert got["info"]["last_step_forced_rows"] == 0          # no force row was built: the bank did the projection


def test_switch_off_sends_them_through_the_fp64_kernels(monkeypatch):
    objs, evs = _scene(n_obj=3)
    a = run_engine(objs, evs, NB)
    monkeypatch.setenv("PBSO_DIRECT_HITS", "0")
    b = run_engine(objs, evs, NB)
    assert a["info"]["last_step_forced_rows"] == 0 and b["info"]["last_step_forced_rows"] > 0
    assert np.abs(a["audio"] - b["audio"]).max() <= 1e-5 * np.abs(b["audio"]).max()      # f32 table vs rounded fp64 dot, 24 buffers on


def test_messages_that_keep_the_general_path():
    """a face hit, a hit while a Gaussian force is alive, a hit on a sustained object and a hit in the same buffer as a
    clear: none of them is a DESC_DIRECT descriptor, all must match the oracle"""
    n_modes = 200
    s = synth.seed_for(7, 40)
    shapes = synth.mode_shapes(n_modes, s)
    objs = [ObjSpec(synth.eigenvalues(n_modes, synth.seed_for(7, 40 + i)), shapes=shapes) for i in range(4)]
    vn = synth.unit_normals(8, s)
    rng = np.random.default_rng(3)
    evs = [
        force_ev(1, 0, vids=[3, 9, 20], coords=np.array([0.1, 0.6, 0.3]), vn=vn[0]),                   # face hit
        force_ev(0, 1, data=rng.standard_normal(n_modes) * 1e-3, force_type=1, width=900.0),          # Gaussian spanning buffers
        force_ev(1, 1, vid=5, vn=vn[1]),                                                                # ... and a hit meanwhile
        force_ev(0, 2, data=rng.standard_normal(n_modes) * 1e-3, force_type=2, start=True),           # sustained AR
        force_ev(2, 2, vid=7, vn=vn[2]),                                                                # hit on the sustained object: data swap
        force_ev(4, 2, force_type=2, end=True),
        force_ev(6, 2, vid=8, vn=vn[3]),                                                                # idle again: direct
        force_ev(2, 3, vid=2, vn=vn[4]), force_ev(3, 3, clear=True), force_ev(4, 3, vid=4, vn=vn[5]),
    ] + [dict(t=0, obj=i, kind="use_transfer", use=False) for i in range(4)]
    want = run_oracle(objs, evs, 10)
    got = run_engine(objs, evs, 10)
    assert np.array_equal(got["emitted"], want["emitted"])
    mx, l2 = rel_errors(got["audio"], want["audio"])
    assert (mx <= 5e-4).all() and (l2 <= 1e-3).all(), (mx, l2)


@pytest.mark.parametrize("form,mpl", [(capi.FORM_BLOCK_BF16, 4), (capi.FORM_BLOCK_BF16, 1), (capi.FORM_BLOCK, 2), (capi.FORM_VELOCITY, 2)])
def test_vertex_hits_on_large_teams(form, mpl, monkeypatch):
    """4096-mode objects stepped by teams of 8 waves (16 in the per-sample kernel): the landing areas of the direct-hit
    rows then sit beyond the first 64 KB of the workgroup's LDS, and an object is cut into several teams"""
    monkeypatch.setenv("PBSO_TEAM_WAVES", "16")
    n_modes, nb = 4096, 8
    objs, evs = [], []
    for i in range(2):
        s = synth.seed_for(7, 60 + i)
        objs.append(ObjSpec(synth.eigenvalues(n_modes, s), shapes=synth.mode_shapes(n_modes, s)))
        vns = synth.unit_normals(nb, s)
        for b in (0, 1, 3, 4, 6):
            evs.append(force_ev(b, i, vid=17 * b + 3 * i, vn=vns[b]))
        evs.append(dict(t=0, obj=i, kind="use_transfer", use=False))
    want = run_oracle(objs, evs, nb)
    got = run_engine(objs, evs, nb, form=form, modes_per_lane=mpl)
    mx, l2 = rel_errors(got["audio"], want["audio"])
    assert (mx <= 5e-4).all() and (l2 <= 1e-3).all(), (mx.max(), l2.max())
    assert got["info"]["last_step_forced_rows"] == 0 and got["info"]["waves_per_object"] >= 8


def test_hit_script_overflow_is_rejected_like_try_enqueue_not_fatal():
    """a script that brings more hits than a BUSY object's 1023-slot queue can take: the surplus is rejected the way
    enqueueForceMessage returns false for a full queue (modal_solver.h:329-333; the tool ignores the bool,
    tools/real_time_modal_sound.cpp:610) -- counted in pbso_engine_info::total_dropped_hits, the step succeeds and the engine
    stays usable (round 3: PBSO_ERR_STATE, and the engine refused every later step)"""
    from openpbso_amd import Engine, ForceMessage
    n_modes, nb = 96, 4
    s = synth.seed_for(9, 0)
    lam, shapes = synth.eigenvalues(n_modes, s), synth.mode_shapes(n_modes, s)
    nv = shapes.shape[1] // 3
    n_hits = 1100
    vns = synth.unit_normals(n_hits, s)
    with Engine() as eng:
        for _ in range(2):
            eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes)
        eng.finalize()
        for i in range(2):
            eng.set_use_transfer(i, False)
        # object 1 is busy (a Gaussian force is alive), so its hits take the queue: stamped far ahead, they pile up
        assert eng.enqueue_force(1, ForceMessage(data=np.full(n_modes, 1e-3), forceType=1, gaussianWidth=5000.0), 0)
        o = np.concatenate([np.zeros(2, np.int32), np.ones(n_hits, np.int32)])
        v = (np.arange(2 + n_hits) % nv).astype(np.int32)
        t = np.concatenate([np.array([0, 1]), np.full(n_hits, 1000)]).astype(np.int64)
        assert eng.enqueue_vertex_hits(o, v, np.concatenate([vns[:2], vns]), t) == 2 + n_hits
        eng.step(nb)
        a = eng.audio()
        info = eng.info()
        # (the Gaussian message still sat in object 1's queue when the script arrived: 1022 free slots)
        assert info["total_dropped_hits"] == n_hits - 1022 and np.isfinite(a).all() and np.abs(a[0]).max() > 0
        eng.step(nb)                                         # still usable
        assert eng.enqueue_force(0, ForceMessage(vid=1, vn=vns[0]), 2 * nb)
        eng.step(nb)
        assert np.abs(eng.audio()[0]).max() > 0 and eng.info()["total_dropped_hits"] == n_hits - 1022
