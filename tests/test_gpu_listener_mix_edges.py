"""Edges of the multi-listener mix (pbso_listeners_enable / pbso_mix_listeners) that round 2 left uncovered:
an engine created with PBSO_QNORM_OFF, a step with a clearAllForces buffer (modal_solver.h:186-189: no samples), and a step
in which a wave left the scaled-state representation (no block states: the call must refuse, never return NaN)."""
import numpy as np
import pytest

from openpbso_amd import ForceMessage, capi, synth

pytestmark = pytest.mark.gpu


def _oracle_per_listener(lam, omaps, pos, nb, hits, clear_at=()):
    from oracle import oracle_py as orc
    out = []
    for p in pos:
        s = orc.Solver(lam, synth.RHO, synth.ALPHA, synth.BETA)
        if omaps is not None:
            s.read_ffat_maps(omaps)
            s.compute_transfer(p)
        else:
            s.set_use_transfer(False)
        bufs = []
        for b in range(nb):
            if b in clear_at:
                s.enqueue_force(None, clear_all=True)
            elif b in hits:
                s.enqueue_force(hits[b])
            snd = s.step()
            bufs.append(np.zeros(513) if snd is None or snd[0] is None else snd[0])
        out.append(np.concatenate(bufs))
    return np.array(out)


@pytest.mark.parametrize("form", [capi.FORM_BLOCK, capi.FORM_BLOCK_BF16])
def test_listeners_on_an_engine_without_qnorm_rows(form):
    """PBSO_QNORM_OFF + pbso_listeners_enable: the build of the bank that keeps block-start states evaluates the closed-form
    qnorm rows whatever the mode says, so the engine has to own the G planes by then (round 2: a null pointer, a GPU fault)."""
    from openpbso_amd import Engine
    from oracle import oracle_py as orc
    n_modes, nb, L = 130, 4, 3
    lam = synth.eigenvalues(n_modes, 4242)
    maps = synth.ffat_maps(lam, 4243, dim=4, cell_size=0.01)
    rng = np.random.default_rng(5)
    pos = rng.standard_normal((L, 3))
    pos = 0.4 * pos / np.linalg.norm(pos, axis=1, keepdims=True)
    hits = {0: rng.standard_normal(n_modes) * 1e-3, 2: rng.standard_normal(n_modes) * 1e-3}
    with Engine(form=form, qnorm=capi.QNORM_OFF) as eng:
        oid = eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA)
        eng.set_ffat_maps(oid, maps)
        eng.finalize()
        eng.listeners_enable(oid)
        eng.compute_transfer(oid, pos[0], 0)
        for b, d in hits.items():
            assert eng.enqueue_force(oid, ForceMessage(data=d), b)
        eng.step(nb)
        single = eng.audio()[oid].astype(np.float64)
        mix = eng.mix_listeners(oid, pos).astype(np.float64)
    omaps = [orc.uniform_cube(m["mode_id"], m["k"], m["center"], m["cell_size"], 4, m["psi"]) for m in maps]
    want = _oracle_per_listener(lam, omaps, pos, nb, hits)
    assert np.isfinite(mix).all()
    for l in range(L):
        assert np.abs(mix[l] - want[l]).max() <= 5e-4 * np.abs(want[l]).max(), l
    assert np.abs(single - want[0]).max() <= 5e-4 * np.abs(want[0]).max()


@pytest.mark.parametrize("form", [capi.FORM_BLOCK, capi.FORM_BLOCK_BF16])
def test_mix_is_silent_in_a_clear_all_forces_buffer(form):
    """a clearAllForces message makes step() return before it produces a buffer (modal_solver.h:186-189; the engine writes
    zeros and reports emitted = 0): the mix must do the same for every listener -- round 2 divided by a zero scale there"""
    from openpbso_amd import Engine
    n_modes, nb, L = 96, 5, 9
    lam = synth.eigenvalues(n_modes, 909)
    rng = np.random.default_rng(9)
    pos = rng.standard_normal((L, 3))
    hits = {0: rng.standard_normal(n_modes) * 1e-3, 3: rng.standard_normal(n_modes) * 1e-3}
    with Engine(form=form) as eng:
        oid = eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA)
        eng.finalize()
        eng.listeners_enable(oid)
        eng.set_use_transfer(oid, False)
        # (the force queue is a FIFO: messages are enqueued in the order of their stamps)
        assert eng.enqueue_force(oid, ForceMessage(data=hits[0]), 0)
        assert eng.enqueue_force(oid, ForceMessage(clearAllForces=True), 2)
        assert eng.enqueue_force(oid, ForceMessage(data=hits[3]), 3)
        eng.step(nb)
        single = eng.audio()[oid].astype(np.float64)
        emitted = eng.emitted()[oid]
        mix = eng.mix_listeners(oid, pos).astype(np.float64)
    assert list(emitted) == [True, True, False, True, True]
    assert np.isfinite(mix).all()
    assert (mix[:, 2 * 513:3 * 513] == 0).all() and (single[2 * 513:3 * 513] == 0).all()
    # no FFAT maps: every listener hears the unit transfer (TransMessage::setToUnit), i.e. the engine's own output
    for l in range(L):
        assert np.abs(mix[l] - single).max() <= 1e-4 * np.abs(single).max(), l
    assert np.abs(single).max() > 0


def test_mix_refuses_a_step_without_block_states_for_some_wave():
    """a transfer weight outside [2^-20, 2^40] takes its wave to the per-sample path for that buffer (no block states):
    the planner cannot know, the kernel marks the buffer and pbso_mix_listeners returns PBSO_ERR_STATE instead of NaN"""
    from openpbso_amd import Engine
    from openpbso_amd.solver import PbsoError
    n_modes, nb = 128, 3
    lam = synth.eigenvalues(n_modes, 77)
    maps = synth.ffat_maps(lam, 77, dim=4, cell_size=0.01)
    for m in maps:
        m["psi"] = np.array(m["psi"], dtype=np.float64)
    maps[3]["psi"][:] = 0.0                                     # weight 0: wave 0 cannot scale its state
    rng = np.random.default_rng(1)
    pos = np.array([[0.5, 0.1, 0.2], [0.1, 0.5, 0.2]])
    with Engine(form=capi.FORM_BLOCK) as eng:
        oid = eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA)
        eng.set_ffat_maps(oid, maps)
        eng.finalize()
        eng.listeners_enable(oid)
        assert eng.enqueue_force(oid, ForceMessage(data=rng.standard_normal(n_modes) * 1e-3), 0)
        eng.step(nb)                                            # unit transfer: every wave scaled, the mix works
        assert np.isfinite(eng.mix_listeners(oid, pos)).all()
        eng.compute_transfer(oid, pos[0], nb)                   # now the listener's own transfer has a zero weight
        eng.step(nb)
        with pytest.raises(PbsoError) as ei:
            eng.mix_listeners(oid, pos)
        assert ei.value.status == capi.ERR_STATE
        assert np.isfinite(eng.audio()).all()                   # the engine's own output is unaffected
