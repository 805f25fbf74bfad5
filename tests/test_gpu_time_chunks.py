"""K5 (kernels_scan.hip + the chunked launches of kernels_block.hip): a launch cut along the TIME axis.

The reference admits forces only at the first sample of a buffer (modal_solver.h:184-205) and a mode's recurrence is
linear (modal_integrator.h:103-113): a per-mode scan x_{b+1} = A^513 x_b + impulse_b gives the state at the start of
every buffer, after which the block kernel runs (team, chunk of buffers) workgroups side by side.  Checked here against
the fp64 oracle for every buffer kind, every team shape the engine picks from, both projections, any chunk length and
any cut of the step into launches; plus the policy (when the engine chooses it by itself) and the two claims the design
makes about it: one buffer per chunk makes the audio independent of how a step is cut, and two engines of one process
may differ in their settings (pbso_engine_desc, ABI 4).
"""
import numpy as np
import pytest

from openpbso_amd import capi, synth
from tests.scenarios import B, ObjSpec, force_ev, rel_errors, run_engine, run_oracle

pytestmark = pytest.mark.gpu


def _check(got, want, tol_max=5e-4, tol_l2=1e-3, qnorm=True):
    assert np.isfinite(got["audio"]).all()
    assert np.array_equal(got["emitted"], want["emitted"])
    mx, l2 = rel_errors(got["audio"], want["audio"])
    assert (mx <= tol_max).all() and (l2 <= tol_l2).all(), (mx.max(), l2.max())
    if qnorm:
        for key, w in want["qnorm"].items():
            assert np.abs(got["qnorm"][key] - w).max() <= 5e-4 * max(np.abs(w).max(), 1e-30) + 2e-6 * np.abs(want["audio"][key[0]]).max(), key
    for i, (g, w) in enumerate(zip(got["state"], want["state"])):
        np.testing.assert_allclose(g[0], w[0], rtol=0, atol=5e-4 * max(np.abs(w[0]).max(), 1e-30))
    return mx.max()


def _every_kind_scene(nb=16, sizes=(1, 64, 65, 300, 600)):
    """force-free and impulse buffers (explicit data, vertex hits on an idle object: DESC_DIRECT, a face hit), a Gaussian over
    several buffers, sustained AR scraping with a parameter update, a clearAllForces hole, a listener moving through FFAT
    maps with a zero weight on one mode, objects of 1 .. 600 modes (padding lanes, one to three waves at four modes per lane)"""
    rng = np.random.default_rng(4242)
    objs, evs = [], []
    for i, m in enumerate(sizes):
        lam = synth.eigenvalues(m, 1900 + i)
        shapes = synth.mode_shapes(m, 1900 + i)
        maps = synth.ffat_maps(lam, 1900 + i, dim=4, cell_size=0.01) if i == 3 else None
        if maps is not None:
            for mm in maps:
                mm["psi"] = np.array(mm["psi"], dtype=np.float64)
            maps[7]["psi"][:] = 0.0
        objs.append(ObjSpec(lam, shapes=shapes, maps=maps))
        nv = shapes.shape[1] // 3
        vns = synth.unit_normals(nb, 1900 + i)
        evs += [force_ev(0, i, vid=int(rng.integers(0, nv)), vn=vns[0]),
                force_ev(1, i, vid=int(rng.integers(0, nv)), vn=vns[1]),
                force_ev(2, i, data=rng.standard_normal(m) * 1e-3, force_type=1, width=1500.0),
                force_ev(5, i, clear=True),
                force_ev(6, i, data=rng.standard_normal(m) * 1e-3, force_type=2, start=True),
                force_ev(8, i, data=rng.standard_normal(m) * 1e-3, force_type=2),
                dict(t=9, obj=i, kind="arprm", a=[0.5, 0.3], sigma=0.004, mu=0.2),
                force_ev(10, i, force_type=2, end=True),
                force_ev(12, i, vids=[0, 1, 2], coords=[0.2, 0.3, 0.5], vn=vns[2]),
                force_ev(13, i, vid=int(rng.integers(0, nv)), vn=vns[3]),
                force_ev(14, i, data=rng.standard_normal(m) * 1e-3)]
        if maps is None:
            evs.append(dict(t=0, obj=i, kind="use_transfer", use=False))
        else:
            dirs = np.array([[1, .2, .3], [.2, 1, .3], [.2, .3, 1], [-1, .2, .3]], dtype=float)
            evs += [dict(t=b, obj=i, kind="listener", pos=0.5 * dirs[b % 4] / np.linalg.norm(dirs[b % 4])) for b in range(0, nb, 3)]
    return objs, evs


@pytest.mark.parametrize("form", [capi.FORM_BLOCK, capi.FORM_BLOCK_BF16])
@pytest.mark.parametrize("qnorm", [capi.QNORM_ALL, capi.QNORM_OFF])
@pytest.mark.parametrize("cb", [1, 3, 16])
def test_time_chunks_every_buffer_kind(cb, qnorm, form):
    nb = 16
    objs, evs = _every_kind_scene(nb)
    want = run_oracle(objs, evs, nb)
    for split in (None, [1, 4, 11]):
        got = run_engine(objs, evs, nb, split=split, form=form, qnorm=qnorm, time_chunks=cb)
        info = got["info"]
        n_launch = 1 if split is None else 3
        assert info["total_time_chunk_launches"] == info["total_block_launches"] == n_launch and info["total_sample_launches"] == 0
        _check(got, want, qnorm=qnorm != capi.QNORM_OFF)
        assert not got["emitted"][:, 5].any()


@pytest.mark.parametrize("mpl_scene", ["small", "large"])
def test_one_buffer_per_chunk_is_independent_of_the_cut(mpl_scene):
    """time_chunks = 1: every buffer starts from the scan's state, and the scan is one sequential pass whose arithmetic does
    not depend on where a step is cut into launches -- bit-identical audio, qnorm rows and state for any cut"""
    nb = 16
    objs, evs = _every_kind_scene(nb, sizes=(64, 300) if mpl_scene == "small" else (600, 1100))
    a = run_engine(objs, evs, nb, time_chunks=1)
    for split in ([1, 4, 11], [8, 8], [1] * 16):
        b = run_engine(objs, evs, nb, split=split, time_chunks=1)
        assert np.array_equal(a["audio"], b["audio"]), split
        for key in a["qnorm"]:
            assert np.array_equal(a["qnorm"][key], b["qnorm"][key]), (split, key)
        for sa, sb in zip(a["state"], b["state"]):
            assert np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1])


def _poisson_scene(n_obj, n_modes, nb, p_hit=0.233, seed=31):
    rng = np.random.default_rng(seed)
    objs, evs = [], []
    for i in range(n_obj):
        s = synth.seed_for(2, i)
        objs.append(ObjSpec(synth.eigenvalues(n_modes, s), shapes=synth.mode_shapes(n_modes, s)))
        nv = objs[-1].shapes.shape[1] // 3
        vns = synth.unit_normals(nb, s)
        for b in range(nb):
            if rng.random() < p_hit:
                evs.append(force_ev(b, i, vid=int(rng.integers(0, nv)), vn=vns[b]))
        evs.append(dict(t=0, obj=i, kind="use_transfer", use=False))
    return objs, evs


def test_policy_small_scenes_take_it_full_chips_and_single_buffers_do_not():
    """auto (time_chunks = 0): BASELINE configs[1] (1 x 512, 86 buffers) runs time-chunked; the same engine stepping ONE buffer
    (the real-time facade's call) does not; an engine pinned to the buffer-by-buffer walk (time_chunks < 0) never does; a launch
    that is mostly dense-profile buffers (sustained scraping) is cut in time too since round 5: the increments of its dense buffers
    (dense_increment_kernel) take their place in the scan"""
    nb = 86
    objs, evs = _poisson_scene(1, 512, nb)
    want = run_oracle(objs, evs, nb)
    got = run_engine(objs, evs, nb)
    assert got["info"]["total_time_chunk_launches"] == 1 and got["info"]["total_split_launches"] == 0
    err_tc = _check(got, want)
    one = run_engine(objs, evs, nb, split=[1] * nb)
    assert one["info"]["total_time_chunk_launches"] == 0
    _check(one, want)
    off = run_engine(objs, evs, nb, time_chunks=-1)
    assert off["info"]["total_time_chunk_launches"] == 0
    err_walk = _check(off, want)
    print(f"1 x 512 x 86: time-chunked {err_tc:.2e}, buffer-by-buffer {err_walk:.2e} of peak")
    # sustained scraping: every buffer dense
    rng = np.random.default_rng(5)
    sevs = [force_ev(0, 0, data=rng.standard_normal(512) * 1e-3, force_type=2, start=True), dict(t=0, obj=0, kind="use_transfer", use=False)]
    sevs += [force_ev(b, 0, data=rng.standard_normal(512) * 1e-3, force_type=2) for b in range(1, 12)]
    scr = run_engine(objs, sevs, 12)
    assert scr["info"]["total_time_chunk_launches"] == 1 and scr["info"]["total_dense_increment_launches"] == 1
    assert scr["info"]["last_time_chunk_shape"] == 1         # one mode per lane: the shape that holds the increment table
    _check(scr, run_oracle(objs, sevs, 12))
    walk = run_engine(objs, sevs, 12, time_chunks=-1)
    assert walk["info"]["total_time_chunk_launches"] == 0
    _check(walk, run_oracle(objs, sevs, 12))


def test_two_engines_of_one_process_with_different_settings():
    """pbso_engine_desc (ABI 4) selects kernels per ENGINE: a time-chunked engine, a pipeline-kernel engine and a
    buffer-by-buffer block engine side by side in one process, all inside the tolerance of the same oracle run"""
    from openpbso_amd import Engine, ForceMessage
    nb = 20
    objs, evs = _poisson_scene(3, 300, nb, p_hit=0.4)
    want = run_oracle(objs, evs, nb)
    engines = [Engine(time_chunks=2), Engine(bank_kernel=capi.BANK_PIPE), Engine(bank_kernel=capi.BANK_BLOCK, time_chunks=-1, direct_hits=-1)]
    try:
        for e in engines:
            for o in objs:
                e.add_object(o.lam, o.rho, o.alpha, o.beta, o.n_modes, o.shapes)
            e.finalize()
            for ev in sorted(evs, key=lambda x: x["t"]):
                if ev["kind"] == "force":
                    assert e.enqueue_force(ev["obj"], ForceMessage(vid=ev["vid"], vn=ev["vn"]), ev["t"])
                else:
                    e.set_use_transfer(ev["obj"], ev["use"], ev["t"])
        for e in engines:                            # interleaved: all three have work in flight at once
            e.step(nb)
        outs = [e.audio().copy() for e in engines]
        infos = [e.info() for e in engines]
    finally:
        for e in engines:
            e.close()
    assert infos[0]["total_time_chunk_launches"] == 1 and infos[1]["total_split_launches"] == 1
    assert infos[2]["total_time_chunk_launches"] == 0 and infos[2]["total_split_launches"] == 0 and infos[2]["last_step_forced_rows"] > 0
    for a in outs:
        mx, l2 = rel_errors(a, want["audio"])
        assert (mx <= 5e-4).all() and (l2 <= 1e-3).all()
    assert not np.array_equal(outs[0], outs[1])      # (different arithmetic: they are different kernels)


@pytest.mark.parametrize("time_chunks", [0, -1, 3])
def test_stream_hand_over_by_value_equals_the_event_path(time_chunks):
    """desc.stream_sync = 2: the preparation stream tells the bank stream that a launch is ready through a value in signal
    memory (hipStreamWaitValue64) instead of an event; 3: the next launch's preparation kernels wait for a value the bank
    kernel stores when it starts; 4 (round 5): the same gate on the host -- the submitting thread waits for that value in pinned
    host memory.  Ordering only: audio, qnorm rows and state are bit-identical to the
    event path over many launches, with the scan (K5) and without, with the combine kernel (a Gaussian force) in the batch."""
    nb = 24
    objs, evs = _every_kind_scene(nb)
    cut = [1, 2, 5, 4, 1, 1, 3, 7]
    a = run_engine(objs, evs, nb, split=cut, time_chunks=time_chunks, stream_sync=1)
    for mode in (2, 3, 4):                           # (3: events + the start gate -- the next preparation behind the bank's START; 4: that gate as a wait of the submitting thread)
        b = run_engine(objs, evs, nb, split=cut, time_chunks=time_chunks, stream_sync=mode, latency_path=-1)
        assert np.array_equal(a["audio"], b["audio"]) and np.array_equal(a["emitted"], b["emitted"])
        for key in a["qnorm"]:
            assert np.array_equal(a["qnorm"][key], b["qnorm"][key]), key
        for x, y in zip(a["state"], b["state"]):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    _check(b, run_oracle(objs, evs, nb))
    with pytest.raises(Exception):
        run_engine(objs[:1], [], 1, stream_sync=5)


@pytest.mark.parametrize("time_chunks", [0, -1, 3])
def test_forked_preparation_is_ordering_only(time_chunks, monkeypatch):
    """round 5: a launch with dense-profile rows may fork its preparation -- force profiles and dense increments on the preparation
    stream, projection + FFAT + combine on a second one, joined in front of the scan / the hand-over to the bank (Engine::step_chunk;
    by policy from 64 dense rows up, PBSO_PREP_SPLIT=2 forks whenever there is anything to fork, 0 never).  Ordering only: audio,
    qnorm rows and state bit-identical over many launches of every buffer kind, cut in time or not"""
    nb = 24
    objs, evs = _every_kind_scene(nb)
    cut = [1, 2, 5, 4, 1, 1, 3, 7]
    monkeypatch.setenv("PBSO_PREP_SPLIT", "0")
    a = run_engine(objs, evs, nb, split=cut, time_chunks=time_chunks, latency_path=-1)
    monkeypatch.setenv("PBSO_PREP_SPLIT", "2")
    b = run_engine(objs, evs, nb, split=cut, time_chunks=time_chunks, latency_path=-1)
    assert np.array_equal(a["audio"], b["audio"]) and np.array_equal(a["emitted"], b["emitted"])
    for key in a["qnorm"]:
        assert np.array_equal(a["qnorm"][key], b["qnorm"][key]), key
    for x, y in zip(a["state"], b["state"]):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    _check(b, run_oracle(objs, evs, nb))


def test_long_launches_are_gated_by_policy_and_equal_the_ungated_run():
    """launches of >= 256 buffers hold their preparation kernels behind the START of the previous launch's bank (the start gate:
    hipStreamWaitValue64 on the preparation stream) by policy; ordering only -- three 300-buffer steps of a Poisson scene,
    time-chunked and not, bit-identical to stream_sync = 1 (events, no gate) and inside the oracle's tolerance"""
    nb, steps = 300, 3
    objs, evs = _poisson_scene(3, 300, nb * steps, p_hit=0.3)
    want = run_oracle(objs, evs, nb * steps)
    for tc in (0, -1):
        a = run_engine(objs, evs, nb * steps, split=[nb] * steps, time_chunks=tc, chunk_buffers=512)
        b = run_engine(objs, evs, nb * steps, split=[nb] * steps, time_chunks=tc, chunk_buffers=512, stream_sync=1)
        assert np.array_equal(a["audio"], b["audio"]) and np.array_equal(a["emitted"], b["emitted"])
        for x, y in zip(a["state"], b["state"]):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
        assert a["info"]["total_time_chunk_launches"] == (steps if tc == 0 else 0)
        _check(a, want, qnorm=False)


def test_latency_path_is_ordering_only():
    """A step of at most four buffers submitted while the device is idle (step, read, step: the real-time facade) prepares on
    the bank's own stream (desc.latency_path, info.total_one_stream_launches): same kernels, same arguments -- bit-identical to
    the two-stream path, every buffer kind, and the preparation stream takes over again behind it"""
    nb = 16
    objs, evs = _every_kind_scene(nb)
    for cut, n_one in (([1] * nb, nb), ([4, 4, 8], 2), ([2, 8, 1, 1, 4], 4)):
        a = run_engine(objs, evs, nb, split=cut)
        b = run_engine(objs, evs, nb, split=cut, latency_path=-1)
        assert a["info"]["total_one_stream_launches"] == n_one and b["info"]["total_one_stream_launches"] == 0
        assert np.array_equal(a["audio"], b["audio"]) and np.array_equal(a["emitted"], b["emitted"])
        for key in a["qnorm"]:
            assert np.array_equal(a["qnorm"][key], b["qnorm"][key]), key
        for x, y in zip(a["state"], b["state"]):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    _check(a, run_oracle(objs, evs, nb))


@pytest.mark.parametrize("n_obj,n_modes", [(128, 512), (64, 256)])
def test_strong_scaling_shares_full_size(n_obj, n_modes):
    """the per-GPU share of BASELINE configs[3] on 8 GPUs (128 x 512) and configs[2]'s shape, 86 buffers, auto policy:
    time-chunked, sampled objects against the oracle"""
    nb = 86
    objs, evs = _poisson_scene(n_obj, n_modes, nb)
    got = run_engine(objs, evs, nb)
    assert got["info"]["total_time_chunk_launches"] == 1
    pick = list(range(0, n_obj, max(1, n_obj // 6)))
    want = run_oracle(objs, evs, nb, only=pick, threads=4)
    mx, l2 = rel_errors(got["audio"][pick], want["audio"])
    assert (mx <= 1e-4).all() and (l2 <= 1e-3).all(), (mx.max(), l2.max())


def test_scenes_larger_than_the_chip_are_cut_in_time_too():
    """1100 x 512 is 1.07 rounds of full-length workgroups for the walk (a second, nearly empty round: 1.68 ms per 86 buffers);
    the policy cuts it in time (1.22 ms).  A scene that fills the chip exactly (1024 x 512) keeps the walk.  Sampled objects
    against the oracle on both paths."""
    from openpbso_amd import Engine
    nb, n_obj = 86, 1100
    lam, shp = synth.eigenvalues(512, 77), synth.mode_shapes(512, 77)
    nv = shp.shape[1] // 3
    rng = np.random.default_rng(99)
    objs = [ObjSpec(lam, shapes=shp) for _ in range(n_obj)]
    evs = []
    for i in range(n_obj):
        vns = synth.unit_normals(nb, 300 + i % 7)
        evs += [force_ev(int(b), i, vid=int(rng.integers(0, nv)), vn=vns[b]) for b in np.nonzero(rng.random(nb) < 0.233)[0]]
        evs.append(dict(t=0, obj=i, kind="use_transfer", use=False))
    pick = [0, 333, 1024, 1099]
    want = run_oracle(objs, evs, nb, only=pick, threads=4)
    for tc, n_tc in ((0, 1), (-1, 0)):
        with Engine(time_chunks=tc) as eng:
            for o in objs:
                eng.add_object(o.lam, o.rho, o.alpha, o.beta, o.n_modes, o.shapes)
            eng.finalize()
            from openpbso_amd import ForceMessage
            for ev in evs:
                if ev["kind"] == "force":
                    assert eng.enqueue_force(ev["obj"], ForceMessage(vid=ev["vid"], vn=ev["vn"]), ev["t"])
                else:
                    eng.set_use_transfer(ev["obj"], ev["use"], ev["t"])
            eng.step(nb)
            got = eng.audio_rows(pick)
            assert eng.info()["total_time_chunk_launches"] == n_tc
        mx, l2 = rel_errors(got, want["audio"])
        assert (mx <= 1e-4).all() and (l2 <= 1e-3).all(), (tc, mx.max(), l2.max())
    with Engine() as eng:                              # exactly one round: the walk
        for o in objs[:1024]:
            eng.add_object(o.lam, o.rho, o.alpha, o.beta, o.n_modes, o.shapes)
        eng.finalize()
        eng.step(nb)
        eng.sync()
        assert eng.info()["total_time_chunk_launches"] == 0


def test_step_to_host_delivers_what_read_audio_returns():
    """pbso_step_to_host: the step's audio of all objects in pinned host memory, double-buffered on the device so that the copy
    of step k runs beside the bank of step k + 1 -- three steps into two alternating host buffers, each equal to what a second
    engine's pbso_read_audio returns for the same step, bit for bit"""
    from openpbso_amd import Engine, ForceMessage
    nb = 6
    objs, evs = _poisson_scene(5, 200, 3 * nb, p_hit=0.5)
    with Engine() as a, Engine() as b:
        for e in (a, b):
            for o in objs:
                e.add_object(o.lam, o.rho, o.alpha, o.beta, o.n_modes, o.shapes)
            e.finalize()
            for ev in sorted(evs, key=lambda x: x["t"]):
                if ev["kind"] == "force":
                    assert e.enqueue_force(ev["obj"], ForceMessage(vid=ev["vid"], vn=ev["vn"]), ev["t"])
                else:
                    e.set_use_transfer(ev["obj"], ev["use"], ev["t"])
        bufs = [a.host_buffer(nb), a.host_buffer(nb)]
        want = []
        for k in range(3):
            b.step(nb)
            want.append(b.audio().copy())
        got = []
        a.step_to_host(nb, bufs[0])
        a.step_to_host(nb, bufs[1])                  # (step 1's bank runs while step 0's samples travel)
        a.host_wait()
        got.append(bufs[0].copy())
        got.append(bufs[1].copy())
        a.step_to_host(nb, bufs[0])
        a.host_wait()
        got.append(bufs[0].copy())
        for k in range(3):
            assert np.array_equal(got[k], want[k]) and np.abs(want[k]).max() > 0, k
        # a PAGEABLE target takes the staged path (device buffer, then a copy on its own stream): same samples
        b.step(nb)
        plain = np.empty_like(got[0])
        a.step_to_host(nb, plain)
        a.host_wait()
        assert np.array_equal(plain, b.audio())


@pytest.mark.parametrize("qnorm", [capi.QNORM_ALL, capi.QNORM_OFF])
@pytest.mark.parametrize("shape", [1, 2, 4])
def test_dense_launches_cut_in_time_every_team_shape(shape, qnorm):
    """round 5: launches that are MOSTLY dense-profile buffers (sustained AutoregressiveForce contact, forces.h:107-128,
    modal_solver.h:222-240, with a parameter update, an end and free ringing; a Gaussian on the side) cut along the time axis:
    dense_increment_kernel evaluates what every dense buffer leaves in the state per unit gain, the scan takes g V where an impulse
    has (g amp) A^512 u, the bank steps the chunks in its forced block path -- for every team shape (time_chunk_shape), by policy
    and forced chunk lengths, objects of one team and of several, and with ONE buffer per chunk bit-identical for any cut"""
    nb = 24
    rng = np.random.default_rng(99 + shape)
    objs, evs = [], []
    for i, m in enumerate((64, 600, 1100)):
        objs.append(ObjSpec(synth.eigenvalues(m, 2300 + i), shapes=synth.mode_shapes(m, 2300 + i)))
        vns = synth.unit_normals(nb, 2300 + i)
        evs.append(dict(t=0, obj=i, kind="use_transfer", use=False))
        evs.append(force_ev(0, i, data=rng.standard_normal(m) * 1e-3, force_type=2, start=True))
        for b in range(1, 19):
            bary = rng.random(3)
            evs.append(force_ev(b, i, vids=rng.integers(0, synth.N_VERTS, 3), coords=bary / bary.sum(), vn=vns[b], force_type=2))
        evs.append(dict(t=7, obj=i, kind="arprm", a=[0.5, 0.3], sigma=0.004, mu=0.2))
        evs.append(force_ev(19, i, force_type=2, end=True))
        evs.append(force_ev(21, i, data=rng.standard_normal(m) * 1e-3, force_type=1, width=900.0))
    want = run_oracle(objs, evs, nb)
    kw = dict(qnorm=qnorm, form=capi.FORM_BLOCK, time_chunk_shape=shape, bank_kernel=capi.BANK_BLOCK)
    got = run_engine(objs, evs, nb, **kw)                                   # by policy (the pipeline kernel excluded)
    info = got["info"]
    assert info["total_time_chunk_launches"] == 1 and info["total_dense_increment_launches"] == 1 and info["last_time_chunk_shape"] == shape
    _check(got, want, qnorm=qnorm != capi.QNORM_OFF)
    for cb in (1, 5):
        a = run_engine(objs, evs, nb, time_chunks=cb, **kw)
        _check(a, want, qnorm=qnorm != capi.QNORM_OFF)
        if cb == 1:
            b = run_engine(objs, evs, nb, split=[3, 9, 12], time_chunks=cb, **kw)
            assert b["info"]["total_dense_increment_launches"] == 3
            assert np.array_equal(a["audio"], b["audio"])
            for sa, sb in zip(a["state"], b["state"]):
                assert np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1])


@pytest.mark.parametrize("qnorm", [capi.QNORM_ALL, capi.QNORM_OFF])
@pytest.mark.parametrize("cb", [2, 3, 7, 16])
def test_scan_cut_along_the_time_axis_equals_the_serial_scan(cb, qnorm):
    """round 5: the scan of chunk-start states itself cut in time (kernels_scan.hip, iir_scan_seg_kernel: one wave per chunk scans its
    buffers as an affine map of the state -- the zero-state response and the images of the two basis vectors --, the maps are
    composed in LDS).  Every buffer kind (impulses, direct hits, a Gaussian, sustained AR contact with its dense increments, a
    clear, listener moves with a zero weight), chunk lengths that give 1 .. 8 chunks, launch cuts: against the oracle, and against
    the serial scan of the same engine settings within rounding"""
    nb = 16
    objs, evs = _every_kind_scene(nb)
    want = run_oracle(objs, evs, nb)
    n_chunks = (nb + cb - 1) // cb
    for split in (None, [9, 7]):
        seg = run_engine(objs, evs, nb, split=split, qnorm=qnorm, time_chunks=cb, scan_kernel=2)
        ser = run_engine(objs, evs, nb, split=split, qnorm=qnorm, time_chunks=cb, scan_kernel=1)
        launches = 1 if split is None else 2
        assert seg["info"]["total_time_chunk_launches"] == launches and ser["info"]["total_segmented_scans"] == 0
        if split is None:
            assert seg["info"]["total_segmented_scans"] == (1 if 2 <= n_chunks <= 8 else 0)
        _check(seg, want, qnorm=qnorm != capi.QNORM_OFF)
        _check(ser, want, qnorm=qnorm != capi.QNORM_OFF)
        mx, _ = rel_errors(seg["audio"], ser["audio"])
        assert mx.max() <= 2e-5, mx.max()
    # by policy: small scans of chunks of several buffers take it, one buffer per chunk keeps the serial scan (bit-identical for any cut)
    pol = run_engine(objs, evs, nb, qnorm=qnorm, time_chunks=cb)
    assert pol["info"]["total_segmented_scans"] == (1 if 2 <= n_chunks <= 8 else 0)      # (5 objects x 12 tiles x n_chunks waves: a few hundred)
    one = run_engine(objs, evs, nb, qnorm=qnorm, time_chunks=1, split=[8, 8])
    assert one["info"]["total_segmented_scans"] == 0
