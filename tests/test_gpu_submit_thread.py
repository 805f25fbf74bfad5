"""The second submitting thread (pbso_engine_desc::submit_thread, round 6; openpbso_amd/csrc/submit_queue.h).

With it pbso_step RECORDS a launch's stream calls and returns; a worker thread makes them while the caller plans the next launch.
Same kernels, same arguments, same order in the same streams: everything a run produces is BIT-IDENTICAL to the run without the
thread -- checked here on steps issued back to back without reading anything in between (the queue several launches deep), with
messages enqueued between the steps, for the kinds of launch the engine knows (whole-buffer, time chunks, the pipeline kernel,
one-buffer steps with fused preparation, several launches per step), and through the facade's pinned-host step."""
import numpy as np
import pytest
import torch

from openpbso_amd import ForceMessage, capi, synth
from openpbso_amd.solver import Engine
from tests.scenarios import B

pytestmark = pytest.mark.gpu


def _scene(n_obj, n_modes, seed):
    objs = []
    for i in range(n_obj):
        lam = synth.eigenvalues(n_modes, seed + i)
        objs.append((lam, synth.mode_shapes(n_modes, seed + i), synth.ffat_maps(lam, seed + i, dim=4, cell_size=0.01)))
    return objs


def _run(objs, n_steps, nb, submit_thread, seed, read_every=0, **select):
    """n_steps steps of nb buffers each, into their own device rows, nothing read until the end (unless read_every)"""
    n_modes = len(objs[0][0])
    eng = Engine(submit_thread=submit_thread, **select)
    try:
        for lam, shapes, maps in objs:
            oid = eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA, n_modes, shapes)
            eng.set_ffat_maps(oid, maps)
        eng.finalize()
        N = len(objs)
        out = torch.zeros((n_steps, N, nb * B), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        rng = np.random.default_rng(seed)
        vns = synth.unit_normals(n_steps * nb + 1, seed)
        peeks = []
        t = 0
        for o in range(N):
            eng.set_use_transfer(o, True, 0)
        # one object scrapes for the whole run (AR force, hits projected on the device)
        assert eng.enqueue_force(0, ForceMessage(forceType=2, sustainedForceStart=True), 0)
        for s in range(n_steps):
            for b in range(nb):
                if N > 1:
                    o = 1 + int(rng.integers(0, N - 1))
                    if rng.random() < 0.6:
                        assert eng.enqueue_force(o, ForceMessage(vid=int(rng.integers(0, synth.N_VERTS)), vn=vns[t]), t)
                    if rng.random() < 0.15:
                        assert eng.enqueue_force(o, ForceMessage(data=rng.standard_normal(n_modes) * 1e-3, forceType=1, gaussianWidth=1200.0), t)
                if rng.random() < 0.5:
                    assert eng.enqueue_force(0, ForceMessage(vid=int(rng.integers(0, synth.N_VERTS)), vn=vns[t], forceType=2), t)
                if rng.random() < 0.3:
                    eng.compute_transfer(int(rng.integers(0, N)), 0.02 * rng.standard_normal(3) + [0.0, 0.0, 0.05], t)
                t += 1
            if s == 2:
                eng.enqueue_arprm(0, [0.6, 0.2], 0.003, 0.1, t - nb)
            eng.step(nb, into=out[s].data_ptr())
            if read_every and s % read_every == read_every - 1:
                peeks.append(eng.audio_rows(np.arange(N)).copy())      # (an entry point that reads: waits for the worker first)
        eng.sync()
        info = eng.info()
        res = dict(audio=out.cpu().numpy(), state=[eng.state(i) for i in range(N)], latest=[eng.latest_transfer(i) for i in range(N)],
                   emitted=eng.emitted().copy(), peeks=peeks, info=info)
        return res
    finally:
        eng.close()


def _same(a, b):
    assert np.isfinite(a["audio"]).all() and np.abs(a["audio"]).max() > 0
    assert np.array_equal(a["audio"], b["audio"])
    assert np.array_equal(a["emitted"], b["emitted"])
    for x, y in zip(a["state"], b["state"]):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    for x, y in zip(a["latest"], b["latest"]):
        assert np.array_equal(x, y)
    assert len(a["peeks"]) == len(b["peeks"])
    for x, y in zip(a["peeks"], b["peeks"]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("name,n_obj,n_modes,n_steps,nb,select", [
    ("real-time steps, fused preparation", 4, 96, 40, 1, {}),
    ("real-time steps, separate preparation", 4, 96, 24, 1, dict(fuse_short_launches=-1)),
    ("short steps through time chunks", 6, 128, 12, 8, {}),
    ("several launches per step", 6, 128, 6, 16, dict(chunk_buffers=4)),
    ("whole-buffer bank only", 5, 200, 8, 6, dict(time_chunks=-1)),
    ("the pipeline kernel", 3, 320, 8, 6, dict(bank_kernel=2)),
])
def test_steps_with_the_submitting_thread_are_bit_identical(name, n_obj, n_modes, n_steps, nb, select):
    objs = _scene(n_obj, n_modes, 4100)
    on = _run(objs, n_steps, nb, 1, 77, **select)
    off = _run(objs, n_steps, nb, 0, 77, **select)
    _same(on, off)
    assert on["info"]["buffers_done"] == off["info"]["buffers_done"] == n_steps * nb


def test_reads_between_steps_wait_for_the_worker():
    """an entry point that reads results (pbso_read_audio_rows here) between run-ahead steps sees the step it follows"""
    objs = _scene(4, 96, 4200)
    on = _run(objs, 18, 2, 1, 78, read_every=3)
    off = _run(objs, 18, 2, 0, 78, read_every=3)
    _same(on, off)
    for k, p in enumerate(on["peeks"]):
        assert np.array_equal(p, on["audio"][3 * k + 2])


def test_flush_orders_the_callers_own_stream_work():
    """pbso_step_into + pbso_flush: a copy the CALLER puts on the engine's stream behind a step sees that step's audio"""
    objs = _scene(3, 96, 4300)
    n_modes = 96
    for thread in (1, 0):
        stream = torch.cuda.Stream()
        eng = Engine(submit_thread=thread, stream=stream.cuda_stream)
        try:
            for lam, shapes, maps in objs:
                eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA, n_modes, shapes)
            eng.finalize()
            a = torch.zeros((3, 4 * B), dtype=torch.float32, device="cuda")
            copies = []
            torch.cuda.synchronize()
            for s in range(10):
                assert eng.enqueue_force(s % 3, ForceMessage(vid=s % synth.N_VERTS, vn=np.array([0.0, 0.6, 0.8])), 4 * s)
                eng.step(4, into=a.data_ptr())
                eng.flush()
                with torch.cuda.stream(stream):
                    copies.append(a.clone())
            eng.sync()
            got = torch.stack(copies).cpu().numpy()
        finally:
            eng.close()
        if thread:
            with_thread = got
    assert np.abs(got).max() > 0 and np.array_equal(with_thread, got)


def test_group_engines_and_the_host_gate_refuse_or_ignore_the_thread():
    with pytest.raises(Exception, match="stream_sync = 4"):
        Engine(submit_thread=1, stream_sync=4)


@pytest.mark.parametrize("seed", range(24))
def test_random_scripts_with_the_submitting_thread_match_the_oracle(seed, monkeypatch):
    """the fuzz suite's scripts (tests/test_gpu_fuzz.py: arbitrary interleavings of forces, hits, listener moves, clears, AR updates) on
    engines with the thread, on-device hit projection, launches cut at random lengths, the kernels the fuzz suite rotates through"""
    from tests.test_gpu_fuzz import _run_seed
    rng = np.random.default_rng(55000 + seed)
    monkeypatch.setenv("PBSO_CHUNK_BUFFERS", str(int(rng.choice([1, 2, 5, 128]))))
    kw = dict(submit_thread=1, time_chunks=int(rng.choice([0, -1, 1, 3])), qnorm=int(rng.choice([capi.QNORM_ALL, capi.QNORM_CLOSED])),
              form=int(rng.choice([capi.FORM_BLOCK, capi.FORM_BLOCK_BF16, capi.FORM_VELOCITY])))
    _run_seed(seed + 200000, [5, 64, 200, 1100], kw, projected_hits=True)


def test_transfer_rows_grow_while_the_worker_lags():
    """the planner grows the pool of transfer rows (and re-cuts its per-set scratch areas) when a step fires more listener events than any
    before it; with the thread that must wait for the recorded launches of the steps before -- they hold the old block's address and
    the keep-copy has to come behind them.  Steps of 1, 2, 4 ... 64 buffers with a listener move per object and buffer, issued back to back"""
    objs = _scene(6, 96, 4400)
    n_modes = 96
    # (a first step of 64 buffers without listener events sizes every per-step buffer: a buffer that grows waits for the worker by
    #  itself, GROWTRY, and would hide what this test is after)
    steps = [64, 1, 2, 4, 8, 16, 32, 64, 3, 64]
    total = sum(steps)
    pos = [synth.listener_path(total - 64, radius=0.3 + 0.03 * i) for i in range(len(objs))]

    def run(thread):
        with Engine(submit_thread=thread) as eng:
            for lam, shapes, maps in objs:
                oid = eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA, n_modes, shapes)
                eng.set_ffat_maps(oid, maps)
            eng.finalize()
            N = len(objs)
            out = [torch.zeros((N, nb * B), dtype=torch.float32, device="cuda") for nb in steps]
            torch.cuda.synchronize()
            rng = np.random.default_rng(9)
            for o in range(N):
                assert eng.enqueue_force(o, ForceMessage(data=rng.standard_normal(n_modes) * 1e-3), 0)
            # (the whole script up front, stamped: nothing between two steps but the next step's plan -- the worker is still making
            #  the calls of step k when the planner of step k + 1 finds its rows too few)
            done = 0
            for k, nb in enumerate(steps):
                for o in range(N):
                    if k % 2:
                        assert eng.enqueue_force(o, ForceMessage(vid=int(rng.integers(0, synth.N_VERTS)), vn=np.array([0.6, 0.0, 0.8])), done)
                done += nb
            for o in range(N):
                # (a position every third buffer: most launches START with the row the launch before them left in force)
                # (single stamped calls: they wait in the object's pending list, and a plan counts the ones due in ITS buffers -- a path
                #  array is counted whole by the first plan that sees it)
                for j, t in enumerate(range(65, total, 3)):
                    eng.compute_transfer(o, pos[o][j], t)
            ptrs = [t.data_ptr() for t in out]
            for k, nb in enumerate(steps):
                eng.step(nb, into=ptrs[k])
            eng.sync()
            return [t.cpu().numpy() for t in out], [eng.latest_transfer(o).copy() for o in range(N)]

    a_audio, a_latest = run(1)
    b_audio, b_latest = run(0)
    for k in range(len(steps)):
        assert np.array_equal(a_audio[k], b_audio[k]), k
    for x, y in zip(a_latest, b_latest):
        assert np.array_equal(x, y)
    assert np.abs(a_audio[-1]).max() > 0
