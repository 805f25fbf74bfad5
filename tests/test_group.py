"""The device group (include/openpbso_amd.h "device group"; SURVEY.md 8(b) device list, 8(e)): one engine per GPU, objects
sharded by the sum of their modes, RCCL called from the C++ library.  CPU part: the entry points exist and fail cleanly
without a GPU.  GPU part (one device, all this pool has): the group's three gather modes equal the single engine bit for
bit (in-place self-gather, root, on-device object mix), the shards equal the Python rule the gloo tests use, the error paths."""
import ctypes as C

import numpy as np
import pytest

from openpbso_amd import capi, synth
from openpbso_amd.distributed import shard_by_modes


def test_group_symbols_and_clean_failure_without_a_gpu():
    lib = capi.lib()
    for name in capi.EXPORTS:
        assert name.startswith("pbso_") and getattr(lib, name)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the failure path is the no-device one")
    from openpbso_amd.group import Group
    from openpbso_amd.solver import PbsoError
    with pytest.raises(PbsoError) as ei:
        Group([0])
    assert ei.value.status == capi.ERR_HIP                   # no CPU fallback, and no crash
    d = capi.GroupDesc()
    h = C.c_void_p()
    assert lib.pbso_group_create(C.byref(d), C.byref(h)) == capi.ERR_INVALID      # abi_version 0
    assert b"abi_version" in lib.pbso_group_last_error(h)
    lib.pbso_group_destroy(h)


@pytest.mark.parametrize("world", [1, 2, 3, 8, 13])
def test_c_sharding_rule_equals_the_python_one(world):
    """pbso_shard_by_modes (what pbso_group_plan uses) is a port of openpbso_amd.distributed.shard_by_modes (what the gloo tests
    and bench.py's launcher-less fallback use): the same cuts for equal, ragged and degenerate jobs"""
    lib = capi.lib()
    rng = np.random.default_rng(world)
    for modes in ([512] * 1024, list(rng.integers(1, 4096, 37)), [5], [0, 0, 7, 0], list(rng.integers(0, 3, 11)), [], [4096] * 8,
                  list(rng.integers(64, 600, 1000))):
        m = np.ascontiguousarray(modes, dtype=np.int32)
        cuts = np.zeros(world + 1, dtype=np.int32)
        assert lib.pbso_shard_by_modes(m.ctypes.data_as(C.POINTER(C.c_int)), m.size, world, cuts.ctypes.data_as(C.POINTER(C.c_int))) == capi.OK
        want = [shard_by_modes(modes, world, r) for r in range(world)]
        assert [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)] == [tuple(w) for w in want], (world, modes[:8])
    bad = np.array([3, -1], dtype=np.int32)
    cuts = np.zeros(3, dtype=np.int32)
    assert lib.pbso_shard_by_modes(bad.ctypes.data_as(C.POINTER(C.c_int)), 2, 2, cuts.ctypes.data_as(C.POINTER(C.c_int))) == capi.ERR_INVALID


@pytest.mark.gpu
def test_group_of_one_device_equals_the_engine_bit_for_bit():
    """in-place all-gather on a one-rank group is the engine's own buffer (nothing to send), gather-to-root likewise, and the
    mix is the object sum of the engine's audio in object order -- against pbso_step on a second engine fed the same messages"""
    from openpbso_amd import Engine, ForceMessage
    from openpbso_amd.group import Group
    n_obj, n_modes, nb = 12, 200, 6
    rng = np.random.default_rng(12)
    lams = [synth.eigenvalues(n_modes, 700 + i) for i in range(n_obj)]
    shapes = [synth.mode_shapes(n_modes, 700 + i) for i in range(n_obj)]
    nv = shapes[0].shape[1] // 3
    hits = [(int(rng.integers(0, n_obj)), int(rng.integers(0, nv)), int(rng.integers(0, 2 * nb))) for _ in range(40)]
    vns = synth.unit_normals(len(hits), 3)
    with Engine() as eng, Group([0]) as grp:
        grp.plan([n_modes] * n_obj)
        assert grp.span(0) == (0, n_obj)
        for i in range(n_obj):
            eng.add_object(lams[i], synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes[i])
            grp.add_object(i, lams[i], synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes[i])
        eng.finalize()
        grp.finalize()
        ge = grp.engine(0)
        for i in range(n_obj):
            eng.set_use_transfer(i, False)
            ge.set_use_transfer(i, False)
        for (o, v, t), vn in zip(sorted(hits, key=lambda h: h[2]), vns):
            m = ForceMessage(vid=v, vn=vn)
            assert eng.enqueue_force(o, m, t) and grp.enqueue_force(o, m, t)
        for k in range(2):                                    # two steps: both gather targets are used
            eng.step(nb)
            want = eng.audio()
            grp.step(nb)
            grp.gather(capi.GATHER_ALL)
            got = grp.result(0)
            assert got.shape == (n_obj, nb * 513) and np.array_equal(got, want), k
            grp.gather(capi.GATHER_ROOT)
            assert np.array_equal(grp.result(0), want)
            grp.gather(capi.GATHER_MIX)
            mix = grp.result(0)
            assert mix.shape == (1, nb * 513)
            ref = want.astype(np.float64).sum(axis=0)
            assert np.abs(mix[0] - ref).max() <= 4e-6 * np.abs(ref).max()         # f32 sum of 12 rows
            assert np.abs(want).max() > 0
        assert ge.info()["buffers_done"] == 2 * nb


@pytest.mark.gpu
def test_group_error_paths():
    from openpbso_amd.group import Group
    from openpbso_amd.solver import PbsoError
    with pytest.raises(PbsoError):
        Group([0, 0])                                        # one rank per GPU
    with pytest.raises(PbsoError):
        Group([0], world_size=2)                             # a job of several processes needs the shared id
    with Group([0]) as g:
        with pytest.raises(PbsoError):
            g.finalize()                                     # before plan
        g.plan([64, 64])
        lam = synth.eigenvalues(64, 1)
        with pytest.raises(PbsoError):
            g.add_object(1, lam, synth.RHO, synth.ALPHA, synth.BETA)      # ascending order within a rank
        g.add_object(0, lam, synth.RHO, synth.ALPHA, synth.BETA)
        with pytest.raises(PbsoError):
            g.finalize()                                     # object 1 missing
        g.add_object(1, lam, synth.RHO, synth.ALPHA, synth.BETA)
        g.finalize()
        with pytest.raises(PbsoError):
            g.gather(capi.GATHER_ALL)                        # before a step
        g.step(2)
        with pytest.raises(PbsoError):
            g.gather(7)
        g.gather(capi.GATHER_ALL)
        assert g.result(0).shape == (2, 2 * 513)


def _scene(modes, nb_total, seed):
    rng = np.random.default_rng(seed)
    lams = [synth.eigenvalues(m, 900 + i) for i, m in enumerate(modes)]
    shapes = [synth.mode_shapes(m, 900 + i) for i, m in enumerate(modes)]
    nv = shapes[0].shape[1] // 3
    hits = sorted(((int(rng.integers(0, len(modes))), int(rng.integers(0, nv)), int(rng.integers(0, nb_total))) for _ in range(6 * len(modes))),
                  key=lambda h: h[2])
    return lams, shapes, hits, synth.unit_normals(len(hits), seed)


def _feed(eng, grp, modes, lams, shapes, hits, vns):
    from openpbso_amd import ForceMessage
    grp.plan(modes)
    for i in range(len(modes)):
        eng.add_object(lams[i], synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes[i])
        grp.add_object(i, lams[i], synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes[i])
    eng.finalize()
    grp.finalize()
    for i in range(len(modes)):
        eng.set_use_transfer(i, False)
    for r in grp.local_ranks():
        lo, hi = grp.span(r)
        for l in range(hi - lo):
            grp.engine(r).set_use_transfer(l, False)
    for (o, v, t), vn in zip(hits, vns):
        m = ForceMessage(vid=v, vn=vn)
        assert eng.enqueue_force(o, m, t) and grp.enqueue_force(o, m, t)


def _check_gathered(grp, got, want, world, rank_label):
    spans = [grp.span(r) for r in range(world)]
    cmax = max(hi - lo for lo, hi in spans)
    assert got.shape == (world * cmax, want.shape[1]), rank_label
    for r, (lo, hi) in enumerate(spans):
        assert np.array_equal(got[r * cmax:r * cmax + (hi - lo)], want[lo:hi]), (rank_label, r)
        assert not got[r * cmax + (hi - lo):(r + 1) * cmax].any(), ("padding rows of a ragged shard must be silent", rank_label, r)


@pytest.mark.gpu
def test_one_rank_group_runs_every_rccl_call_on_one_gpu():
    """PBSO_GROUP_RCCL_ALWAYS: librccl is dlopen'ed, the eleven symbols resolved, ncclGetUniqueId + ncclCommInitRank(1 rank) build a
    communicator, and the gathers ISSUE their collectives -- the in-place ncclAllGather, an ncclSend / ncclRecv pair to itself
    (the received rows are the result), ncclAllReduce -- behind the same stream / event ordering a larger job uses; results equal
    the single engine (modal_solver.h:100-126: objects are independent)"""
    from openpbso_amd import Engine
    from openpbso_amd.group import Group, unique_id
    uid = unique_id()
    assert len(uid) == capi.GROUP_ID_BYTES and any(uid)
    modes = [200] * 6 + [64, 333]
    nb = 5
    lams, shapes, hits, vns = _scene(modes, 3 * nb, 21)
    for given_id in (uid, None):                             # the id a launcher handed over / one the group asks for itself
        with Engine() as eng, Group([0], transport=capi.GROUP_RCCL_ALWAYS, unique_id=given_id) as grp:
            _feed(eng, grp, modes, lams, shapes, hits, vns)
            for k in range(3):                               # three steps: both gather targets, and the first one again
                eng.step(nb)
                want = eng.audio()
                grp.step(nb)
                grp.gather(capi.GATHER_ALL)
                assert np.array_equal(grp.result(0), want), k
                grp.gather(capi.GATHER_ROOT)
                assert np.array_equal(grp.result(0), want), k     # (the rows that came back through ncclRecv)
                grp.gather(capi.GATHER_MIX)
                mix = grp.result(0)
                ref = want.astype(np.float64).sum(axis=0)
                assert mix.shape == (1, nb * 513) and np.abs(mix[0] - ref).max() <= 4e-6 * np.abs(ref).max()
                assert np.abs(want).max() > 0


@pytest.mark.gpu
def test_one_rank_gathers_at_the_headline_size():
    """the gathers at the size the headline line steps -- 1024 objects x 860 buffers x 513 floats = 1.8 GB per
    target -- on a one-rank communicator (PBSO_GROUP_RCCL_ALWAYS): the in-place ncclAllGather of the whole target, and the
    gather-to-root whose ncclSend / ncclRecv pair to itself goes in pieces of 256 MB (one piece of 1.8 GB came back wrong on RCCL
    2.26.6, group.cpp: P2P_MAX).  pbso_group_read_result against pbso_read_audio of the rank's engine, bit for bit.  (What a
    one-GPU box can show: the calls, their sizes and the events around them; nothing crosses a link.)"""
    from openpbso_amd import ForceMessage
    from openpbso_amd.group import Group
    n_obj, M, nb = 1024, 512, 860
    rng = np.random.default_rng(3)
    with Group([0], transport=capi.GROUP_RCCL_ALWAYS) as grp:
        grp.plan([M] * n_obj)
        for i in range(n_obj):
            grp.add_object(i, synth.eigenvalues(M, 3000 + i), synth.RHO, synth.ALPHA, synth.BETA)
        grp.finalize()
        eng = grp.engine(0)
        for i in range(n_obj):
            eng.set_use_transfer(i, False)
            for t in (0, 400, 859):
                assert grp.enqueue_force(i, ForceMessage(data=rng.standard_normal(M) * 1e-3), t)
        grp.step(nb)
        want = eng.audio().copy()
        assert want.shape == (n_obj, nb * 513) and np.isfinite(want).all() and np.abs(want[-1, -513:]).max() > 0
        grp.gather(capi.GATHER_ALL)
        assert np.array_equal(grp.result(0), want)
        grp.gather(capi.GATHER_ROOT)
        assert np.array_equal(grp.result(0), want)              # (the rows that came back through eight ncclRecv pieces)


@pytest.mark.gpu
@pytest.mark.parametrize("world,modes", [(2, [4096, 64, 64, 64]), (3, [300] * 7), (8, [512] * 19 + [64] * 5), (3, [128, 128]), (4, [64] * 4)])
def test_loopback_ranks_on_one_device_equal_the_single_engine(world, modes):
    """PBSO_GROUP_LOOPBACK: the job's ranks as engines of ONE process on ONE device, the collectives as device copies -- everything
    the group does around them is the product's code: shards by the sum of modes (ragged: [4096, 64, 64, 64] on two ranks is 1 + 3
    objects; two objects on three ranks leave a rank empty), padding rows of the smaller shards SILENT (also when a later step is
    shorter than an earlier one and the rows move), slice offsets, both gather targets and their events, gather-to-root, the mixed
    stream.  Every rank's result against pbso_step on one engine fed the same messages, bit for bit (mix: f32 sums in another order)"""
    from openpbso_amd import Engine
    from openpbso_amd.group import Group
    steps = [4, 4, 2, 5]
    lams, shapes, hits, vns = _scene(modes, sum(steps), world)
    with Engine() as eng, Group([0] * world, transport=capi.GROUP_LOOPBACK) as grp:
        _feed(eng, grp, modes, lams, shapes, hits, vns)
        assert [grp.span(r) for r in range(world)] == [tuple(shard_by_modes(modes, world, r)) for r in range(world)]
        for k, nb in enumerate(steps):
            eng.step(nb)
            want = eng.audio()
            grp.step(nb)
            grp.gather(capi.GATHER_ALL)
            for r in range(world):
                _check_gathered(grp, grp.result(r), want, world, f"all, step {k}, rank {r}")
            grp.gather(capi.GATHER_ROOT)
            _check_gathered(grp, grp.result(0), want, world, f"root, step {k}")
            for r in range(1, world):
                lo, hi = grp.span(r)
                assert np.array_equal(grp.result(r)[:hi - lo], want[lo:hi]), (k, r)
            grp.gather(capi.GATHER_MIX)
            ref = want.astype(np.float64).sum(axis=0)
            for r in range(world):
                mix = grp.result(r)
                assert mix.shape == (1, nb * 513) and np.abs(mix[0] - ref).max() <= 1e-5 * np.abs(ref).max(), (k, r)
        assert np.abs(want).max() > 0


def test_loopback_and_transport_arguments_are_checked():
    """descriptor validation needs no GPU work beyond device discovery; without a GPU the group fails cleanly first"""
    import torch
    from openpbso_amd.group import Group
    from openpbso_amd.solver import PbsoError
    with pytest.raises(PbsoError):
        Group([0], transport=7)
    with pytest.raises(PbsoError):
        Group([0, 0], world_size=3, transport=capi.GROUP_LOOPBACK)     # every rank of the job lives in the one process
    if torch.cuda.is_available():
        with pytest.raises(PbsoError):
            Group([0, 0])                                    # the product's transport: one rank per GPU
