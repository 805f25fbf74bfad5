"""N > 1 path on CPU: world_size 2, gloo.  Objects shard across ranks with no
data-path collective; the audio gather restores global object order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openpbso_amd.distributed import gather_audio, shard_by_modes, shard_range


def test_shard_range_partitions_objects():
    for n in (0, 1, 7, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_by_modes_balances_the_sum_of_modes():
    """contiguous blocks balanced by sum of M, identical to shard_range for equal objects"""
    for w in (1, 2, 3, 8):
        for n in (1, 7, 1024):
            spans = [shard_by_modes([512] * n, w, r) for r in range(w)]
            assert spans == [shard_range(n, w, r) for r in range(w)] or max(hi - lo for lo, hi in spans) - min(
                hi - lo for lo, hi in spans) <= 1
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    rng = np.random.default_rng(5)
    modes = rng.choice([64, 256, 512, 4096], 300).tolist()
    for w in (2, 4, 8):
        spans = [shard_by_modes(modes, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == len(modes) and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        loads = [sum(modes[lo:hi]) for lo, hi in spans]
        assert max(loads) - min(loads) <= 2 * 4096            # within one largest object of each other
        by_count = [sum(modes[slice(*shard_range(len(modes), w, r))]) for r in range(w)]
        assert max(loads) <= max(by_count)                    # never worse than balancing the object count
    assert [shard_by_modes([4096, 64, 64, 64], 2, r) for r in range(2)] == [(0, 1), (1, 4)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_objects, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle_py as orc
        from openpbso_amd import synth
        lo, hi = shard_range(n_objects, world, rank)
        # each rank steps ITS objects (here with the CPU oracle standing in for the
        # engine: no GPU in this test) -- seeds are global object ids, as in bench.py
        rows = []
        for obj in range(lo, hi):
            lam = synth.eigenvalues(24, synth.seed_for(4, obj))
            s = orc.Solver(lam, synth.RHO, synth.ALPHA, synth.BETA)
            s.set_use_transfer(False)
            s.enqueue_force(np.full(24, 1e-3 * (obj + 1)))
            rows.append(np.concatenate([s.step()[0] for _ in range(2)]))
        local = torch.tensor(np.array(rows), dtype=torch.float32).reshape(hi - lo, -1)
        counts = [shard_range(n_objects, world, r)[1] - shard_range(n_objects, world, r)[0] for r in range(world)]
        full = gather_audio(local, counts)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_objects", [4, 5])
def test_two_rank_gather_restores_object_order(tmp_path, n_objects):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_objects, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    assert a.shape == (n_objects, 2 * 513) and np.array_equal(a, b)
    # row i must be object i: the impulse amplitude scales with (i + 1)
    from oracle import oracle_py as orc
    from openpbso_amd import synth
    for obj in range(n_objects):
        lam = synth.eigenvalues(24, synth.seed_for(4, obj))
        s = orc.Solver(lam, synth.RHO, synth.ALPHA, synth.BETA)
        s.set_use_transfer(False)
        s.enqueue_force(np.full(24, 1e-3 * (obj + 1)))
        want = np.concatenate([s.step()[0] for _ in range(2)]).astype(np.float32)
        assert np.array_equal(a[obj], want)


def _engine_worker(rank, world, port, n_objects, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        from openpbso_amd import Engine, ForceMessage, capi, synth
        # the shards the C ABI's device group would own (pbso_shard_by_modes), the engine stepping this rank's share
        modes = [24 + 8 * (i % 3) for i in range(n_objects)]
        m = np.ascontiguousarray(modes, dtype=np.int32)
        cuts = np.zeros(world + 1, dtype=np.int32)
        assert capi.lib().pbso_shard_by_modes(m.ctypes.data_as(C.POINTER(C.c_int)), m.size, world, cuts.ctypes.data_as(C.POINTER(C.c_int))) == capi.OK
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        with Engine() as eng:
            for obj in range(lo, hi):
                eng.add_object(synth.eigenvalues(modes[obj], synth.seed_for(4, obj)), synth.RHO, synth.ALPHA, synth.BETA)
            eng.finalize()
            for k, obj in enumerate(range(lo, hi)):
                eng.set_use_transfer(k, False)
                assert eng.enqueue_force(k, ForceMessage(data=np.full(modes[obj], 1e-3 * (obj + 1))), 0)
            eng.step(2)
            local = torch.tensor(eng.audio(), dtype=torch.float32).reshape(hi - lo, -1)
        full = gather_audio(local, [int(cuts[r + 1] - cuts[r]) for r in range(world)])
        np.save(os.path.join(out_dir, f"erank{rank}.npy"), full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_gather_steps_the_engine(tmp_path):
    """the same N > 1 path with the ENGINE in each rank (two processes share the one GPU of the box; gloo carries the gather,
    RCCL needs a GPU per rank): ragged shards cut by the C ABI's pbso_shard_by_modes, object order restored, every row
    inside the stated tolerance of the oracle"""
    n_objects = 7
    port = _free_port()
    mp.spawn(_engine_worker, args=(2, port, n_objects, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "erank0.npy")
    b = np.load(tmp_path / "erank1.npy")
    assert a.shape == (n_objects, 2 * 513) and np.array_equal(a, b)
    from oracle import oracle_py as orc
    from openpbso_amd import synth
    for obj in range(n_objects):
        nm = 24 + 8 * (obj % 3)
        s = orc.Solver(synth.eigenvalues(nm, synth.seed_for(4, obj)), synth.RHO, synth.ALPHA, synth.BETA)
        s.set_use_transfer(False)
        s.enqueue_force(np.full(nm, 1e-3 * (obj + 1)))
        want = np.concatenate([s.step()[0] for _ in range(2)])
        assert np.abs(a[obj] - want).max() <= 5e-4 * np.abs(want).max(), obj
