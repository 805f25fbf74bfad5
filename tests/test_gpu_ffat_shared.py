"""K4 for objects whose modes share one FFAT map geometry (round 6; kernels_exact.hip, ffat_lookup_shared_kernel).

Engine::finalize compares the map headers of an object's modes field by field; when they agree the object's maps are also kept
transposed ([cell][mode]) and its listener events go to a kernel that locates the position once per event -- lane = mode, the four
corner reads and the write coalesced.  Same expressions in the same order (ffat_locate + ffat_combine): the transfer rows are
bit-identical to the per-(event, mode) kernels (PBSO_FFAT_SHARED=0) and to the oracle."""
import numpy as np
import pytest

from openpbso_amd import ForceMessage, synth
from openpbso_amd.solver import Engine
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def _scene():
    sizes = [96, 64, 300, 17, 1]
    lams = [synth.eigenvalues(n, 700 + i) for i, n in enumerate(sizes)]
    maps = []
    maps.append(synth.ffat_maps(lams[0], 700, dim=4, cell_size=0.01))                       # 0: one geometry, every mode
    maps.append(synth.ffat_maps(lams[1], 701, dim=4, cell_size=0.012))                      # 1: another cell size
    maps.append(synth.ffat_maps(lams[2], 702, dim=8, cell_size=0.006, center=(0.01, -0.02, 0.005)))      # 2: a finer cube, off-centre
    m3 = synth.ffat_maps(lams[3], 703, dim=4, cell_size=0.01)                               # 3: mode 3 over another box: per-mode kernels
    other = synth.uniform_cube_geometry((0.004, 0.0, -0.003), 0.011, 4)
    for key, val in other.items():
        m3[3][key] = val
    maps.append(m3)
    maps.append(synth.ffat_maps(lams[4], 704, dim=4, cell_size=0.01))                       # 4: one mode, a single map: nothing to share
    return sizes, lams, maps


def _oracle_row(maps, n_modes, p):
    row = np.zeros(n_modes)
    for m in maps:
        dim = int(m["n_elements"][0][0])
        om = orc.uniform_cube(m["mode_id"], m["k"], m["center"], m["cell_size"], dim, m["psi"])
        row[m["mode_id"]] = abs(orc.ffat_get_map_val(om, p))
    return row


def _run(monkeypatch, shared, steps, pos, **select):
    monkeypatch.setenv("PBSO_FFAT_SHARED", "1" if shared else "0")
    sizes, lams, maps = _scene()
    total = sum(steps)
    with Engine(**select) as eng:
        for i, n in enumerate(sizes):
            eng.add_object(lams[i], synth.RHO, synth.ALPHA, synth.BETA)
            eng.set_ffat_maps(i, maps[i])
        eng.finalize()
        rng = np.random.default_rng(5)
        for i, n in enumerate(sizes):
            assert eng.enqueue_force(i, ForceMessage(data=rng.standard_normal(n) * 1e-3), 0)
            assert eng.compute_transfer_path(np.full(total, i, dtype=np.int32), pos[i], np.arange(total, dtype=np.int64)).all()
        audio, latest = [], []
        for nb in steps:
            eng.step(nb)
            audio.append(eng.audio().copy())
            latest.append([eng.latest_transfer(i).copy() for i in range(len(sizes))])
        return audio, latest, eng.info()


@pytest.mark.parametrize("select", [{}, dict(time_chunks=-1), dict(bank_kernel=2)])
def test_shared_geometry_lookup_is_bit_identical_to_the_per_mode_kernels_and_the_oracle(monkeypatch, select):
    sizes, lams, maps = _scene()
    steps = [1, 9, 1, 4]
    total = sum(steps)
    rng = np.random.default_rng(11)
    pos = []
    for i in range(len(sizes)):
        d = rng.standard_normal((total, 3))
        p = (0.3 + 0.4 * rng.random((total, 1))) * d / np.linalg.norm(d, axis=1, keepdims=True)
        # rays through face edges and corners, and nearly along an axis (clamped bilinear weights)
        p[2] = [0.3, 0.3, 0.3]
        p[5] = [0.5, -0.5, 0.5000001]
        p[7] = [2.0, 1e-9, 1e-9]
        pos.append(p)
    a_audio, a_latest, a_info = _run(monkeypatch, True, steps, pos, **select)
    b_audio, b_latest, b_info = _run(monkeypatch, False, steps, pos, **select)
    # objects 0, 1, 2 through the new kernel, 3 and 4 through the old ones -- in the same launches
    assert a_info["total_ffat_shared_events"] == 3 * total and a_info["total_ffat_general_events"] == 2 * total
    assert b_info["total_ffat_shared_events"] == 0 and b_info["total_ffat_general_events"] == 5 * total
    done = 0
    for k, nb in enumerate(steps):
        done += nb
        assert np.array_equal(a_audio[k], b_audio[k]), k
        for i in range(len(sizes)):
            assert np.array_equal(a_latest[k][i], b_latest[k][i]), (k, i)
            want = _oracle_row(maps[i], sizes[i], pos[i][done - 1])
            got = a_latest[k][i][:sizes[i]]
            assert np.array_equal(got, want), (k, i)               # bit for bit
        assert np.abs(a_audio[k]).max() > 0


def test_many_events_of_one_object_in_one_launch(monkeypatch):
    """a whole listener path in one launch (86 positions per object, the headline listener scene's shape) and a single event"""
    steps = [86, 1]
    total = sum(steps)
    sizes, _, maps = _scene()
    pos = [synth.listener_path(total, radius=0.4 + 0.05 * i) for i in range(len(sizes))]
    a_audio, a_latest, a_info = _run(monkeypatch, True, steps, pos)
    b_audio, b_latest, _ = _run(monkeypatch, False, steps, pos)
    assert a_info["total_ffat_shared_events"] == 3 * total
    for k in range(len(steps)):
        assert np.array_equal(a_audio[k], b_audio[k])
        for i in range(len(sizes)):
            assert np.array_equal(a_latest[k][i], b_latest[k][i])
