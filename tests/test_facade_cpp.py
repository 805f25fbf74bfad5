"""The C++ facade (include/openpbso_amd_facade.h) used the way the reference's
tool uses modal_solver.h: tests/cpp/facade_smoke.cpp is built with g++ against
the C-ABI library; on the GPU its audio is compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_smoke.cpp")


def _build(tmp_path):
    from openpbso_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    exe = str(tmp_path / "facade_smoke")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
                    "-L" + libdir, "-lopenpbso_amd", "-Wl,-rpath," + libdir, "-lpthread"], check=True)
    return exe


def test_facade_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


def test_facade_sound_queue_is_a_lock_free_spsc_ring(tmp_path):
    """the queue the PortAudio callback reads (modal_solver.h:105-109): 3 usable slots, FIFO, and a
    200 000-message producer / consumer run on two threads (no GPU involved)"""
    r = subprocess.run([_build(tmp_path), "--ring-selftest"], capture_output=True, text=True)
    assert r.returncode == 0 and "ring selftest ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
    src = open(os.path.join(ROOT, "include", "openpbso_amd_facade.h")).read()
    body = src[src.index("bool dequeueSoundMessage"):]
    body = body[:body.index("\n")]
    assert "mutex" not in body and "lock" not in body              # the audio thread takes no lock


@pytest.mark.gpu
def test_facade_audio_matches_oracle(tmp_path):
    from oracle import oracle_py as orc
    exe = _build(tmp_path)
    out = str(tmp_path / "out.f32")
    r = subprocess.run([exe, out], check=True, capture_output=True, text=True)
    assert "transfer[0]=1e+07" in r.stdout            # setUseTransfer(false) -> unit transfer
    assert "clear: enqueued=1 sound_after_clear=0" in r.stdout       # "Clear force": accepted, that step emits no buffer
    assert "missing_ffat_dir: out_of_range=1" in r.stdout            # empty map -> at(0) throws, as in the reference
    # enqueueArprmMessageNoFail (modal_solver.h:382-393): bounded tries on a full slot fail, the unbounded call of a second thread
    # gets in once a step() has taken the message before it
    assert "arprm: started=1 first=1 bounded=0 second=1 after_a_step=1" in r.stdout, r.stdout
    got = np.fromfile(out, dtype=np.float32).astype(np.float64)
    n_modes, n_verts, nb = 96, 8, 6
    raw = np.fromfile(out + ".model", dtype=np.float64)
    lam, shapes = raw[:n_modes], raw[n_modes:].reshape(n_modes, 3 * n_verts)
    vn = np.array([0.6, 0.0, 0.8])
    s = orc.Solver(lam, 2500.0, 6.0, 1e-7)
    s.set_use_transfer(False)
    s.enqueue_force(orc.modal_force_vertex(shapes, 3, vn))
    want = []
    for b in range(nb):
        if b == 2:
            s.enqueue_force(orc.modal_force_vertex(shapes, 5, vn), orc.make_force(orc.GAUSSIAN, 300.0))
        want.append(s.step()[0])
    assert s.enqueue_force(np.zeros(0), None, False, False, True)     # clearAllForces with empty data
    assert s.step() is None                                           # no buffer for that step
    want.append(s.step()[0])
    want = np.concatenate(want) / 1e10                 # PaModalCallback scaling
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 5e-4 * np.abs(want).max()
