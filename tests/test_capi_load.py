"""CPU-side checks of the product boundary: the C-ABI library loads and exports
every symbol include/openpbso_amd.h declares (no compute without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from openpbso_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return capi


def test_header_symbols_are_exported(capi):
    hdr = open(os.path.join(ROOT, "include", "openpbso_amd.h")).read()
    declared = set(re.findall(r"\b(pbso_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = capi.lib()
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version_and_status_strings(capi):
    lib = capi.lib()
    assert lib.pbso_abi_version() == capi.ABI_VERSION
    assert lib.pbso_status_string(0) == b"ok"
    assert b"HIP" in lib.pbso_status_string(capi.ERR_HIP)


def test_no_gpu_means_loud_failure(capi):
    """Without a HIP device engine creation must fail (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from openpbso_amd.solver import Engine, PbsoError
    with pytest.raises(PbsoError) as ei:
        Engine()
    assert ei.value.status == capi.ERR_HIP


def test_product_does_not_touch_oracle():
    """The product tree must not import, link or read anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "openpbso_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle/|pbso_oracle|oracle_py|from oracle|import oracle", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_pa_convert_matches_the_portaudio_callback(capi):
    """A9, PaModalCallback (tools/real_time_modal_sound.cpp:207-210): every frame is written twice
    (interleaved stereo) as (float)(sound / 1E10).  Host-only entry point: no GPU needed."""
    import ctypes as C

    import numpy as np
    lib = capi.lib()
    rng = np.random.default_rng(9)
    sound = np.concatenate([rng.standard_normal(513) * 1e10, [0.0, -0.0, 1e10, -3.5e9, 1e-30, 3.4e38, -3.4e38]]).astype(np.float32)
    out = np.full(2 * sound.size + 2, 7.0, dtype=np.float32)           # two guard values behind the frames
    lib.pbso_pa_convert(sound.ctypes.data_as(C.POINTER(C.c_float)), sound.size, out.ctypes.data_as(C.POINTER(C.c_float)))
    want = (sound.astype(np.float64) / 1E10).astype(np.float32)
    assert np.array_equal(out[0:-2:2], want) and np.array_equal(out[1:-2:2], want)      # L == R, bit for bit
    assert np.array_equal(np.signbit(out[0:-2:2]), np.signbit(want))                   # -0.0 stays -0.0
    assert out[-2] == 7.0 and out[-1] == 7.0                                           # exactly 2 * frames written
    lib.pbso_pa_convert(sound.ctypes.data_as(C.POINTER(C.c_float)), 0, out.ctypes.data_as(C.POINTER(C.c_float)))   # zero frames: no write
    assert np.array_equal(out[0:-2:2], want)
