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
