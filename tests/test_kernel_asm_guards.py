"""Build-time guards on the oscillator-bank kernel's generated code (no GPU needed:
hipcc cross-compiles).  The kernel writes its LDS tile with ds_write_addtid_b32, whose
address comes from M0: nothing else in the kernel may write M0, there must be no
scratch (spilled registers), and descriptors/profiles must be scalar loads."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openpbso_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_iir_kernel_generated_code(tmp_path):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize",
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    os.path.join(CSRC, "kernels_iir.hip"), "-o", str(tmp_path / "k.s")], check=True, capture_output=True)
    asm = open(tmp_path / "k.s").read()
    kernels = re.split(r"\n(?=_ZN4pbso10iir_scalar15iir_bank_kernel\S*:)", asm)[1:]
    assert len(kernels) >= 20
    headline = [k for k in kernels if k.startswith("_ZN4pbso10iir_scalar15iir_bank_kernelILi2ELi0ELi1ELi256E")]
    assert len(headline) == 1
    for k in kernels:
        body = k.split("s_endpgm")[0]
        name = body.split(":")[0]
        m0_writes = [l for l in body.splitlines() if re.search(r"\bm0\b", l) and not l.strip().startswith(";")]
        assert all("s_mov_b32 m0" in l for l in m0_writes), (name, m0_writes[:3])
        assert "ds_write_addtid_b32" in body, name
    h = headline[0].split("s_endpgm")[0]
    assert "scratch_" not in h                                   # no spills in the headline shape
    assert "s_load_dwordx8" in h                                 # BufDesc via one scalar load
    hot = [l for l in h.splitlines() if re.match(r"\s+v_(fma|fmac|mul|add)_f32", l)]
    assert hot and all("_e64" not in l for l in hot)            # VOP2 only in the arithmetic
    meta = asm[asm.find(".amdgpu_metadata"):]
    m = re.search(r"\.name:\s+_ZN4pbso10iir_scalar15iir_bank_kernelILi2ELi0ELi1ELi256E.*?\.vgpr_count:\s+(\d+)", meta, re.S)
    if m:
        assert int(m.group(1)) <= 128                            # 4 waves per SIMD need <= 128 VGPRs
