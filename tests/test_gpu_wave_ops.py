"""Wave-level helpers of the block kernels (openpbso_amd/csrc/wave_ops.h) on the GPU: tests/cpp/wave_sum16_test.hip is built
with hipcc for gfx950 and run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_wave_sum16_butterfly(tmp_path):
    exe = str(tmp_path / "wave_sum16_test")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "openpbso_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "wave_sum16_test.hip"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "wave_sum16: ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
