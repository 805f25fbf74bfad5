"""SURVEY section 5 (sanitizers), CPU side only: the fp64 oracle and the product's GPU-free file readers
(csrc/loaders.cpp) are rebuilt with -fsanitize=address,undefined and run over the golden fixtures, over every
truncation and over three corruptions of every byte of the .fatcube files, over truncated .modes files,
malformed materials and .obj meshes.  A sanitizer report (overread, leak, signed overflow, misaligned
access ...) aborts the checker with a non-zero status.  Never the GPU build: gpurun has no GPU sanitizers."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _fixtures(tmp_path):
    files = sorted(glob.glob(os.path.join(GOLD, "*.fatcube")) + glob.glob(os.path.join(GOLD, "*.modes")) +
                   glob.glob(os.path.join(GOLD, "material_*.txt")))
    assert len(files) >= 8
    # malformed inputs of tests/test_loaders_golden.py, as files
    for name, data in (("bad_varint.fatcube", bytes([0x0A, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x01])),
                       ("len_past_end.fatcube", bytes([0x0A, 0x7F, 0x09, 0x00])), ("empty.fatcube", b""),
                       ("garbage.modes", b"\xff" * 37), ("neg.modes", (-5).to_bytes(4, "little", signed=True) * 2),
                       ("binary.txt", bytes(range(256))), ("empty.txt", b"")):
        (tmp_path / name).write_bytes(data)
        files.append(str(tmp_path / name))
    (tmp_path / "mesh.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1/1 -1/1 2/1\nf 4 3 2 1\n")
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0\nf 1 2 999999999999999999999\nf\n" + "f " + "1 " * 5000 + "\n")
    return files + [str(tmp_path / "mesh.obj"), str(tmp_path / "bad.obj")]


def _run(cmd):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)


def _check(r, what):
    assert r.returncode == 0 and what in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    rejected = {os.path.basename(l.split(": ", 1)[1]) for l in r.stdout.splitlines() if l.startswith("rejected: ")}
    # every committed fixture parses; every malformed file is turned down -- both without a sanitizer report
    assert not [f for f in rejected if not f.startswith(("bad", "len_past_end", "garbage", "neg"))], rejected
    return rejected


def test_oracle_under_asan_and_ubsan(tmp_path):
    b = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    files = [f for f in _fixtures(tmp_path) if not f.endswith(".obj")]            # the oracle has no mesh reader
    rejected = _check(_run([os.path.join(ROOT, "oracle", "oracle_asan_check")] + files), "oracle asan/ubsan check ok")
    assert {"bad_varint.fatcube", "len_past_end.fatcube", "garbage.modes", "neg.modes"} <= rejected


def test_product_loaders_under_asan_and_ubsan(tmp_path):
    b = subprocess.run(["make", "-C", os.path.join(ROOT, "openpbso_amd", "csrc"), "loaders_asan"], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    rejected = _check(_run([os.path.join(ROOT, "openpbso_amd", "loaders_asan_check")] + _fixtures(tmp_path)),
                      "loaders asan/ubsan check ok")
    assert {"bad_varint.fatcube", "len_past_end.fatcube", "garbage.modes", "neg.modes", "bad.obj"} <= rejected
