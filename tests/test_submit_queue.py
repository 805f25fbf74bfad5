"""The engine's second submitting thread (pbso_engine_desc::submit_thread, openpbso_amd/csrc/submit_queue.h).

Here, without a GPU: the queue itself under ThreadSanitizer (tests/cpp/submit_queue_tsan.cpp -- order of the calls, arguments
captured by value, wait / drain, a failing call, the destructor).  On the GPU (tests/test_gpu_submit_thread.py): steps with the
thread are bit-identical to steps without it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_submit_queue_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "sq_tsan")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "openpbso_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "submit_queue_tsan.cpp"), "-o", exe, "-lpthread"], check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "submit queue ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ThreadSanitizer" not in r.stderr


def test_desc_field_and_flush_are_bound():
    from openpbso_amd import capi, solver
    assert "submit_thread" in dict(capi.EngineDesc._fields_) and "submit_thread" in solver.SELECT_FIELDS
    assert "pbso_flush" in capi.EXPORTS
