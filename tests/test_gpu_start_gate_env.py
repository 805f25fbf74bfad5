"""The start gate of long launches (pbso_engine_desc::stream_sync) is a kernel that waits on the device for another kernel's start
(hipStreamWaitValue64 runs as `__amd_rocclr_streamOpsWait` on this stack).  Round 6: every rocprofv3 --pmc pass of a gated launch
hung until its time limit -- counter collection serialises dispatches, the waiting kernel goes first and never ends
(profiles/r06_pmc_passes.txt).  The policy now looks for an environment that serialises kernels and orders its launches by
events only there.  These tests run a gated workload (steps of >= 256 buffers, time-chunked: the preparation runs ahead) in a
child process under the serialising settings a box allows without a profiler, with a time limit: it must complete, and say
which gate it chose."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r)
from openpbso_amd import Engine, ForceMessage, capi, synth
n_obj, M, nb = 128, 256, int(os.environ.get("PBSO_TEST_NB", "300"))
rng = np.random.default_rng(1)
with Engine(qnorm=capi.QNORM_OFF, form=capi.FORM_BLOCK, chunk_buffers=nb) as eng:
    for i in range(n_obj):
        eng.add_object(synth.eigenvalues(M, 100 + i), synth.RHO, synth.ALPHA, synth.BETA)
    eng.finalize()
    for i in range(n_obj):
        eng.set_use_transfer(i, False)
        eng.enqueue_force(i, ForceMessage(data=rng.standard_normal(M) * 1e-3))
    for _ in range(int(os.environ.get("PBSO_TEST_STEPS", "4"))):
        eng.step(nb)
    a = eng.audio()
    info = eng.info()
print(json.dumps(dict(start_gate=info["start_gate"], tc=info["total_time_chunk_launches"], finite=bool(np.isfinite(a).all()),
                      peak=float(np.abs(a).max()), checksum=float(np.abs(a).sum()))))
""" % ROOT


def _run(extra_env, limit=180):
    env = dict(os.environ)
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=limit)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_gated_steps_complete_under_serialising_settings():
    base = _run({})
    assert base["finite"] and base["peak"] > 0 and base["tc"] == 4
    assert base["start_gate"] in (0, 1)                       # 1 where the device has hipStreamWaitValue64
    # HIP waits for every kernel before it launches the next: a waiting kernel would block the launch that releases it
    ser = _run({"AMD_SERIALIZE_KERNEL": "3"})
    assert ser["start_gate"] == -1 and ser["finite"]
    # the gate is ordering only: the same samples
    assert ser["checksum"] == base["checksum"] and ser["peak"] == base["peak"]
    # a tool the engine's list does not know, declared by hand
    off = _run({"PBSO_START_GATE": "0"})
    assert off["start_gate"] == -1 and off["checksum"] == base["checksum"]
    # one hardware queue for all streams: packets in submission order, the bank that releases a wait is always in front of it
    one = _run({"GPU_MAX_HW_QUEUES": "1"})
    assert one["finite"] and one["checksum"] == base["checksum"]


def test_short_steps_handed_over_by_value_complete_under_serialising_settings():
    """round 6: launches of fewer than 256 buffers hand over to the bank's stream through a value in signal memory by policy -- again a
    kernel that waits on the device, so the same rule: events only where kernels are serialised; the same samples either way"""
    short = {"PBSO_TEST_NB": "40", "PBSO_TEST_STEPS": "12"}
    base = _run(short)
    assert base["finite"] and base["peak"] > 0 and base["tc"] == 12
    for extra in ({"AMD_SERIALIZE_KERNEL": "3"}, {"PBSO_START_GATE": "0"}, {"GPU_MAX_HW_QUEUES": "1"}, {"HIP_LAUNCH_BLOCKING": "1"}):
        got = _run(dict(short, **extra))
        assert got["finite"] and got["checksum"] == base["checksum"] and got["peak"] == base["peak"], extra
        if "GPU_MAX_HW_QUEUES" not in extra:
            assert got["start_gate"] == -1, extra
