"""The bench.py output contract (driver side): the committed headline line carries every required
key with the right type, and the defaults of bench.py are the single-GPU headline configuration."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("rnd", ["r01", "r02"])
def test_committed_headline_line_has_the_contract_keys(rnd):
    d = json.load(open(os.path.join(ROOT, "profiles", rnd + "_bench_default.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k, t in dict(metric=str, value=float, unit=str, n_gpus=int, steps=int, warmup=int, ms_per_step=float,
                     higher_is_better=bool, scaling=str, dtype=str, data=str, config=dict, roofline=dict,
                     cpu_baseline=dict).items():
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True and d["n_gpus"] == 1
    assert d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["metric"].split()[0] in base["metric"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["unit"] == "TFLOP/s" and r["hbm"]["unit"] == "GB/s" and r["traffic"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"]
    # value is whole-job throughput: objects x buffers x 513 samples x steps / time
    cfg = d["config"]
    per_step = cfg["objects_per_gpu"] * cfg["buffers_per_step"] * cfg["frames_per_buffer"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    if rnd != "r01":
        # round 2 on: the run checks its own first timed step against the oracle, and says how the kernel was timed
        assert d["parity"]["pass"] and d["parity_checked_objects"] == 8 and d["max_err"] <= d["parity"]["tol_max"]
        assert "kernel_ms_source" in r and r["bound"] in ("valu", "mfma", "hbm")


def test_gpus_n_without_a_launcher_starts_the_ranks_before_touching_the_gpu(monkeypatch):
    """python bench.py --gpus 2 with no WORLD_SIZE: the parent runs torch.distributed.run as a CHILD (no exec,
    no torch import, hence no HIP initialisation in the parent) and exits with its code."""
    sys.path.insert(0, ROOT)
    import bench
    calls = []

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return Done()

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    torch_loaded_before = "torch" in sys.modules
    try:
        bench.main()
        assert False, "main() must exit with the child's code"
    except SystemExit as e:
        assert e.code == 7
    assert len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert env["PBSO_BENCH_SPAWNED"] == "1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert ("torch" in sys.modules) == torch_loaded_before        # the spawn path imports nothing that could touch the GPU
    a = bench.parse(["--gpus", "2"])
    assert not a.no_gather                                        # the RCCL gather is part of the N > 1 line by default


def test_bench_defaults_are_the_single_gpu_headline(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.objects, a.modes, a.buffers) == (1, 1024, 512, 86)
    assert a.steps > 0 and a.warmup >= 0 and a.qnorm == "sample" and a.form == "block"
