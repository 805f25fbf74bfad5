"""The bench.py output contract (driver side): the committed headline line carries every required
key with the right type, and the defaults of bench.py are the single-GPU headline configuration."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("rnd", ["r01", "r02", "r03", "r04"])
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
    if rnd in ("r03", "r04"):
        # round 3 on: `value` is the all-f32 block form; the roofline spells its minimum work out; HBM fraction at top level; the
        # host-delivered rate is measured; the mixed-precision leg is labelled and carries its own roofline and parity
        assert 0 < d["hbm_frac"] < 1 and abs(d["hbm_frac"] - r["hbm"]["frac"]) < 1e-12
        mw = r["min_work"]
        assert set(mw["flop_per_mode_sample_by_pipe"]) == {"f32_matrix_pipe", "f32_vector_alu"} and mw["min_kernel_ms"] < r["kernel_ms"]
        assert abs(r["frac"] - mw["min_kernel_ms"] / r["kernel_ms"]) < 1e-6
        hd = d["host_delivered"]
        assert hd["d2h_ms_per_step"] > 0 and hd["bytes_per_step"] == cfg["objects_per_gpu"] * cfg["buffers_per_step"] * cfg["frames_per_buffer"] * 4
        mp = d["mixed_precision_projection"]
        assert mp["dtype"] != "f32" and mp["parity_pass"] and mp["roofline"]["frac"] > 0 and mp["realtime_x"] > d["realtime_x"]
    if rnd == "r04":
        # round 4: the strong-scaling proxy (the per-rank shares of the configuration on 2 / 4 / 8 GPUs, each oracle-checked), the
        # job's object count, delivery to the host measured through the product's own path, the on-device object mix
        assert cfg["objects_total"] == cfg["objects_per_gpu"] == 1024
        shares = d["strong_share"]["shares"]
        assert [(x["n_gpus"], x["objects"]) for x in shares] == [(2, 512), (4, 256), (8, 128)]
        for x in shares:
            assert x["parity_pass"] and x["max_err"] <= 5e-4 and x["ms_per_step"] < d["ms_per_step"]
            assert abs(x["implied_efficiency_at_N"] - d["ms_per_step"] / x["n_gpus"] / x["ms_per_step"]) < 1e-9
            assert x["time_chunked_launches"] == x["bank_launches"] > 0          # the shares run cut along the time axis (K5)
        assert cfg["time_chunked_launches"] == 0                                 # ... the full chip does not
        assert 0.5 < hd["mix_on_device"]["frac_of_value"] <= 1.0
        # the line of the round's final form steps ten seconds of audio per call and carries the one-second-step rate beside it
        if cfg["buffers_per_step"] > 86:
            one = d["steps_of_one_second"]
            assert one["buffers_per_step"] == 86 and 0.9 * d["value"] < one["value"] < d["value"]
        # delivery to the host through the product's own path is measured by `bench.py --host-delivery` (its launches are the
        # headline kernel at the PCIe link's pace: kept out of the default command so that a profiler's per-kernel average over it
        # stays the headline's)
        assert hd["to_host_ms_per_step_measured"] is None and "--host-delivery" in hd["to_host_measured_by"]
        h = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_host_delivery.json")))["host_delivered"]
        assert h["realtime_x_overlapped_measured"] > h["realtime_x_if_copied_after_each_step"]
        assert h["to_host_ms_per_step_measured"] >= 0.95 * h["d2h_ms_per_step"]          # PCIe-bound: no faster than the bare copy


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
    assert (a.gpus, a.objects, a.modes, a.buffers) == (1, 1024, 512, 860)      # 10 s of audio per step: SURVEY 8(d)'s throughput duration
    assert a.settle * a.buffers >= 3000                                        # (the clock-ramp steps follow the step size)
    assert a.steps > 0 and a.warmup >= 0 and a.qnorm == "sample" and a.form == "block"
    # settle steps follow the DEVICE time of a step (round 5: short legs were timed inside the shader clock's ramp): at least 40 s of
    # audio, at least ~0.1 s of device time by the headline's rate, capped by what the oracle check of the first timed step costs
    assert a.settle_auto and a.settle == bench.auto_settle(1024, 512, 860) == 11 and a.clock_ramp_ms > 0
    assert bench.auto_settle(128, 512, 860) == 75 and bench.auto_settle(128, 512, 86) == 750
    assert bench.auto_settle(8, 4096, 86) == 93 and bench.auto_settle(1024, 512, 86) >= 100
    for o, m, nb in ((1, 512, 86), (64, 256, 86), (8, 4096, 860), (4096, 64, 1)):
        assert bench.auto_settle(o, m, nb) * nb >= 3000                        # never fewer than the 40 s of audio of rounds 1 - 4
    assert bench.parse(["--settle", "3"]).settle == 3 and not bench.parse(["--settle", "3"]).settle_auto
