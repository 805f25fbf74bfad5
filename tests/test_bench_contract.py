"""The bench.py output contract (driver side): the committed headline line carries every required
key with the right type, and the defaults of bench.py are the single-GPU headline configuration."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_headline_line_has_the_contract_keys():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_default.json")))
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


def test_bench_defaults_are_the_single_gpu_headline(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.objects, a.modes, a.buffers) == (1, 1024, 512, 86)
    assert a.steps > 0 and a.warmup >= 0 and a.qnorm == "sample" and a.form == "block"
