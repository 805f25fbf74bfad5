#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
timeout 900 python bench.py --steps 20 --warmup 5 "$@" > gpurun_out/r04/bench_full.json 2> gpurun_out/r04/bench_full.err
tail -3 gpurun_out/r04/bench_full.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04/bench_full.json").read().strip().splitlines()[-1])
print("x", round(d["realtime_x"]), "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"])
hd=d.get("host_delivered",{})
print({k:hd.get(k) for k in ("d2h_ms_per_step","to_host_ms_per_step_measured","realtime_x_overlapped_measured")}, hd.get("mix_on_device"))
for r in (d.get("strong_share") or {}).get("shares", []): print(r)
print("cpu", d.get("cpu_baseline",{}).get("value"), "second", (d.get("mixed_precision_projection") or {}).get("realtime_x"))
PY
