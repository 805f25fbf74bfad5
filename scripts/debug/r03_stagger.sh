#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03st
for st in 0 257 513 258 514 260 516 769; do
  PBSO_STAGGER=$st timeout 600 python bench.py --no-cpu-baseline --no-second-form --no-parity --form block --steps 60 --warmup 3 > gpurun_out/r03st/s$st.json 2>/dev/null
  python - gpurun_out/r03st/s$st.json $st <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); st = int(sys.argv[2])
print("stagger mask=%d units=%d  rt=%.1f ms/step=%.4f kernel=%.4f" % (st >> 8, st & 255, d["realtime_x"], d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
done
