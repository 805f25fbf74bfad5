#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke =="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/smoke.txt
echo "== pytest gpu =="; timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|C2 R|C1 |C5|direct form|Error|error" gpurun_out/pytest_gpu.txt | tail -60
for cfg in "1 1 0" "1 0 0" "0 1 0" "0 0 0" "1 1 4" "1 1 1"; do
  set -- $cfg
  echo "== bench packed=$1 addtid=$2 mpl=$3 =="
  PBSO_IIR_PACKED=$1 PBSO_LDS_ADDTID=$2 timeout 600 python bench.py --steps 4 --warmup 2 --modes-per-lane $3 --no-cpu-baseline > gpurun_out/bench_p$1_a$2_r$3.json 2> gpurun_out/bench_p$1_a$2_r$3.err; echo "rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_p$1_a$2_r$3.json"))
    print("value=%.4g rt=%.1f ms/step=%.2f kernel_ms=%.3f valu_frac=%.3f plan_ms=%.2f dev_ms=%.2f R=%d W=%d"%(d["value"],d["realtime_x"],d["ms_per_step"],d["roofline"]["kernel_ms"],d["roofline"]["frac"],d["timing"]["host_plan_ms"],d["timing"]["device_pipeline_ms"],d["config"]["modes_per_lane"],d["config"]["waves_per_object"]))
except Exception as e:
    print("bad json", e)
PY
  tail -2 gpurun_out/bench_p$1_a$2_r$3.err
done
echo "== bench no-qnorm / direct =="
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-qnorm > gpurun_out/bench_noqn.json 2>/dev/null; cat gpurun_out/bench_noqn.json | python -c "import json,sys; d=json.load(sys.stdin); print('noqn', d['realtime_x'], d['roofline']['kernel_ms'])"
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --form direct > gpurun_out/bench_direct.json 2>/dev/null; cat gpurun_out/bench_direct.json | python -c "import json,sys; d=json.load(sys.stdin); print('direct', d['realtime_x'], d['roofline']['kernel_ms'])"
echo "== rocprofv3 kernel trace =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_r01" -- python3 "$OLDPWD/bench.py" --steps 4 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_r01.log" 2>&1; echo "rc=$?"
cd "$OLDPWD"; for f in $(find gpurun_out/prof_r01 -name "*kernel_stats.csv" | head -1); do head -6 $f | cut -c1-200; done
