#!/bin/bash
# quick iteration pass: parity tests + bench variants (+ optional PMC)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu =="; timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/pytest_gpu.txt
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("value=%.4g rt=%.1f ms/step=%.2f kernel_ms=%.3f valu_frac=%.3f plan_ms=%.2f dev_ms=%.2f R=%d W=%d" % (
        d["value"], d["realtime_x"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
        d["timing"]["host_plan_ms"], d["timing"]["device_pipeline_ms"], d["config"]["modes_per_lane"], d["config"]["waves_per_object"]))
except Exception as e:
    print("bad json", e)
PY
}
for cfg in "1 1 0 " "0 1 0 " "1 0 0 " "1 1 4 " "1 1 0 --no-qnorm" "1 1 0 --form=direct"; do
  set -- $cfg
  tag="p$1_a$2_r$3$(echo ${4:-} | tr -d ' =-')"
  echo "== bench packed=$1 addtid=$2 mpl=$3 ${4:-} =="
  PBSO_IIR_PACKED=$1 PBSO_LDS_ADDTID=$2 timeout 600 python bench.py --steps 4 --warmup 2 --modes-per-lane $3 --no-cpu-baseline ${4:-} > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "rc=$?"
  summ gpurun_out/bench_$tag.json; grep -v amdgpu.ids gpurun_out/bench_$tag.err | tail -2
done
if [ "${PMC:-0}" = "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OLDPWD/gpurun_out/pmc_sq1" -- python3 "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmc_sq1.log" 2>&1)
  f=$(find gpurun_out/pmc_sq1 -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "iir_bank" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for c, v in agg.items(): print(f"  {c} = {v / cnt[c]:.4g}")
PY
fi
