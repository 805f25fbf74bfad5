#!/bin/bash
# Round 6 evidence, one call:  (a) pytest -m gpu, (b) the default bench line, (c) rocprofv3 kernel-trace stats of the default command,
# (d) PMC passes of the headline launch -- each counter group in a pass of its own, the program directly behind `--`, the start gate OFF
#     (PBSO_ENGINE_OPTS=stream_sync=1: no waiting kernel in the stream while the profiler serialises dispatches), no clock ramp, no
#     side legs -- and ONE contrast pass with the gate as the policy sets it (round 5: four of five passes ended at their time limit).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p6; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p6
WHAT="${1:-tests bench stats pmc}"
if [[ $WHAT == *tests* ]]; then timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt; fi
if [[ $WHAT == *bench* ]]; then
  t0=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; t1=$(date +%s); echo "python bench.py: $((t1 - t0)) s wall" | tee $O/default_command_wall_seconds.txt
  python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags_steps20_warmup5.json 2> /dev/null
fi
if [[ $WHAT == *stats* ]]; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_default -- python3 $R/bench.py --no-cpu-baseline --no-second-form --no-one-second-leg > $O/st_default.log 2>&1)
  f=$(find $O/st_default -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_default.csv; rm -rf $O/st_default; head -6 $O/kernel_stats_default.csv | cut -c1-140
fi
if [[ $WHAT == *pmc* ]]; then
  BARGS="--steps 3 --warmup 1 --settle 0 --clock-ramp-ms 0 --no-cpu-baseline --no-parity --no-second-form --no-strong-share --no-one-second-leg"
  pmc() { name=$1; lim=$2; shift; shift; t0=$(date +%s); (cd /tmp && timeout $lim rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 $R/bench.py $BARGS > $O/pmc_$name.log 2>&1); rc=$?; echo "pmc $name ($*): rc=$rc after $(( $(date +%s) - t0 )) s, PBSO_ENGINE_OPTS=${PBSO_ENGINE_OPTS:-}" | tee -a $O/pmc_passes.txt; }
  rm -f $O/pmc_passes.txt
  export PBSO_ENGINE_OPTS=stream_sync=1
  pmc fetch 300 FETCH_SIZE
  pmc write 300 WRITE_SIZE
  pmc m1 300 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
  pmc m2 300 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
  unset PBSO_ENGINE_OPTS
  pmc write_gated 200 WRITE_SIZE
  python - <<'PY' > $O/pmc_summary_block.txt
import csv, glob, collections
print("per-dispatch averages for pbso kernels (rocprofv3 --kernel-trace --pmc, separate passes; bench.py --steps 3 --warmup 1 --settle 0 --clock-ramp-ms 0 --no-second-form; start gate off unless the pass says gated)")
for name in ("fetch", "write", "m1", "m2", "write_gated"):
    fs = glob.glob(f"gpurun_out/p6/pmc_{name}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(f"{name:12s} no counter file (the pass did not finish)")
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "iir_b" not in r["Kernel_Name"] and "streamOps" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        for c, v in sorted(agg[k].items()):
            print(f"{name:12s} {k:62s} {c:28s} {v / cnt[(k, c)]:.6g}  (n={cnt[(k, c)]})")
PY
  rm -rf $O/pmc_*/; cat $O/pmc_passes.txt; grep -E "FETCH|WRITE|INSTS_MFMA|INSTS_VALU |INSTS_LDS|BANK_CONFLICT" $O/pmc_summary_block.txt
fi
