#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== microbench2 =="; timeout 600 ./openpbso_amd/microbench2_gfx950 > gpurun_out/microbench2.txt 2>&1; echo "rc=$?"; cat gpurun_out/microbench2.txt
echo "== pytest gpu (failed ones) =="; timeout 900 python -m pytest tests -m gpu -q -k "config2 or config5_shape" > gpurun_out/pytest_gpu2.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/pytest_gpu2.txt
echo "== counters list =="; (cd /tmp && timeout 120 rocprofv3 -L > "$OLDPWD/gpurun_out/counters_list.txt" 2>&1); grep -c "" gpurun_out/counters_list.txt; grep -oE "\bSQ_[A-Z_0-9]+" gpurun_out/counters_list.txt | sort -u | tr '\n' ' ' | head -c 6000; echo
run_pmc() {
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OLDPWD/gpurun_out/pmc_$name" -- python3 "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmc_$name.log" 2>&1); echo "pmc $name rc=$?"
  f=$(find gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "iir_bank" in k:
        for c, v in agg[k].items():
            print(f"  {c} = {v / max(cnt[(k, c)],1):.4g} per dispatch ({cnt[(k,c)]} dispatches)")
PY
  fi
}
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run_pmc sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA
run_pmc sq3 SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 GRBM_GUI_ACTIVE
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
