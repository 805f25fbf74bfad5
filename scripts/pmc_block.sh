#!/bin/bash
# PMC passes for the block kernel (own runs, kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2; export TMPDIR=/tmp; R=$PWD
ARGS="${PMC_ARGS:-}"
pmc() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/r2/pmc_$name -- python3 $R/bench.py --steps 3 --warmup 1 --settle 0 --no-cpu-baseline $ARGS > $R/gpurun_out/r2/pmc_$name.log 2>&1); echo "pmc $name rc=$?"; }
pmc m1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
pmc m2 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
pmc m3 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_CYCLES
python - <<'PY' > gpurun_out/r2/pmc_block_summary.txt
import csv, glob, collections
print("per-dispatch averages for pbso kernels (rocprofv3 --pmc, bench.py --steps 3 --warmup 1 --settle 0)")
for name in ("m1", "m2", "m3"):
    fs = glob.glob(f"gpurun_out/r2/pmc_{name}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-48:]
        if "iir_b" not in r["Kernel_Name"]: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        for c, v in sorted(agg[k].items()):
            print(f"{name:6s} {k:50s} {c:28s} {v / cnt[(k, c)]:.6g}  (n={cnt[(k, c)]})")
PY
cat gpurun_out/r2/pmc_block_summary.txt
