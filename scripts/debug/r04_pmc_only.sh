#!/bin/bash
# the PMC passes of scripts/gpu_profiles_r04.sh alone (+ the three headline lines that read their results)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p4; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p4
pmc() { form=$1; name=$2; shift; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_${form}_$name -- python3 $R/bench.py --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-parity --no-second-form --no-strong-share --no-one-second-leg --form $form > $O/pmc_${form}_$name.log 2>&1); echo "pmc $form $name rc=$?"; }
for form in block; do
pmc $form m1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
pmc $form m2 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
pmc $form m3 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_CYCLES
pmc $form fetch FETCH_SIZE
pmc $form write WRITE_SIZE
python - $form <<'PY' > gpurun_out/p4/pmc_summary_$form.txt
import csv, glob, collections, sys
form = sys.argv[1]
print(f"per-dispatch averages for pbso kernels (rocprofv3 --kernel-trace --pmc, separate passes; bench.py --steps 3 --warmup 1 --settle 0 --no-second-form --form {form})")
for name in ("m1", "m2", "m3", "fetch", "write"):
    fs = glob.glob(f"gpurun_out/p4/pmc_{form}_{name}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-56:]
        if "pbso" not in r["Kernel_Name"]: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        for c, v in sorted(agg[k].items()):
            print(f"{name:6s} {k:58s} {c:28s} {v / cnt[(k, c)]:.6g}  (n={cnt[(k, c)]})")
PY
rm -rf $O/pmc_${form}_*/ ; grep -E "iir_block.*(INSTS_MFMA|INSTS_VALU |COEXEC|FETCH|WRITE)" $O/pmc_summary_$form.txt
done
