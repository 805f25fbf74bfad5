#!/bin/bash
# Collects the round's evidence under gpurun_out/final/ (copied to profiles/ afterwards by
# scripts/collect_profiles.py): rocprofv3 kernel-trace stats of the default bench command, PMC passes
# (own runs, no trace domains besides kernel-trace), the workgroup census, and bench JSON lines.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/final
export TMPDIR=/tmp
R=$PWD
echo "== env =="; (rocminfo | grep -E "Marketing Name|gfx9" | sort | uniq -c | head -4; lscpu | grep -E "Model name|^CPU\(s\)"; echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; free -g | head -2) > gpurun_out/final/env.txt 2>&1; cat gpurun_out/final/env.txt
echo "== bench default (driver flags, with cpu baseline and parity) =="; timeout 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo rc=$?; cut -c1-400 gpurun_out/final/bench_default.json
timeout 600 python bench.py --no-cpu-baseline --qnorm off > gpurun_out/final/bench_qnorm_off.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --form block > gpurun_out/final/bench_block_f32.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --form velocity > gpurun_out/final/bench_velocity.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --form velocity --qnorm closed > gpurun_out/final/bench_velocity_qnorm_closed.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --form direct > gpurun_out/final/bench_direct.json 2>/dev/null
echo "== rocprofv3 kernel trace + stats of the default command =="
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/rocprof_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/rocprof_stats.log 2>&1); echo rc=$?
f=$(find gpurun_out/final/rocprof_stats -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-260
python scripts/trace_gaps.py gpurun_out/final/rocprof_stats > gpurun_out/final/trace_gaps.txt 2>&1; tail -8 gpurun_out/final/trace_gaps.txt
pmc() { form=$1; name=$2; shift; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/final/pmc_${form}_$name -- python3 $R/bench.py --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-parity --form $form > $R/gpurun_out/final/pmc_${form}_$name.log 2>&1); echo "pmc $form $name rc=$?"; }
for form in block_bf16 block; do
pmc $form m1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
pmc $form m2 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
pmc $form m3 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_CYCLES
pmc $form m4 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD SQ_INSTS_SMEM SQ_IFETCH
pmc $form fetch FETCH_SIZE
pmc $form write WRITE_SIZE
python - $form <<'PY' > gpurun_out/final/pmc_summary_$form.txt
import csv, glob, collections, sys
form = sys.argv[1]
print(f"per-dispatch averages for pbso kernels (rocprofv3 --pmc, bench.py --steps 3 --warmup 1 --settle 0 --form {form})")
for name in ("m1", "m2", "m3", "m4", "fetch", "write"):
    fs = glob.glob(f"gpurun_out/final/pmc_{form}_{name}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-48:]
        if "pbso" not in r["Kernel_Name"]: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        for c, v in sorted(agg[k].items()):
            print(f"{name:6s} {k:50s} {c:28s} {v / cnt[(k, c)]:.6g}  (n={cnt[(k, c)]})")
PY
grep -E "iir_block.*(MFMA|INSTS_VALU |ACTIVE_INST_VALU|WAIT|FETCH|WRITE|GRBM)" gpurun_out/final/pmc_summary_$form.txt
done
cp gpurun_out/final/pmc_summary_block_bf16.txt gpurun_out/final/pmc_summary.txt
echo "== census =="; PBSO_CENSUS=1 timeout 300 python scripts/census.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/final/census.txt; tail -8 gpurun_out/final/census.txt
PBSO_FORM=block PBSO_CENSUS=1 timeout 300 python scripts/census.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/final/census_block_f32.txt
