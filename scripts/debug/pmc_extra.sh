cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/final; export TMPDIR=/tmp; R=$PWD
pmc() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/final/pmc_$name -- python3 $R/bench.py --steps 3 --warmup 1 --settle 0 --no-cpu-baseline > $R/gpurun_out/final/pmc_$name.log 2>&1); echo "pmc $name rc=$?"; }
pmc sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32
pmc sq4 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD SQ_ACTIVE_INST_VALU2 SQ_IFETCH
python - <<'PY'
import csv, glob, collections, os
for name in ("sq3", "sq4"):
    fs = sorted(glob.glob(f"gpurun_out/final/pmc_{name}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not fs: print("no file", name); continue
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[-1])):
        if "iir_bank" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    for c, v in sorted(agg.items()): print(f"{name} {c:28s} {v / cnt[c]:.6g} (n={cnt[c]})")
PY
