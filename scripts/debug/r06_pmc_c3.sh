#!/bin/bash
# HBM traffic of the listener scene's kernels (64 x 256, a move per buffer, 86 buffers per step): FETCH_SIZE and WRITE_SIZE in passes of
# their own, the program directly behind `--`; the engine's policy orders its launches by events under counter collection
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p6; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p6
BARGS="--objects 64 --modes 256 --scenario listener --buffers 86 --steps 6 --warmup 2 --settle 0 --clock-ramp-ms 0 --no-cpu-baseline --no-parity --no-second-form --no-strong-share --no-one-second-leg --submit-thread 0"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_c3_$c -- python3 $R/bench.py $BARGS > $O/pmc_c3_$c.log 2>&1); echo "pmc c3 $c rc=$?"
done
python - <<'PY' > gpurun_out/p6/pmc_summary_c3_listener.txt
import csv, glob, collections
print("per-dispatch averages, 64 x 256 listener scene at 86 buffers per step (rocprofv3 --kernel-trace --pmc, one counter per pass; KB as the counter reports them;")
print("gfx950 FETCH_SIZE counts 64 B per 128-B request: reads = 2 x FETCH_SIZE, MI355X_MICROARCH.md)")
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/p6/pmc_c3_{name}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(name, "no counter file"); continue
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-48:]
        if not any(t in k for t in ("ffat", "iir_", "copy_rows", "mix_objects")): continue
        agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in sorted(agg):
        print(f"{name:11s} {k:50s} {agg[k] / cnt[k]:12.1f} KB  (n={cnt[k]})")
PY
rm -rf $O/pmc_c3_*/; cat gpurun_out/p6/pmc_summary_c3_listener.txt
