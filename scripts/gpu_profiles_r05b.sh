#!/bin/bash
# Round 5 evidence, second half (the first run of scripts/gpu_profiles_r05.sh hit its time limit in the PMC passes): PMC, census, latency,
# scan stages, host delivery, group tests, the one-rank torchrun line -- into gpurun_out/p5/ as the first half
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p5; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p5
(PBSO_BENCH_GATHER_SELF=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 1 --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_1rank_torchrun_device_group_selfgather.json 2> $O/bench_1rank.err; echo "selfgather rc=$?")
echo "== PMC passes"
pmc() { form=$1; name=$2; shift; shift; (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_${form}_$name -- python3 $R/bench.py --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-parity --no-second-form --no-strong-share --no-one-second-leg --form $form > $O/pmc_${form}_$name.log 2>&1); echo "pmc $form $name rc=$?"; }
for form in block; do
pmc $form m1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
pmc $form m2 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
pmc $form m3 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_CYCLES
pmc $form fetch FETCH_SIZE
pmc $form write WRITE_SIZE
python - $form <<'PY' > gpurun_out/p5/pmc_summary_$form.txt
import csv, glob, collections, sys
form = sys.argv[1]
print(f"per-dispatch averages for pbso kernels (rocprofv3 --kernel-trace --pmc, separate passes; bench.py --steps 3 --warmup 1 --settle 0 --no-second-form --form {form})")
for name in ("m1", "m2", "m3", "fetch", "write"):
    fs = glob.glob(f"gpurun_out/p5/pmc_{form}_{name}/**/*counter_collection.csv", recursive=True)
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
echo "== census"
PBSO_CENSUS=1 timeout 300 python scripts/census.py 1024 2>&1 | grep -v amdgpu.ids > $O/census_1024x512_block_f32.txt; tail -5 $O/census_1024x512_block_f32.txt
timeout 300 python scripts/latency.py > $O/realtime_latency.txt 2>&1
(for a in "" "off"; do python scripts/debug/r05_census_tc_dense.py $a 1 2>&1 | grep -v amdgpu.ids; done) > $O/census_8x4096_scraping_cut_in_time.txt
(PBSO_PIPE_CONSUMERS=4 python scripts/debug/census_split.py off; PBSO_PIPE_CONSUMERS=4 python scripts/debug/census_split.py; python scripts/debug/r05_pipe5_placement.py off) 2>&1 | grep -v amdgpu.ids > $O/census_8x4096_scraping_five_role_teams.txt
echo "== the device group: RCCL on one rank, the loopback ranks"
timeout 600 python -m pytest tests/test_group.py -q -m gpu -rA 2>&1 | grep -E "PASSED|FAILED|passed|failed|RCCL version|Librccl" > $O/group_tests.txt
echo "== the scan kernel by stages (ablated builds, wrong results on purpose) and the host delivery paths"
(echo "every build timed with the device otherwise idle (step, wait, step: scripts/debug/r05_scan_abl.sh <objects> <modes>)"; bash scripts/debug/r05_scan_abl.sh 1 512; bash scripts/debug/r05_scan_abl.sh 128 512) 2>&1 | grep -E "every build|stop" > $O/scan_kernel_stages.txt
timeout 300 python scripts/debug/r04_d2h.py 2>&1 | grep -v amdgpu.ids > $O/host_delivery.txt
ls $O | head -80
