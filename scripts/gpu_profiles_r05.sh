#!/bin/bash
# Round 5 evidence, collected on a gpurun box into gpurun_out/p5/ (copied to profiles/r05_* by scripts/collect_profiles_r05.py):
# bench lines of every BASELINE configuration and form, rocprofv3 kernel-trace stats of the headline command in BOTH block forms
# and of configs[1,2,4], PMC passes (own runs: --kernel-trace + --pmc only) for both block forms, workgroup census.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p5; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p5
(rocminfo | grep -E "Marketing Name|gfx9" | sort | uniq -c | head -4; lscpu | grep -E "Model name|^CPU\(s\)"; echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; free -g | head -2) > $O/env.txt 2>&1
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"; }
echo "== headline"
b default
b driver_flags_steps20_warmup5 --steps 20 --warmup 5
b form_block_bf16 --no-cpu-baseline --form block_bf16 --no-strong-share
b qnorm_off --no-cpu-baseline --qnorm off --no-second-form --no-strong-share
b host_delivery --host-delivery --no-cpu-baseline --no-second-form --no-strong-share
echo "== other BASELINE configurations"
# (one second of audio per step, as in rounds 1 - 3, and the bench's default ten)
b c2_1x512 --no-cpu-baseline --objects 1 --modes 512 --buffers 86 --steps 40 --warmup 2
b c3_64x256_listener --no-cpu-baseline --objects 64 --modes 256 --scenario listener --buffers 86 --steps 40 --warmup 2
b c5_8x4096_scraping --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2
b c5_8x4096_scraping_qnorm_off --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40 --warmup 2
b c2_1x512_10s_steps --no-cpu-baseline --objects 1 --modes 512
b c3_64x256_listener_10s_steps --no-cpu-baseline --objects 64 --modes 256 --scenario listener
b c5_8x4096_scraping_10s_steps --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping
b c5_8x4096_scraping_qnorm_off_10s_steps --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off
echo "== configs[4] on the paths the policy does NOT take (A/B): three-wave pipeline teams, the walk, the other kernel per qnorm mode"
PBSO_TIME_CHUNKS=-1 b c5_8x4096_scraping_three_wave_teams --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2
PBSO_PIPE_CONSUMERS=2 b c5_8x4096_scraping_qnorm_off_three_wave_teams --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40 --warmup 2
PBSO_ENGINE_OPTS=bank_kernel=1 b c5_8x4096_scraping_qnorm_off_cut_in_time --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40 --warmup 2
PBSO_ENGINE_OPTS=time_chunks=-1,pipe_consumers=4 b c5_8x4096_scraping_five_role_teams --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40 --warmup 2
echo "== the per-rank shares of configs[3] on 2 / 4 / 8 GPUs, alone (the default line carries them as strong_share), and the same without K5"
for o in 512 256 128; do
b share_${o}x512 --no-cpu-baseline --no-second-form --objects $o --buffers 86 --steps 40 --warmup 3
PBSO_TIME_CHUNKS=-1 b share_${o}x512_buffer_by_buffer --no-cpu-baseline --no-second-form --objects $o --buffers 86 --steps 40 --warmup 3
b share_${o}x512_10s_steps --no-cpu-baseline --no-second-form --no-one-second-leg --objects $o
done
PBSO_TIME_CHUNKS=-1 b c2_1x512_buffer_by_buffer --no-cpu-baseline --objects 1 --modes 512 --buffers 86 --steps 40 --warmup 2
PBSO_TIME_CHUNKS=-1 b c3_64x256_listener_buffer_by_buffer --no-cpu-baseline --objects 64 --modes 256 --scenario listener --buffers 86 --steps 40 --warmup 2
PBSO_ENGINE_OPTS=time_chunks=-1,bank_kernel=1 b c2_1x512_block_kernel_only --no-cpu-baseline --objects 1 --modes 512 --buffers 86 --steps 40 --warmup 2
echo "== N > 1 path on one GPU"
PBSO_BENCH_BACKEND=gloo b 2ranks_one_gpu_gloo --no-cpu-baseline --gpus 2 --steps 20 --warmup 2
(PBSO_BENCH_GATHER_SELF=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 1 --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_1rank_torchrun_device_group_selfgather.json 2> $O/bench_1rank.err; echo "selfgather rc=$?")
echo "== rocprofv3 kernel trace + stats"
st() { name=$1; shift; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -- python3 $R/bench.py --no-cpu-baseline --no-second-form "$@" > $O/st_$name.log 2>&1); f=$(find $O/st_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$name.csv; [ $name = default ] && python scripts/trace_gaps.py $O/st_$name > $O/trace_gaps.txt 2>&1; rm -rf $O/st_$name; echo "stats $name: $(sed -n 2p $O/kernel_stats_$name.csv | cut -c1-70 | tr -d '\n') ... $(sed -n 2p $O/kernel_stats_$name.csv | awk -F, '{print $(NF-5), $(NF-4)}')"; }
st default --no-one-second-leg
st c2_1x512 --objects 1 --modes 512 --buffers 86 --steps 40
st c3_64x256_listener --objects 64 --modes 256 --scenario listener --buffers 86 --steps 40
st c5_8x4096_scraping --objects 8 --modes 4096 --scenario scraping --buffers 86 --steps 40
st c5_8x4096_scraping_qnorm_off --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40
PBSO_ENGINE_OPTS=bank_kernel=1 st c5_8x4096_scraping_qnorm_off_cut_in_time --objects 8 --modes 4096 --scenario scraping --qnorm off --buffers 86 --steps 40
st share_128x512 --objects 128 --buffers 86 --steps 40
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tl_128 -- python3 $R/bench.py --no-cpu-baseline --no-second-form --no-parity --no-strong-share --objects 128 --buffers 86 --steps 20 --warmup 3 > /dev/null 2>&1); python scripts/debug/r04_timeline.py $O/tl_128 24 > $O/timeline_share_128x512.txt 2>&1; rm -rf $O/tl_128
echo "== PMC passes"
pmc() { form=$1; name=$2; shift; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_${form}_$name -- python3 $R/bench.py --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-parity --no-second-form --no-strong-share --no-one-second-leg --form $form > $O/pmc_${form}_$name.log 2>&1); echo "pmc $form $name rc=$?"; }
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
