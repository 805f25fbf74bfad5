#!/bin/bash
# First GPU pass: VALU/LDS microbenchmark, parity tests, smoke, a short bench of
# both kernel builds, and a rocprofv3 kernel trace.  Everything under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo ==" > gpurun_out/env.txt
(rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8; nproc; lscpu | grep "Model name"; free -g | head -2) >> gpurun_out/env.txt 2>&1
echo "== microbench =="; timeout 300 ./openpbso_amd/microbench_gfx950 > gpurun_out/microbench.txt 2>&1; echo "rc=$?"; cat gpurun_out/microbench.txt
echo "== smoke =="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; echo "rc=$?"; tail -5 gpurun_out/smoke.txt
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -40 gpurun_out/pytest_gpu.txt
for packed in 0 1; do
  for mpl in 4 2; do
    echo "== bench packed=$packed mpl=$mpl =="
    PBSO_IIR_PACKED=$packed timeout 600 python bench.py --steps 3 --warmup 1 --modes-per-lane $mpl --no-cpu-baseline > gpurun_out/bench_p${packed}_r${mpl}.json 2> gpurun_out/bench_p${packed}_r${mpl}.err; echo "rc=$?"
    cat gpurun_out/bench_p${packed}_r${mpl}.json; tail -3 gpurun_out/bench_p${packed}_r${mpl}.err
  done
done
echo "== bench default with cpu baseline =="
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "rc=$?"; cat gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
echo "== rocprofv3 kernel trace =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_first" -- python3 "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_first.log" 2>&1; echo "rc=$?"
cd "$OLDPWD"; find gpurun_out/prof_first -name "*stats*" | head; for f in $(find gpurun_out/prof_first -name "*kernel_stats.csv" | head -1); do head -12 $f; done
