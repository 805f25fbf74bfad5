#!/bin/bash
# re-measures only the configs[4] lines and the pipeline kernel's census of scripts/gpu_profiles_r03.sh (same file names)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/p3; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/p3
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"; }
b c5_8x4096_scraping --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --steps 40 --warmup 2
b c5_8x4096_scraping_qnorm_off --no-cpu-baseline --objects 8 --modes 4096 --scenario scraping --qnorm off --steps 40 --warmup 2
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_c5 -- python3 $R/bench.py --no-cpu-baseline --no-second-form --objects 8 --modes 4096 --scenario scraping --steps 40 > $O/st_c5.log 2>&1); f=$(find $O/st_c5 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c5_8x4096_scraping.csv; rm -rf $O/st_c5
(timeout 300 python scripts/debug/census_split.py; timeout 300 python scripts/debug/census_split.py off; timeout 300 python scripts/debug/census_split_free.py) 2>&1 | grep -v amdgpu.ids > $O/census_pipeline_kernel.txt
cat $O/census_pipeline_kernel.txt
