cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/c5p
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5p/stats -- python3 $R/bench.py --no-cpu-baseline --no-parity --objects 8 --modes 4096 --scenario scraping --steps 30 --warmup 2 > $R/gpurun_out/c5p/log.txt 2>&1)
f=$(find gpurun_out/c5p/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/c5p/kernel_stats.csv
