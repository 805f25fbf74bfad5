cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/tr
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/tr/stats -- python3 $R/bench.py --no-cpu-baseline --no-parity --steps 30 > $R/gpurun_out/tr/log.txt 2>&1)
python scripts/trace_gaps.py gpurun_out/tr/stats > gpurun_out/tr/gaps.txt 2>&1
f=$(find gpurun_out/tr/stats -name "*memory_copy_trace.csv" | head -1); tail -5 "$f" > gpurun_out/tr/memcpy_tail.txt
k=$(find gpurun_out/tr/stats -name "*kernel_trace.csv" | head -1); tail -6 "$k" | cut -c1-300 > gpurun_out/tr/kernel_tail.txt
