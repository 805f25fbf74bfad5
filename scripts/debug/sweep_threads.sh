cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for rep in 1 2 3; do for t in 2 3 4 6; do
  python bench.py --no-cpu-baseline --no-parity --plan-threads $t > gpurun_out/ab/bench_t${t}_$rep.json 2> gpurun_out/ab/bench_t${t}_$rep.err
done; done
