# A/B of one environment switch on one box: bash scripts/debug/ab_env.sh VAR valueA valueB [bench args]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
var=$1; a=$2; b=$3; shift; shift; shift
for rep in 1 2 3; do for v in $a $b; do
  env $var=$v python bench.py --no-cpu-baseline --no-parity "$@" > gpurun_out/ab/bench_${v}_$rep.json 2> gpurun_out/ab/bench_${v}_$rep.err
done; done
