# A/B/C of one environment switch on one box: bash scripts/debug/ab_env3.sh VAR a b c [bench args]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
var=$1; a=$2; b=$3; c=$4; shift; shift; shift; shift
for rep in 1 2 3; do for v in $a $b $c; do
  env $var=$v python bench.py --no-cpu-baseline --no-parity "$@" > gpurun_out/ab/bench_${v}_$rep.json 2> gpurun_out/ab/bench_${v}_$rep.err
done; done
