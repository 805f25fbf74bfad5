cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/ab/tests.txt
for rep in 1 2 3; do
for lib in A new; do
  if [ $lib = A ]; then export PBSO_LIB=$PWD/openpbso_amd/libpbso_A.so; else unset PBSO_LIB; fi
  python bench.py --no-cpu-baseline > gpurun_out/ab/bench_${lib}_$rep.json 2> gpurun_out/ab/bench_${lib}_$rep.err
done; done
