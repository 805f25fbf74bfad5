# kernel time of ablated builds (results are wrong on purpose; --no-parity): bash scripts/debug/ablate.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abl
for rep in 1 2; do for v in base nowrite nosplit nomfma noread; do
  if [ $v = base ]; then unset PBSO_LIB; else export PBSO_LIB=$PWD/openpbso_amd/libpbso_abl_$v.so; fi
  python bench.py --no-cpu-baseline --no-parity --steps 40 > gpurun_out/abl/bench_${v}_$rep.json 2> gpurun_out/abl/bench_${v}_$rep.err
done; done
