# configs[4] (8 x 4096, sustained scraping): team shape / modes per lane / qnorm sweep of the per-sample kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c5
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-parity --objects 8 --modes 4096 --scenario scraping --steps 30 --warmup 2 $EXTRA > gpurun_out/c5/$name.json 2> gpurun_out/c5/$name.err; }
EXTRA="" run base X=1
EXTRA="--modes-per-lane 2" run r2 X=1
EXTRA="--modes-per-lane 4" run r4 X=1
EXTRA="" run tw1 PBSO_TEAM_WAVES=1
EXTRA="" run tw4 PBSO_TEAM_WAVES=4
EXTRA="" run tw8 PBSO_TEAM_WAVES=8
EXTRA="--qnorm off" run qoff X=1
EXTRA="--qnorm closed" run qclosed X=1
EXTRA="--form velocity" run vel X=1
