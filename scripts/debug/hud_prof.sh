cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/hud
python scripts/hud_sphere.py > gpurun_out/hud/new.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hud/new -- python3 $R/scripts/hud_sphere.py > /dev/null 2>&1)
export PBSO_LIB=$R/openpbso_amd/libpbso_A.so
python scripts/hud_sphere.py > gpurun_out/hud/old.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hud/old -- python3 $R/scripts/hud_sphere.py > /dev/null 2>&1)
for v in new old; do f=$(find gpurun_out/hud/$v -name "*kernel_stats.csv" | head -1); grep -i "ffat" "$f" | cut -c1-60,200-400 > gpurun_out/hud/${v}_k.txt; done
