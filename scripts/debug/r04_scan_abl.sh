#!/bin/bash
# where the scan kernel's time goes: ablated builds (wrong results on purpose) timed by rocprofv3
cd "$GRAFT_REPO_ROOT/openpbso_amd/csrc"
for st in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -fno-slp-vectorize -DPBSO_SCAN_STOP=$st -c kernels_scan.hip -o /tmp/ks$st.o
  hipcc --offload-arch=gfx950 -shared -fPIC kernels_iir.o kernels_block.o /tmp/ks$st.o kernels_pipe.o kernels_exact.o engine.o loaders.o capi.o group.o -ldl -o /tmp/libscan$st.so
done
cd "$GRAFT_REPO_ROOT"
for st in 1 2 3 9; do
  [ $st = 9 ] && unset PBSO_LIB || export PBSO_LIB=/tmp/libscan$st.so
  echo "== stop $st"; bash scripts/debug/r04_prof.sh ${1:-1} ${2:-512} | grep iir_scan
done
