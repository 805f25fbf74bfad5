#!/bin/bash
# does the size of the warm-up copies in pbso_finalize matter for the runtime's one-time 6-7 ms blocks?
cd "$GRAFT_REPO_ROOT/openpbso_amd/csrc"
for mb in 4; do
  for f in engine; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DPBSO_WARM_CHUNK_MB=$mb -c $f.cpp -o /tmp/${f}_w$mb.o; done
  hipcc --offload-arch=gfx950 -shared -fPIC kernels_iir.o kernels_block.o kernels_scan.o kernels_pipe.o kernels_exact.o /tmp/engine_w$mb.o loaders.o capi.o group.o -ldl -o /tmp/libwarm$mb.so
done
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
echo "1 MB:"; python scripts/debug/r04_stall.py 1024 60 | tail -1
echo "4 MB:"; PBSO_LIB=/tmp/libwarm4.so python scripts/debug/r04_stall.py 1024 60 | tail -1
echo "none:"; PBSO_WARM_COPIES=0 python scripts/debug/r04_stall.py 1024 60 | tail -1
done
