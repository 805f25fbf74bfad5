#!/bin/bash
# where the scan kernel's time goes: ablated builds (wrong results on purpose: PBSO_SCAN_STOP cuts the kernel short after its stage
# 1 / 2 / 3; 9 = the product) timed by rocprofv3 with the device otherwise IDLE (scripts/debug/r05_scan_isolated.py)
cd "$GRAFT_REPO_ROOT/openpbso_amd/csrc"
for st in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -fno-slp-vectorize -DPBSO_SCAN_STOP=$st -c kernels_scan.hip -o /tmp/ks$st.o
  hipcc --offload-arch=gfx950 -shared -fPIC kernels_iir.o kernels_block.o /tmp/ks$st.o kernels_pipe.o kernels_exact.o engine.o loaders.o capi.o group.o -ldl -o /tmp/libscan$st.so
done
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for st in 1 2 3 9; do
  [ $st = 9 ] && unset PBSO_LIB || export PBSO_LIB=/tmp/libscan$st.so
  rm -rf /tmp/scanabl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/scanabl -o p -- python3 scripts/debug/r05_scan_isolated.py ${1:-1} ${2:-512} > /dev/null 2>&1
  python3 - $st ${1:-1} ${2:-512} <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/scanabl/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'iir_scan' in r['Name']:
            print(f"{sys.argv[2]} x {sys.argv[3]} x 86, stop {sys.argv[1]}: {r['Name'].split('(')[0].split('::')[-1][:28]} calls {r['Calls']} avg_us {float(r['AverageNs']) / 1e3:.2f} min {float(r['MinNs']) / 1e3:.2f} max {float(r['MaxNs']) / 1e3:.2f}")
PY
done
# ... and the scan cut along the time axis (the product's choice for 8 chunks of 11 buffers)
unset PBSO_LIB; export PBSO_SCAN_KERNEL=0
rm -rf /tmp/scanabl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/scanabl -o p -- python3 scripts/debug/r05_scan_isolated.py ${1:-1} ${2:-512} > /dev/null 2>&1
python3 - ${1:-1} ${2:-512} <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/scanabl/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'iir_scan' in r['Name']:
            print(f"{sys.argv[1]} x {sys.argv[2]} x 86, stop 9, cut along the time axis (8 chunks): {r['Name'].split('(')[0].split('::')[-1][:28]} calls {r['Calls']} avg_us {float(r['AverageNs']) / 1e3:.2f} min {float(r['MinNs']) / 1e3:.2f} max {float(r['MaxNs']) / 1e3:.2f}")
PY
